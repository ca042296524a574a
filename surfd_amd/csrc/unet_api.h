// Internal interface between sampler.hip and unet.hip.
#pragma once
#include "common.h"

namespace surfd {
// Precomputes, for `rows` (step, sample) pairs, everything of the denoiser that depends only
// on the timestep / conditioning (time_embed MLP, label/context embedding, the 22 ResBlock
// emb_layers): t_rows[rows] are original-scale timesteps (host), row r uses sample r % B.
int unet_prepare_embeddings(surfd_unet *u, const int64_t *t_rows_host, int rows, const float *ctx,
                            const int64_t *cls, int B, hipStream_t st);
// One denoiser evaluation using embedding rows [row0, row0 + B) of the prepared table.
int unet_forward_prepared(surfd_unet *u, const float *x, int row0, float *out, int B, int L, hipStream_t st);
}  // namespace surfd
