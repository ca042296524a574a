// Internal interface between sampler.hip and unet.hip.
#pragma once
#include "common.h"

namespace surfd {
// Precomputes, for `rows` (step, sample) pairs, everything of the denoiser that depends only
// on the timestep / conditioning (time_embed MLP, label/context embedding, the 22 ResBlock
// emb_layers): t_rows[rows] are original-scale timesteps (host), row r uses sample r % B.
// shared: nothing but t enters the embedding (no context, no labels) -> one row per step, read by every sample.
int unet_prepare_embeddings(surfd_unet *u, const int64_t *t_rows_host, int rows, const float *ctx,
                            const int64_t *cls, int B, hipStream_t st, bool shared = false);
// One denoiser evaluation using embedding rows [row0, row0 + B) of the prepared table (row row0 alone when the rows
// are shared).  With step_ptr != nullptr the row block is (*step_ptr) * B (resp. row *step_ptr) instead (read on the
// device: lets one captured hipGraph serve every iteration of the reverse loop).
int unet_forward_prepared(surfd_unet *u, const float *x, int row0, float *out, int B, int L, hipStream_t st,
                          const int *step_ptr = nullptr);

// Per-handle state of the graph-replayed reverse loop (owned by the unet handle).
struct LoopState {
    int *step_ctr = nullptr;        // device: current loop iteration k
    float *x = nullptr, *x0 = nullptr;   // device: state and x0 prediction [B*L]
    size_t cap = 0;                 // floats allocated for x / x0
    void *params = nullptr;         // device: LoopParams (caller pointers, refreshed per call)
    float *tab = nullptr;           // device: per-iteration coefficient rows [T][8]
    int tab_cap = 0;
    hipGraphExec_t exec = nullptr;  // cached instantiated step graph
    hipGraph_t graph = nullptr;
    hipStream_t cap_stream = nullptr;   // private stream used only to record the graph (the caller's may be the null stream)
    long key[6] = {0, 0, 0, 0, 0, 0};
};
LoopState *unet_loop_state(surfd_unet *u);
// changes whenever a device buffer that a captured loop graph refers to is reallocated (or the kernel choice changes)
long unet_workspace_generation(surfd_unet *u);
}  // namespace surfd
