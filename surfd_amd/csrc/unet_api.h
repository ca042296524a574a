// Internal interface between sampler.hip and unet.hip.
#pragma once
#include "common.h"

namespace surfd {
// Posterior update of the reverse loop folded into the epilogue of the head convolution (conv_f16x2.hip): the launch
// that produces the x0 prediction also writes x_{t-1} (and the trajectory slot) and, through an arrival ticket, advances
// the loop counter — two launches fewer per iteration than loop_step_kernel + loop_advance_kernel (sampler.hip).
struct LoopParams {          // device-resident; refreshed per call so the cached graph is pointer-free
    const float *noise;      // [T'+1, n]
    float *traj;             // [T', n] or null
    int T;
};
struct LoopFuse {
    const float *tab;        // device: per-iteration coefficient rows [T][8]
    const LoopParams *lp;    // device
    float *x;                // device: loop state [B*L], updated in place
    int *step;               // device: loop counter, advanced by the last workgroup of the head launch
    int *done;               // device: arrival ticket of the head launch's workgroups (zero between launches)
    int sampler, clip;       // 0 DDPM / 1 DDIM; clip_denoised
    float eta;
};

// Precomputes, for `rows` (step, sample) pairs, everything of the denoiser that depends only
// on the timestep / conditioning (time_embed MLP, label/context embedding, the 22 ResBlock
// emb_layers): t_rows[rows] are original-scale timesteps (host), row r uses sample r % B.
// shared: nothing but t enters the embedding (no context, no labels) -> one row per step, read by every sample.
int unet_prepare_embeddings(surfd_unet *u, const int64_t *t_rows_host, int rows, const float *ctx,
                            const int64_t *cls, int B, hipStream_t st, bool shared = false);
// Conditioned reverse loops: what depends only on t (time_embed MLP, one row per STEP: t_steps_host[T]) and what depends only on
// the sample (context / label embedding, one row per SAMPLE) are evaluated once; the 22 per-ResBlock rows of an iteration are
// then produced inside the loop by one Linear launch (unet_forward_prepared with a step pointer).  Nothing is sized T x B.
int unet_prepare_loop_embeddings(surfd_unet *u, const int64_t *t_steps_host, int T, const float *ctx, const int64_t *cls, int B,
                                 hipStream_t st);
// One denoiser evaluation using embedding rows [row0, row0 + B) of the prepared table (row row0 alone when the rows
// are shared).  With step_ptr != nullptr the row block is (*step_ptr) * B (resp. row *step_ptr) instead (read on the
// device: lets one captured hipGraph serve every iteration of the reverse loop).
// lf (a DEVICE pointer) != nullptr asks for the posterior update inside the head convolution; *lf_done says whether that happened (it does
// not when the head runs on the exact-fp32 kernel: the caller then launches the separate step kernels).
int unet_forward_prepared(surfd_unet *u, const float *x, int row0, float *out, int B, int L, hipStream_t st,
                          const int *step_ptr = nullptr, const LoopFuse *lf = nullptr, bool *lf_done = nullptr);
// One posterior update of element e (shared by loop_step_kernel and the fused head epilogue; every step separately rounded,
// diffusion/gaussian_diffusion.py:471-520, 711-761)
__device__ __forceinline__ float loop_update(int sampler, int clip, float eta, const float *row, float x0v, float xv, float zv) {
    float xs = x0v;
    if (clip) xs = fminf(fmaxf(xs, -1.f), 1.f);
    if (sampler == 0) {
        const float mean = __fadd_rn(__fmul_rn(row[0], xs), __fmul_rn(row[1], xv));
        const float sd = expf(__fmul_rn(0.5f, row[2]));
        return __fadd_rn(mean, __fmul_rn(__fmul_rn(row[3], sd), zv));
    }
    const float sra = row[0], srm1 = row[1], ab = row[2], abp = row[3];
    const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(sra, xv), xs), srm1);
    const float sigma = __fmul_rn(__fmul_rn(eta, __fsqrt_rn(__fdiv_rn(__fsub_rn(1.f, abp), __fsub_rn(1.f, ab)))),
                                  __fsqrt_rn(__fsub_rn(1.f, __fdiv_rn(ab, abp))));
    const float mean = __fadd_rn(__fmul_rn(xs, __fsqrt_rn(abp)),
                                 __fmul_rn(__fsqrt_rn(__fsub_rn(__fsub_rn(1.f, abp), __fmul_rn(sigma, sigma))), eps));
    return __fadd_rn(mean, __fmul_rn(__fmul_rn(row[4], sigma), zv));
}

// Per-handle state of the graph-replayed reverse loop (owned by the unet handle).
struct LoopState {
    int *step_ctr = nullptr;        // device: current loop iteration k ([0]) and the head launch's arrival ticket ([1])
    float *x = nullptr, *x0 = nullptr;   // device: state and x0 prediction [B*L]
    size_t cap = 0;                 // floats allocated for x / x0
    void *params = nullptr;         // device: LoopParams (caller pointers, refreshed per call), followed by the LoopFuse record
    float *tab = nullptr;           // device: per-iteration coefficient rows [T][8]
    int tab_cap = 0;
    hipGraphExec_t exec = nullptr;  // cached instantiated step graph
    hipGraph_t graph = nullptr;
    hipStream_t cap_stream = nullptr;   // private stream used only to record the graph (the caller's may be the null stream)
    hipStream_t poll_stream = nullptr;  // surfd_unet_loop_progress: reads the loop counter beside the stream the loop runs on
    int run_T = 0, run_done = 0;        // surfd_sample_loop_begin / _run / _end: iterations of the open loop, launched so far
    long run_n = 0;                     // floats of its state
    void *prof_ev = nullptr;            // profiling bracket opened by begin, closed by end
    long key[6] = {0, 0, 0, 0, 0, 0};
};
LoopState *unet_loop_state(surfd_unet *u);
int unet_device(surfd_unet *u);          // the device the handle was allocated on (-1 before the first device call)
// changes whenever a device buffer that a captured loop graph refers to is reallocated (or the kernel choice changes)
long unet_workspace_generation(surfd_unet *u);
}  // namespace surfd
