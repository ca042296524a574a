// Posterior updates of the reverse diffusion loop and the host-free loop driver.
//
//   p_sample            diffusion/gaussian_diffusion.py:471-520 (+ p_mean_variance :258-363,
//                       q_posterior_mean_variance :234-256): START_X / FIXED_SMALL
//   ddim_sample         diffusion/gaussian_diffusion.py:711-761
//   p_sample_loop_progressive / ddim_sample_loop_progressive   :635-708 / :908-972
//   _WrappedModel       diffusion/respace.py:123-128 (loop index -> original timestep)
//
// Every arithmetic step is kept as a separately rounded fp32 operation in the order torch
// evaluates the reference expressions (no FMA contraction), so a step is reproducible
// against the oracle to the last bit given the same x0.
#include "common.h"
#include "unet_api.h"
#include <string.h>
#include <stdlib.h>

namespace surfd {

__global__ void ddpm_step_kernel(const float *x_t, const float *x0, const float *z, float c1, float c2, float logvar,
                                 int nonzero, int clip, float *out, long n) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        float xs = x0[e];
        if (clip) xs = fminf(fmaxf(xs, -1.f), 1.f);
        const float mean = __fadd_rn(__fmul_rn(c1, xs), __fmul_rn(c2, x_t[e]));
        // nonzero_mask * exp(0.5 * log_variance) * noise
        const float sd = expf(__fmul_rn(0.5f, logvar));
        const float nz = nonzero ? 1.f : 0.f;
        out[e] = __fadd_rn(mean, __fmul_rn(__fmul_rn(nz, sd), z[e]));
    }
}

__global__ void ddim_step_kernel(const float *x_t, const float *x0, const float *z, float sra, float srm1, float ab,
                                 float abp, float eta, int nonzero, int clip, float *out, long n) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        float xs = x0[e];
        if (clip) xs = fminf(fmaxf(xs, -1.f), 1.f);
        const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(sra, x_t[e]), xs), srm1);
        const float sigma = __fmul_rn(__fmul_rn(eta, __fsqrt_rn(__fdiv_rn(__fsub_rn(1.f, abp), __fsub_rn(1.f, ab)))),
                                      __fsqrt_rn(__fsub_rn(1.f, __fdiv_rn(ab, abp))));
        const float mean = __fadd_rn(__fmul_rn(xs, __fsqrt_rn(abp)),
                                     __fmul_rn(__fsqrt_rn(__fsub_rn(__fsub_rn(1.f, abp), __fmul_rn(sigma, sigma))), eps));
        const float nz = nonzero ? 1.f : 0.f;
        out[e] = __fadd_rn(mean, __fmul_rn(__fmul_rn(nz, sigma), z[e]));
    }
}

static int launch_ddpm(const float *x_t, const float *x0, const float *z, float c1, float c2, float lv, int nonzero,
                       int clip, float *out, long n, hipStream_t st) {
    hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(n, 256), 1024)), dim3(256), 0, st,
                       x_t, x0, z, c1, c2, lv, nonzero, clip, out, n);
    LAUNCH_CHECK();
    return SURFD_OK;
}

static int launch_ddim(const float *x_t, const float *x0, const float *z, float sra, float srm1, float ab, float abp,
                       float eta, int nonzero, int clip, float *out, long n, hipStream_t st) {
    hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(n, 256), 1024)), dim3(256), 0, st,
                       x_t, x0, z, sra, srm1, ab, abp, eta, nonzero, clip, out, n);
    LAUNCH_CHECK();
    return SURFD_OK;
}

}  // namespace surfd

using namespace surfd;

extern "C" {

int surfd_ddpm_step(const float *x_t, const float *x0, const float *z, float coef1, float coef2, float log_variance,
                    int t_nonzero, int clip_denoised, float *out, int64_t n, surfd_stream s) {
    if (!x_t || !x0 || !z || !out || n < 0) SURFD_FAIL(SURFD_ERR_ARG, "surfd_ddpm_step: bad argument");
    if (n == 0) return SURFD_OK;
    return launch_ddpm(x_t, x0, z, coef1, coef2, log_variance, t_nonzero, clip_denoised, out, n, as_stream(s));
}

int surfd_ddim_step(const float *x_t, const float *x0, const float *z, float sqrt_recip_ab, float sqrt_recipm1_ab,
                    float ab, float ab_prev, float eta, int t_nonzero, int clip_denoised, float *out, int64_t n,
                    surfd_stream s) {
    if (!x_t || !x0 || !z || !out || n < 0) SURFD_FAIL(SURFD_ERR_ARG, "surfd_ddim_step: bad argument");
    if (n == 0) return SURFD_OK;
    return launch_ddim(x_t, x0, z, sqrt_recip_ab, sqrt_recipm1_ab, ab, ab_prev, eta, t_nonzero, clip_denoised, out, n,
                       as_stream(s));
}

}  // extern "C"

// ---- graph-replayed loop ---------------------------------------------------------------------------
// The ~115 launches of one iteration are captured ONCE into a hipGraph and replayed T' times: at
// ~23 us of host time per launch the direct loop was host-bound (2.6 s for 1000 steps).  Everything
// that changes from iteration to iteration is read on the device through a loop counter: the
// embedding rows (unet.hip), the coefficient row, the noise row and the trajectory slot.
namespace surfd {

// one row per loop iteration k (t index i = T'-1-k): DDPM {c1, c2, logvar, nonzero} / DDIM {sra, srm1, ab, abp, nonzero}
__global__ void loop_step_kernel(int sampler, int clip, float eta, const float *tab, const LoopParams *lp,
                                 const int *step_ptr, float *x, const float *x0, long n) {
    const int k = *step_ptr;
    const float *row = tab + (long)k * 8;
    const float *z = lp->noise + (long)(1 + k) * n;
    float *tr = lp->traj ? lp->traj + (long)k * n : nullptr;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float out = loop_update(sampler, clip, eta, row, x0[e], x[e], z[e]);
        x[e] = out;
        if (tr) tr[e] = out;
    }
}

__global__ void loop_advance_kernel(int *step_ptr) { *step_ptr += 1; }

}  // namespace surfd

extern "C" {

// A fused loop in three parts (round 5): begin = everything up to the instantiated step graph (embedding rows, coefficient table,
// state <- noise row 0, counter <- 0), run = n graph replays, end = the state copied out.  surfd_sample_loop is the three in a
// row; a host thread that interleaves several loops calls begin on each, run in turns, end on each
// (SpacedDiffusion.fused_loops_interleaved): one deterministic submission order instead of one racing thread per loop.
int surfd_sample_loop_begin(surfd_unet *u, const surfd_sampler_cfg *cfg, const float *noise, const float *ctx,
                            const int64_t *cls, float *traj, int B, int L, surfd_stream s) {
    if (!u || !cfg || !noise || B < 1 || L < 1) SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop: bad argument");
    {   // a begin that fails leaves no loop open (run / end then report SURFD_ERR_STATE); the profiling bracket of a loop that was
        // opened and never closed is dropped
        LoopState *open = unet_loop_state(u);
        open->run_T = 0;
        if (open->prof_ev) { (void)hipEventDestroy((hipEvent_t)open->prof_ev); open->prof_ev = nullptr; }
    }
    const int T = cfg->num_steps;
    if (T < 1 || !cfg->timestep_map) SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop: bad schedule");
    if (cfg->sampler == 0 && (!cfg->coef1 || !cfg->coef2 || !cfg->log_variance))
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop: DDPM tables missing");
    if (cfg->sampler == 1 && (!cfg->sqrt_recip_ab || !cfg->sqrt_recipm1_ab || !cfg->ab || !cfg->ab_prev))
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop: DDIM tables missing");
    hipStream_t st = as_stream(s);
    const long n = (long)B * L;
    struct ProfBracket {                       // closed by surfd_sample_loop_end; destroyed here if this call fails
        hipEvent_t ev;
        bool handed_over = false;
        ~ProfBracket() { if (ev && !handed_over) (void)hipEventDestroy(ev); }
    } prof{prof_begin(PROF_LOOP, st)};
    // every timestep-only quantity of the denoiser for all T' iterations at once: loop
    // iteration k runs original timestep timestep_map[T'-1-k] for every sample
    // (one row per step when nothing but t enters the embedding, else one per (step, sample): T' x B x 14112 floats)
    const bool shared = !ctx && !cls;
    const int per_step = shared ? 1 : B;
    if ((long)T * per_step > 2000000) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "surfd_sample_loop: %d steps x %d samples of embedding rows", T, per_step);
    // conditioned: the time part per step and the sample part per sample, the 22 per-ResBlock rows inside the graph
    // (SURFD_EMB_TABLE=1 keeps round 3's [T' x B, 14112] table: A/B timing)
    static const bool table_env = getenv("SURFD_EMB_TABLE") && atoi(getenv("SURFD_EMB_TABLE")) == 1;
    // a model with BOTH class labels and a context vector (no Surf-D configuration has one: category models carry labels, the
    // sketch / text / image models a context) keeps the table: the reference adds ((time + label) + context)
    // (models/openaimodel.py:725-735), the in-graph rows add time + (context + label) — the same sum in another rounding order
    const bool ingraph = !shared && !table_env && !(ctx && cls);
    std::vector<int64_t> t_rows((size_t)T * (ingraph ? 1 : per_step));
    for (int k = 0; k < T; ++k)
        for (int b = 0; b < (ingraph ? 1 : per_step); ++b) t_rows[(size_t)k * (ingraph ? 1 : per_step) + b] = cfg->timestep_map[T - 1 - k];
    int rc = ingraph ? unet_prepare_loop_embeddings(u, t_rows.data(), T, ctx, cls, B, st)
                     : unet_prepare_embeddings(u, t_rows.data(), T * per_step, ctx, cls, B, st, shared);
    if (rc) return rc;
    LoopState *ls = unet_loop_state(u);
    if (!ls->step_ctr) {
        // the polling stream of surfd_unet_loop_progress is made HERE, before the counter exists: the progress thread only ever
        // READS the two fields (it returns -1 while step_ctr is null), it never allocates beside this call
        if (!ls->poll_stream) HIP_TRY(hipStreamCreateWithFlags(&ls->poll_stream, hipStreamNonBlocking));
        int *ctr = nullptr;
        HIP_TRY(hipMalloc((void **)&ctr, 2 * sizeof(int)));
        HIP_TRY(hipMalloc(&ls->params, sizeof(LoopParams) + sizeof(LoopFuse)));
        HIP_TRY(hipMemsetAsync(ctr, 0, 2 * sizeof(int), st));
        __atomic_store_n(&ls->step_ctr, ctr, __ATOMIC_RELEASE);
    }
    if ((size_t)n > ls->cap) {
        if (ls->x) HIP_TRY(hipFree(ls->x));
        if (ls->x0) HIP_TRY(hipFree(ls->x0));
        HIP_TRY(hipMalloc((void **)&ls->x, n * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&ls->x0, n * sizeof(float)));
        ls->cap = n;
        ls->key[0] = 0;                       // pointers baked into the cached graph changed
    }
    if (T > ls->tab_cap) {
        if (ls->tab) HIP_TRY(hipFree(ls->tab));
        HIP_TRY(hipMalloc((void **)&ls->tab, (size_t)T * 8 * sizeof(float)));
        ls->tab_cap = T;
        ls->key[0] = 0;
    }
    std::vector<float> tab((size_t)T * 8, 0.f);
    for (int k = 0; k < T; ++k) {
        const int i = T - 1 - k;
        float *row = &tab[(size_t)k * 8];
        if (cfg->sampler == 0) { row[0] = cfg->coef1[i]; row[1] = cfg->coef2[i]; row[2] = cfg->log_variance[i]; row[3] = i != 0 ? 1.f : 0.f; }
        else { row[0] = cfg->sqrt_recip_ab[i]; row[1] = cfg->sqrt_recipm1_ab[i]; row[2] = cfg->ab[i]; row[3] = cfg->ab_prev[i]; row[4] = i != 0 ? 1.f : 0.f; }
    }
    LoopParams lp{noise, traj, T};
    HIP_TRY(hipMemcpyAsync(ls->tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ls->params, &lp, sizeof(lp), hipMemcpyHostToDevice, st));
    LoopFuse *lf_dev = reinterpret_cast<LoopFuse *>(static_cast<char *>(ls->params) + sizeof(LoopParams));
    const LoopFuse lf{ls->tab, (const LoopParams *)ls->params, ls->x, ls->step_ctr, ls->step_ctr + 1, cfg->sampler, cfg->clip_denoised, cfg->eta};
    HIP_TRY(hipMemcpyAsync(lf_dev, &lf, sizeof(lf), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(ls->step_ctr, 0, 2 * sizeof(int), st));
    HIP_TRY(hipMemcpyAsync(ls->x, noise, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));        // host staging buffers above go out of scope; also quiesces before capture
    // ---- (re)capture one iteration if the shape of the step changed ---------------------------
    float eta_bits_f = cfg->eta;
    long eta_bits = 0;
    memcpy(&eta_bits, &eta_bits_f, sizeof(float));
    // the workspace generation covers every buffer whose address the graph bakes in (activation arena, embedding
    // table, precision-dependent kernel choice): a reallocation anywhere re-captures instead of replaying stale pointers
    const long key[6] = {unet_workspace_generation(u), B, L, cfg->sampler, cfg->clip_denoised, eta_bits};
    if (memcmp(key, ls->key, sizeof(key)) != 0) {
        if (ls->exec) { (void)hipGraphExecDestroy(ls->exec); ls->exec = nullptr; }
        if (ls->graph) { (void)hipGraphDestroy(ls->graph); ls->graph = nullptr; }
        // dry run outside capture: sizes the workspace (allocations are illegal while capturing); without the fused
        // posterior update, which would advance the state
        if ((rc = unet_forward_prepared(u, ls->x, 0, ls->x0, B, L, st, ls->step_ctr))) return rc;
        HIP_TRY(hipStreamSynchronize(st));
        bool fused = false;
        if (!ls->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&ls->cap_stream, hipStreamNonBlocking));
        hipStream_t cs = ls->cap_stream;
        HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        static const bool fuse_env = !(getenv("SURFD_LOOP_FUSE") && atoi(getenv("SURFD_LOOP_FUSE")) == 0);   // 0: separate step kernels (A/B timing)
        rc = unet_forward_prepared(u, ls->x, 0, ls->x0, B, L, cs, ls->step_ctr, fuse_env ? lf_dev : nullptr, &fused);
        if (!rc && !fused) {
            hipLaunchKernelGGL(loop_step_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(n, 256), 64)), dim3(256), 0, cs,
                               cfg->sampler, cfg->clip_denoised, cfg->eta, (const float *)ls->tab,
                               (const LoopParams *)ls->params, (const int *)ls->step_ctr, ls->x, (const float *)ls->x0, n);
            hipLaunchKernelGGL(loop_advance_kernel, dim3(1), dim3(1), 0, cs, ls->step_ctr);
        }
        hipGraph_t g = nullptr;
        hipError_t ce = hipStreamEndCapture(cs, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (ce != hipSuccess) SURFD_FAIL(SURFD_ERR_HIP, "surfd_sample_loop: stream capture failed: %s", hipGetErrorString(ce));
        ls->graph = g;
        HIP_TRY(hipGraphInstantiate(&ls->exec, g, nullptr, nullptr, 0));
        memcpy(ls->key, key, sizeof(key));
    }
    ls->run_T = T; ls->run_done = 0; ls->run_n = n;
    ls->prof_ev = prof.ev;
    prof.handed_over = true;
    return SURFD_OK;
}

int surfd_sample_loop_run(surfd_unet *u, int iterations, int *remaining, surfd_stream s) {
    if (!u || iterations < 0) SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop_run: bad argument");
    LoopState *ls = unet_loop_state(u);
    if (!ls->exec || ls->run_T <= 0) SURFD_FAIL(SURFD_ERR_STATE, "surfd_sample_loop_run: call surfd_sample_loop_begin first");
    hipStream_t st = as_stream(s);
    const int k = std::min(iterations, ls->run_T - ls->run_done);
    for (int i = 0; i < k; ++i) HIP_TRY(hipGraphLaunch(ls->exec, st));
    ls->run_done += k;
    if (remaining) *remaining = ls->run_T - ls->run_done;
    return SURFD_OK;
}

int surfd_sample_loop_end(surfd_unet *u, float *x_out, surfd_stream s) {
    if (!u || !x_out) SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop_end: null argument");
    LoopState *ls = unet_loop_state(u);
    if (ls->run_T <= 0) SURFD_FAIL(SURFD_ERR_STATE, "surfd_sample_loop_end: no loop is open");
    if (ls->run_done != ls->run_T)
        SURFD_FAIL(SURFD_ERR_STATE, "surfd_sample_loop_end: %d of %d iterations have been launched", ls->run_done, ls->run_T);
    hipStream_t st = as_stream(s);
    HIP_TRY(hipMemcpyAsync(x_out, ls->x, (size_t)ls->run_n * sizeof(float), hipMemcpyDeviceToDevice, st));
    prof_end(PROF_LOOP, (hipEvent_t)ls->prof_ev, st);
    ls->run_T = 0; ls->prof_ev = nullptr;
    return SURFD_OK;
}

int surfd_sample_loop(surfd_unet *u, const surfd_sampler_cfg *cfg, const float *noise, const float *ctx,
                      const int64_t *cls, float *x_out, float *traj, int B, int L, surfd_stream s) {
    if (!x_out) SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop: bad argument");
    int rc = surfd_sample_loop_begin(u, cfg, noise, ctx, cls, traj, B, L, s);
    if (rc) return rc;
    if ((rc = surfd_sample_loop_run(u, cfg->num_steps, nullptr, s))) return rc;
    return surfd_sample_loop_end(u, x_out, s);
}

// Iterations the fused loop of this handle has finished (the device-side counter the head convolution advances), read over a
// private stream BESIDE the stream the loop runs on: a host thread can draw the reference's progress bar
// (diffusion/gaussian_diffusion.py:677-681) while surfd_sample_loop's graph replays are in flight.  -1: no loop has run yet.
int surfd_unet_loop_progress(surfd_unet *u, int *iteration) {
    if (!u || !iteration) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_loop_progress: null argument");
    LoopState *ls = unet_loop_state(u);
    *iteration = -1;
    int *ctr = __atomic_load_n(&ls->step_ctr, __ATOMIC_ACQUIRE);
    if (!ctr || !ls->poll_stream) return SURFD_OK;       // both are made by surfd_sample_loop_begin; this call only reads them
    // HIP's current device is per host thread and defaults to 0: a fresh progress thread polling a model on device N would copy
    // from device-N memory over a device-0 context.  The handle's device is current for the copy and put back afterwards.
    int prev = -1;
    HIP_TRY(hipGetDevice(&prev));
    const int dev = unet_device(u);
    if (dev >= 0 && dev != prev) HIP_TRY(hipSetDevice(dev));
    int v = 0;
    hipError_t e = hipMemcpyAsync(&v, ctr, sizeof(int), hipMemcpyDeviceToHost, ls->poll_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ls->poll_stream);
    if (dev >= 0 && dev != prev) (void)hipSetDevice(prev);
    if (e != hipSuccess) SURFD_FAIL(SURFD_ERR_HIP, "surfd_unet_loop_progress: %s", hipGetErrorString(e));
    *iteration = v;
    return SURFD_OK;
}

}  // extern "C"
