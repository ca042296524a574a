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

int surfd_sample_loop(surfd_unet *u, const surfd_sampler_cfg *cfg, const float *noise, const float *ctx,
                      const int64_t *cls, float *x_out, float *traj, int B, int L, surfd_stream s) {
    if (!u || !cfg || !noise || !x_out || B < 1 || L < 1) SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop: bad argument");
    const int T = cfg->num_steps;
    if (T < 1 || !cfg->timestep_map) SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop: bad schedule");
    if (cfg->sampler == 0 && (!cfg->coef1 || !cfg->coef2 || !cfg->log_variance))
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop: DDPM tables missing");
    if (cfg->sampler == 1 && (!cfg->sqrt_recip_ab || !cfg->sqrt_recipm1_ab || !cfg->ab || !cfg->ab_prev))
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_sample_loop: DDIM tables missing");
    hipStream_t st = as_stream(s);
    const long n = (long)B * L;
    // every timestep-only quantity of the denoiser for all T' iterations at once: loop
    // iteration k runs original timestep timestep_map[T'-1-k] for every sample
    std::vector<int64_t> t_rows((size_t)T * B);
    for (int k = 0; k < T; ++k)
        for (int b = 0; b < B; ++b) t_rows[(size_t)k * B + b] = cfg->timestep_map[T - 1 - k];
    prof_begin(PROF_LOOP, st);
    int rc = unet_prepare_embeddings(u, t_rows.data(), T * B, ctx, cls, B, st);
    if (rc) return rc;
    // ping-pong the state inside the caller's buffers: x lives in x_out, x0 prediction in a
    // library scratch obtained from the handle (the forward writes straight into it)
    float *x0 = nullptr;
    HIP_TRY(hipMallocAsync((void **)&x0, n * sizeof(float), st));
    HIP_TRY(hipMemcpyAsync(x_out, noise, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    for (int k = 0; k < T; ++k) {
        const int i = T - 1 - k;
        if ((rc = unet_forward_prepared(u, x_out, k * B, x0, B, L, st))) break;
        const float *z = noise + (size_t)(1 + k) * n;
        float *dst = traj ? traj + (size_t)k * n : x_out;
        if (cfg->sampler == 0)
            rc = launch_ddpm(x_out, x0, z, cfg->coef1[i], cfg->coef2[i], cfg->log_variance[i], i != 0,
                             cfg->clip_denoised, dst, n, st);
        else
            rc = launch_ddim(x_out, x0, z, cfg->sqrt_recip_ab[i], cfg->sqrt_recipm1_ab[i], cfg->ab[i], cfg->ab_prev[i],
                             cfg->eta, i != 0, cfg->clip_denoised, dst, n, st);
        if (rc) break;
        if (traj) HIP_TRY(hipMemcpyAsync(x_out, dst, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    (void)hipFreeAsync(x0, st);
    prof_end(PROF_LOOP, st);
    return rc;
}

}  // extern "C"
