// Standalone multi-head (cross-)attention over [b, n, c] token tensors on the fp32 matrix pipe — SURVEY.md §8 row a19:
// the LDM CrossAttention module (reference modules/attention.py:152-193): q = x Wq^T, k = ctx Wk^T, v = ctx Wv^T,
// per head softmax(q k^T * dim_head^-0.5 [masked]) v, heads merged, out = o Wo^T + b.  No Surf-D configuration
// instantiates it (use_spatial_transformer is False everywhere, models/openaimodel.py:466), so it is a separate op
// with its own handle, not a stage of the sampling loop.
//
//   xa_linear_kernel — Y[R,O] = X[R,K] W[O,K]^T (+ bias): 64 x 64 output tile per workgroup, one 32 x 32 MFMA tile per
//                      wave, both operands staged through LDS in 32-wide K chunks (rows padded: conflict-free reads)
//   xa_core_kernel   — one (sample, head, 128 queries) per workgroup, 32 queries per wave; keys/values stream through
//                      LDS in blocks of 32; scores are produced TRANSPOSED (rows = keys, columns = queries) so that the
//                      softmax runs in registers (+ one cross-half shuffle) and the probabilities are already the B
//                      operand of the value product (same idiom as attn_kernel in unet.hip); running max / sum
//                      ("online" softmax) across key blocks, so any context length works with O(1) LDS.
#include "common.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <algorithm>

namespace surfd {

__global__ __launch_bounds__(256) void xa_linear_kernel(const float *X, long ldx, const float *W, const float *bias,
                                                        float *Y, long ldy, int R, int K, int O) {
    __shared__ float As[64][33];
    __shared__ float Ws[64][33];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const long r0 = (long)blockIdx.y * 64;
    const int o0 = blockIdx.x * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, row = idx >> 5, kk = idx & 31;
            const bool kin = k0 + kk < K;
            As[row][kk] = (kin && r0 + row < R) ? X[(r0 + row) * ldx + k0 + kk] : 0.f;
            Ws[row][kk] = (kin && o0 + row < O) ? W[(long)(o0 + row) * K + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[32 * wr + col][2 * s + half], Ws[32 * wc + col][2 * s + half], acc, 0, 0, 0);
        __syncthreads();
    }
    const int o = o0 + 32 * wc + col;
    if (o < O) {
        const float bv = bias ? bias[o] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = r0 + 32 * wr + frag_row(r, lane);
            if (row < R) Y[row * ldy + o] = acc[r] + bv;
        }
    }
}

// Q [B, N, inner], K / V [B, M, inner], head h = columns [h*d, (h+1)*d); mask [B, M] bytes (0 = masked) or null
__global__ __launch_bounds__(256) void xa_core_kernel(const float *Q, const float *Kt, const float *Vt, const unsigned char *mask,
                                                      float *Oo, int N, int M, int H, int d, int inner, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ds1 = d + 1;
    float *Ks = lds, *Vs = lds + 32 * ds1;
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y / H, h = blockIdx.y - b * H;
    const int n = blockIdx.x * 128 + wave * 32 + col;             // this lane's query
    const bool n_ok = n < N;
    const int ctn = (d + 31) >> 5;                                 // 32-channel output tiles (<= 4)
    // B operand of the score product: q[n][2s + half], kept in registers for the whole key loop
    float qreg[64];
    {
        const float *qp = Q + ((long)b * N + (n_ok ? n : 0)) * inner + (long)h * d;
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const int c = 2 * s + half;
            qreg[s] = (n_ok && c < d) ? qp[c] : 0.f;
        }
    }
    f32x16 oacc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float *kb = Kt + (long)b * M * inner + (long)h * d;
    const float *vb = Vt + (long)b * M * inner + (long)h * d;
    for (int j0 = 0; j0 < M; j0 += 32) {
        __syncthreads();                                           // the previous block's readers are done
        for (int e = tid; e < 32 * d; e += 256) {
            const int j = e / d, c = e - j * d;
            const bool ok = j0 + j < M;
            Ks[j * ds1 + c] = ok ? kb[(long)(j0 + j) * inner + c] : 0.f;
            Vs[j * ds1 + c] = ok ? vb[(long)(j0 + j) * inner + c] : 0.f;
        }
        __syncthreads();
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            if (2 * s >= d) break;
            const int c = 2 * s + half;
            const float kv = c < d ? Ks[col * ds1 + c] : 0.f;       // A operand: k[key = col][channel]
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qreg[s], sc, 0, 0, 0);
        }
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + frag_row(r, lane);
            float v = sc[r] * scale;
            if (j >= M) v = -INFINITY;                              // keys beyond the context do not exist
            else if (mask && !mask[(long)b * M + j]) v = -FLT_MAX;  // masked_fill_(~mask, -finfo.max)
            sc[r] = v;
            bm = fmaxf(bm, v);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m_run, bm);                      // finite: every block holds at least one real key
        const float alpha = expf(m_run - m_new);                   // 0 on the first block
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float p = expf(sc[r] - m_new); sc[r] = p; ps += p; }
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct >= ctn) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[ct][r] *= alpha;
            const int c = 32 * ct + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float vv = c < d ? Vs[frag_row(r, lane) * ds1 + c] : 0.f;     // A operand: v[key][channel = col]
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sc[r], oacc[ct], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (n_ok) {
        float *op = Oo + ((long)b * N + n) * inner + (long)h * d;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct >= ctn) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * ct + frag_row(r, lane);
                if (c < d) op[c] = oacc[ct][r] / l_tot;
            }
        }
    }
}

}  // namespace surfd

using namespace surfd;

struct surfd_xattn {
    int qdim = 0, cdim = 0, heads = 0, dh = 0, inner = 0;
    float *wq = nullptr, *wk = nullptr, *wv = nullptr, *wo = nullptr, *bo = nullptr;
    bool have[5] = {false, false, false, false, false};
    float *ws = nullptr;
    size_t ws_floats = 0;
};

static int xa_linear(const float *X, long ldx, const float *W, const float *bias, float *Y, long ldy, long R, int K, int O, hipStream_t st) {
    if (R <= 0) return SURFD_OK;
    dim3 grid((unsigned)ceil_div(O, 64), (unsigned)ceil_div<long>(R, 64));
    hipLaunchKernelGGL(xa_linear_kernel, grid, dim3(256), 0, st, X, ldx, W, bias, Y, ldy, (int)R, K, O);
    LAUNCH_CHECK();
    return SURFD_OK;
}

extern "C" {

int surfd_xattn_create(int query_dim, int context_dim, int heads, int dim_head, surfd_xattn **out) {
    if (!out) SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_create: null out");
    if (query_dim < 1 || heads < 1 || dim_head < 1) SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_create: sizes must be positive");
    if (dim_head > 128) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "surfd_xattn_create: dim_head %d > 128 is not supported", dim_head);
    if (context_dim <= 0) context_dim = query_dim;                  // default(context_dim, query_dim), attention.py:156
    surfd_xattn *a = new surfd_xattn();
    a->qdim = query_dim; a->cdim = context_dim; a->heads = heads; a->dh = dim_head; a->inner = heads * dim_head;
    const size_t n[5] = {(size_t)a->inner * a->qdim, (size_t)a->inner * a->cdim, (size_t)a->inner * a->cdim,
                         (size_t)a->qdim * a->inner, (size_t)a->qdim};
    float **p[5] = {&a->wq, &a->wk, &a->wv, &a->wo, &a->bo};
    for (int i = 0; i < 5; ++i)
        if (hipMalloc(p[i], n[i] * sizeof(float)) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipFree(*p[j]);
            delete a;
            SURFD_FAIL(SURFD_ERR_HIP, "surfd_xattn_create: hipMalloc failed");
        }
    *out = a;
    return SURFD_OK;
}

void surfd_xattn_destroy(surfd_xattn *a) {
    if (!a) return;
    (void)hipFree(a->wq); (void)hipFree(a->wk); (void)hipFree(a->wv); (void)hipFree(a->wo); (void)hipFree(a->bo); (void)hipFree(a->ws);
    delete a;
}

// state_dict names of the reference module: to_q.weight [inner, query_dim], to_k.weight / to_v.weight [inner, context_dim],
// to_out.0.weight [query_dim, inner], to_out.0.bias [query_dim]
int surfd_xattn_set_param(surfd_xattn *a, const char *name, const float *src, const int64_t *shape, int ndim, surfd_stream s) {
    if (!a || !name || !src || !shape) SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_set_param: null argument");
    static const char *names[5] = {"to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight", "to_out.0.bias"};
    int which = -1;
    for (int i = 0; i < 5; ++i) if (!strcmp(name, names[i])) which = i;
    if (which < 0) SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_set_param: unknown parameter '%s'", name);
    const int64_t want[5][2] = {{a->inner, a->qdim}, {a->inner, a->cdim}, {a->inner, a->cdim}, {a->qdim, a->inner}, {a->qdim, 0}};
    const int wnd = which == 4 ? 1 : 2;
    if (ndim != wnd || shape[0] != want[which][0] || (wnd == 2 && shape[1] != want[which][1]))
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_set_param: wrong shape for '%s'", name);
    float *dst[5] = {a->wq, a->wk, a->wv, a->wo, a->bo};
    const size_t n = (size_t)want[which][0] * (wnd == 2 ? (size_t)want[which][1] : 1);
    HIP_TRY(hipMemcpyAsync(dst[which], src, n * sizeof(float), hipMemcpyDeviceToDevice, as_stream(s)));
    a->have[which] = true;
    return SURFD_OK;
}

// x [B, N, query_dim]; context [B, M, context_dim] or NULL (self-attention: context = x, M = N, needs
// context_dim == query_dim); mask [B, M] bytes (non-zero = attend) or NULL; out [B, N, query_dim].  All device, fp32.
int surfd_xattn_forward(surfd_xattn *a, const float *x, const float *context, const unsigned char *mask, float *out,
                        int B, int N, int M, surfd_stream s) {
    if (!a) SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_forward: null handle");
    for (int i = 0; i < 5; ++i) if (!a->have[i]) SURFD_FAIL(SURFD_ERR_STATE, "surfd_xattn_forward: parameters not loaded");
    if (B < 0 || N < 0) SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_forward: negative size");
    if (B == 0 || N == 0) return SURFD_OK;                           // empty batch: nothing to do (pointers may be null)
    if (!x || !out) SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_forward: null argument");
    int cdim = a->cdim;
    if (!context) {
        if (a->cdim != a->qdim) SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_forward: self-attention needs context_dim == query_dim");
        context = x; M = N;
    }
    if (M < 1) SURFD_FAIL(SURFD_ERR_ARG, "surfd_xattn_forward: empty context (softmax over zero keys)");
    if ((long)B * a->heads > 65535) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "surfd_xattn_forward: batch x heads > 65535");
    hipStream_t st = as_stream(s);
    const size_t nq = (size_t)B * N * a->inner, nk = (size_t)B * M * a->inner;
    const size_t need = 2 * nq + 2 * nk;
    if (need > a->ws_floats) {
        HIP_TRY(hipStreamSynchronize(st));
        (void)hipFree(a->ws); a->ws = nullptr; a->ws_floats = 0;
        HIP_TRY(hipMalloc(&a->ws, need * sizeof(float)));
        a->ws_floats = need;
    }
    float *q = a->ws, *o = q + nq, *k = o + nq, *v = k + nk;
    int rc;
    if ((rc = xa_linear(x, a->qdim, a->wq, nullptr, q, a->inner, (long)B * N, a->qdim, a->inner, st))) return rc;
    if ((rc = xa_linear(context, cdim, a->wk, nullptr, k, a->inner, (long)B * M, cdim, a->inner, st))) return rc;
    if ((rc = xa_linear(context, cdim, a->wv, nullptr, v, a->inner, (long)B * M, cdim, a->inner, st))) return rc;
    const float scale = (float)pow((double)a->dh, -0.5);             // dim_head ** -0.5, a Python float multiplied into an fp32 tensor
    const size_t lds = (size_t)2 * 32 * (a->dh + 1) * sizeof(float);
    hipLaunchKernelGGL(xa_core_kernel, dim3((unsigned)ceil_div(N, 128), (unsigned)(B * a->heads)), dim3(256), lds, st,
                       (const float *)q, (const float *)k, (const float *)v, mask, o, N, M, a->heads, a->dh, a->inner, scale);
    LAUNCH_CHECK();
    return xa_linear(o, a->inner, a->wo, a->bo, out, a->qdim, (long)B * N, a->inner, a->qdim, st);
}

}  // extern "C"
