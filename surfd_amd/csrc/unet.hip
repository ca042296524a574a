// The latent denoiser (1-D UNet, reference models/openaimodel.py:413-749 as configured by
// models/mdm.py:34-57) on gfx950: execution plan + three kernels.
//
//   conv_kernel  — every Conv1d / Linear of the network as ONE implicit-GEMM kernel family on
//                  fp32 MFMA (v_mfma_f32_32x32x2_f32), with the producer-side elementwise work
//                  fused in: GroupNorm32 statistics + affine + SiLU on the operand while it is
//                  staged through LDS (openaimodel.py:255-275, utils/ldm_utils.py:213-230),
//                  im2col for k=3 / stride-2 (Downsample :134-160) / nearest-x2 upsample
//                  (Upsample :91-119), channel concat of the skip connection as a second
//                  K-segment (1x1 skip_connection conv, :240-241), and in the epilogue bias,
//                  the per-(t,sample) ResBlock embedding add (:264-271) and the residual add.
//   attn_kernel  — QKVAttentionLegacy (:356-372): QK^T, softmax and PV of one (sample, head, 32-query tile) per wave on
//                  fp32 MFMA with LDS-staged q/k/v; the probabilities never leave the accumulator registers.
//   temb_kernel  — timestep_embedding (utils/ldm_utils.py:165-185).
//
// Everything that depends only on (timestep, conditioning) — time_embed MLP, label/context
// embedding, the 22 ResBlock emb_layers (openaimodel.py:724-735, 218-224) — is evaluated for
// ALL loop iterations up front by the same conv_kernel (rows = steps x samples), so one
// denoiser evaluation inside the loop is 98 conv launches + 16 attention launches.
//
// Weights are streamed once per evaluation from HBM/MALL in the fragment-major packing of
// common.h (k order inside a segment: tap-major, channel-minor, channels padded to 8).
#include "common.h"
#include "unet_api.h"
#include "unet_plan.h"
#include <string.h>
#include <algorithm>
#include <stdlib.h>

namespace surfd {

// ---------------------------------------------------------------------------------------------
// conv kernel
// ---------------------------------------------------------------------------------------------
struct SegArgs {
    const float *x;        // source view (channel offset applied)
    long bstride;          // floats between consecutive batch entries
    int C, Cp;             // channels, padded to 8
    int Lin;               // source length
    int taps, stride, ups; // 1|3, 1|2, 0|1
    int gn, act;           // GroupNorm32 on this operand / SiLU on this operand
    int bmod;              // >0: source batch index = b % bmod
    const float *gamma, *beta;
    int kg_off;            // first k-group of this segment in the packed K axis
    int cc;                // channels staged per chunk (multiple of 8; whole GN groups)
    const float *add;      // Linear operands only: x[b][c] + add[step * add_step_stride + c] is what gets staged (nullable)
    long add_step_stride;
};

struct ConvArgs {
    SegArgs seg[2];
    int nseg;
    const float *wp;       // packed [ntiles][KGtot][64][4]
    int KGtot;
    const float *bias;     // [Cout] or null
    const float *emb;      // emb[b * emb_bstride + co] or null
    long emb_bstride;
    const float *res;      // res[b * res_bstride + co * Lout + l] or null
    long res_bstride;
    float *out;
    long out_bstride;
    int Cout, Lout, B;
    int log2Lout;          // Lout is a power of two
    int bchunk;            // batch entries per workgroup
    int Lsl;               // staged positions per batch entry
    int cs_max;            // LDS row stride of the largest chunk (floats)
    int red_off;           // float offset of the reduction scratch behind the slab / GN exchange area
    // cross-workgroup split-K (blockIdx.z = K slice): partial accumulators + arrival counters
    int KS;
    float *part;           // [KS][gridDim.y][gridDim.x][part_stride]
    int part_stride;       // floats per partial tile set (column tiles x 1024)
    int *counters;         // [gridDim.y][gridDim.x], zero between launches
    long long *dbg;        // optional: phase timestamps of workgroup (0,0,last slice), 16 slots
    const int *step_ptr;   // optional: device loop counter; embedding rows advance by emb_step_stride per step
    int ablate;            // developer aid (SURFD_CONV_ABLATE): 1 skip MFMA, 2 +skip GN/act, 3 +skip operand loads, 4 +skip weight loads
    long emb_step_stride;
};

constexpr int CONV_VEC_MAX = 8;   // float4 registers a thread may hold while staging (32 floats)
constexpr int CONV_U = 4;         // weight fragments fetched per software-pipeline stage

// SiLU with the hardware exp/rcp (each ~1 ulp): ~6 instructions instead of ~40 for expf + IEEE divide;
// the operand staging applies it to every element of every GroupNorm'd activation
__device__ __forceinline__ float silu(float v) { return v * __frcp_rn(1.f + __expf(-v)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding
// global load (s_waitcnt vmcnt(0)), which would serialise the weight fragments that are deliberately
// kept in flight across the staging / GroupNorm phases.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// LOG2_LV >= 0: operand rows have Lin = 4 << LOG2_LV positions (4..64), staged one (batch, channel)
//               row per thread as LV float4 loads issued back to back;
// LOG2_LV < 0 : length-1 operands (the Linear layers of the embedding path), staged as float4
//               along the channel axis.
// PARTIAL     : rows shorter than one float4 (Lin = 1 or 2 at the deepest levels of short latents).
// NTW         : column tiles (32 output positions) a wave may own: 1 for almost every launch of the
//               denoiser (<= 4 tiles per workgroup), 4 for the wide embedding GEMMs.  Keeping the
//               common case at 1 quarters the unrolled MFMA/epilogue code the instruction cache sees.
template <int LOG2_LV, bool PARTIAL = false, int NTW = 4>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs A) {
    constexpr bool LIN1 = LOG2_LV < 0;
    constexpr int LV = LIN1 ? 1 : (1 << LOG2_LV);
    constexpr int RPT = (CONV_VEC_MAX / LV) > 0 ? (CONV_VEC_MAX / LV) : 1;   // rows (or vectors, LIN1) per thread
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (A.ablate == 5) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int b0 = blockIdx.y * A.bchunk;
    const int nb = min(A.bchunk, A.B - b0);
    const int M = nb * A.Lout;
    const int nct = (M + 31) >> 5;
    // wave roles: column tiles round-robin; spare waves split K instead
    int KP = 1;
    if (nct == 1) KP = 4; else if (nct == 2) KP = 2;
    const int kpart = (KP == 1) ? 0 : wave / nct;
    const int ct0 = (KP == 1) ? wave : wave % nct;
    const int ct_step = (KP == 1) ? 4 : nct;
    const bool active = ct0 < nct;

    f32x16 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int colb[NTW], coll[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        int m = (ct0 + i * ct_step) * 32 + (lane & 31);
        if (m >= M) m = 0;
        colb[i] = m >> A.log2Lout;
        coll[i] = m & (A.Lout - 1);
    }
    const f32x4 *wbase = reinterpret_cast<const f32x4 *>(A.wp) + (size_t)tile * A.KGtot * 64 + lane;
    float *red = lds + A.red_off;   // cross-wave K reduction scratch
    const int kz = blockIdx.z;
    int chunk_id = 0;
    // Epilogue operands (bias + per-(step,sample) embedding + residual) of this wave's tile, requested at
    // kernel start when a wave owns a single tile: their round trip hides behind the whole K loop.
    const float *embp = A.emb;
    if (embp && A.step_ptr) embp += (long)(*A.step_ptr) * A.emb_step_stride;
    float pre_add[16];
    if constexpr (NTW == 1) {
        const int m = ct0 * 32 + (lane & 31);
        const bool mok = active && kpart == 0 && m < M;
        const int b = b0 + (m >> A.log2Lout), l = m & (A.Lout - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = tile * 32 + frag_row(r, lane);
            float t = 0.f;
            if (mok && co < A.Cout) {
                if (A.bias) t = A.bias[co];
                if (embp) t += embp[b * A.emb_bstride + co];
                if (A.res) t += A.res[b * A.res_bstride + (long)co * A.Lout + l];
            }
            pre_add[r] = t;
        }
    }
    // phase stamps exist only in -DSURFD_CONV_STAMPS builds: even a never-taken stamp branch makes
    // the compiler drain every in-flight load (s_waitcnt vmcnt(0)) behind it
#ifdef SURFD_CONV_STAMPS
    const bool dbg_on = A.dbg != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
#define DBG_STAMP(slot) do { if (dbg_on) A.dbg[(slot)] = (long long)clock64(); } while (0)
#else
    constexpr bool dbg_on = false;
#define DBG_STAMP(slot) do { } while (0)
#endif
    DBG_STAMP(0);

    for (int si = 0; si < A.nseg; ++si) {
        const SegArgs S = A.seg[si];
        const int pad = S.taps == 3 ? 1 : 0;
        const int gs = S.gn ? S.C / 32 : 1;
        const int kgs_per_tap = S.Cp >> 3;
        const int Lcov = S.ups ? 2 * S.Lin : S.Lin;          // slab positions covered by source data
        for (int c0 = 0; c0 < S.Cp; c0 += S.cc) {
            if ((chunk_id++) % A.KS != kz) continue;          // K slices are dealt round-robin by chunk
            if (A.ablate == 6) continue;
            const int cc = min(S.cc, S.Cp - c0);
            // Linear operands are staged with power-of-two rows (S.cc) even for a shorter last chunk:
            // the surplus channels are zero-filled and never multiplied (nkg below uses cc)
            const int cs = (LIN1 ? S.cc : cc) + 4;
            // Weight fragments of this chunk.  Each wave owns a contiguous range of the (tap, 8-channel
            // group) iteration space; three groups of CONV_U fragments (24 KB per wave) are requested
            // before the operand is staged, so the HBM/MALL latency overlaps staging + GroupNorm.
            const int nkg = cc >> 3;
            const int iters = S.taps * nkg;
            const int log2kp = (KP == 4) ? 2 : (KP == 2 ? 1 : 0);
            const int per = (iters + KP - 1) >> log2kp;
            const int it_beg = kpart * per;
            const int it_end = min(iters, it_beg + per);
            const int ngroups = (max(it_end - it_beg, 0) + CONV_U - 1) / CONV_U;
            const int wk0 = S.kg_off + (c0 >> 3);
            f32x4 aA[CONV_U], aB[CONV_U], aC[CONV_U];
            auto load_group = [&](f32x4 (&dst)[CONV_U], int g) {
                if (g < ngroups) {
#pragma unroll
                    for (int u = 0; u < CONV_U; ++u) {
                        const int it = min(it_beg + g * CONV_U + u, it_end - 1);
                        const int tap = (it >= nkg) + (it >= 2 * nkg);
                        dst[u] = wbase[(size_t)(wk0 + tap * kgs_per_tap + (it - tap * nkg)) * 64];
                    }
                }
            };
            // requested AFTER the operand loads below: vmcnt retires in order, so waiting for the (L2-resident)
            // operand must not also wait for the (HBM-resident) weights
            auto prefetch_weights = [&]() {
                if (active && A.ablate < 4) {
                    load_group(aA, 0);
                    load_group(aB, 1);
                    load_group(aC, 2);
                }
            };
            lds_barrier();   // previous chunk's MFMA reads are done
            DBG_STAMP(1);
            if constexpr (LIN1) {
                // ---- length-1 operand: float4 along channels -------------------------------------
                const int vpr = S.cc >> 2;                    // vectors per batch row (power of two: no divisions)
                const int lvpr = 31 - __builtin_clz(vpr);
                const int nvec = nb * vpr;
                f32x4 v[RPT];
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int e = tid + 256 * i;
                    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (e < nvec) {
                        const int b = e >> lvpr, j = e & (vpr - 1);
                        int bs = b0 + b;
                        if (S.bmod) bs %= S.bmod;
                        const int cg = c0 + 4 * j;
                        if (cg + 3 < S.C) v[i] = *reinterpret_cast<const f32x4 *>(S.x + bs * S.bstride + cg);
                        else
#pragma unroll
                            for (int q = 0; q < 4; ++q) if (cg + q < S.C) v[i][q] = S.x[bs * S.bstride + cg + q];
                        if (S.add) {          // the time part of the embedding joins the sample part here (one row per loop step)
                            const float *ap = S.add + (A.step_ptr ? (long)(*A.step_ptr) * S.add_step_stride : 0L);
#pragma unroll
                            for (int q = 0; q < 4; ++q) if (cg + q < S.C) v[i][q] += ap[cg + q];
                        }
                    }
                }
                prefetch_weights();
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int e = tid + 256 * i;
                    if (e < nvec) {
                        const int b = e >> lvpr, j = e & (vpr - 1);
                        f32x4 w = v[i];
                        if (S.act)
#pragma unroll
                            for (int q = 0; q < 4; ++q) w[q] = silu(w[q]);
                        *reinterpret_cast<f32x4 *>(lds + (b * A.Lsl) * cs + 4 * j) = w;
                    }
                }
                lds_barrier();
            } else {
                // ---- thread <-> channel of the chunk (cc <= 256), register row i <-> batch entry i (nb <= RPT):
                //      no integer divisions anywhere in the staging path
                const int rows = nb * cc;
                const int c = tid;
                const int cg = c0 + c;
                const bool cok = c < cc && cg < S.C;
                f32x4 v[RPT][LV];
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const bool ok = cok && i < nb && A.ablate < 3;
                    int bs = b0 + i;
                    if (S.bmod) bs %= S.bmod;
                    if constexpr (PARTIAL) {
                        const float *src1 = S.x + bs * S.bstride + (long)cg * S.Lin;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[i][0][q] = (ok && q < S.Lin) ? src1[q] : 0.f;
                    } else {
                        const f32x4 *src = reinterpret_cast<const f32x4 *>(S.x + bs * S.bstride + (long)cg * S.Lin);
#pragma unroll
                        for (int j = 0; j < LV; ++j) v[i][j] = ok ? src[j] : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                prefetch_weights();
                if (S.gn && A.ablate < 2) {
                    // two-pass GroupNorm statistics without leaving the register file; the small
                    // exchange arrays alias the (not yet written) slab.  gamma/beta are requested
                    // now, consumed after the statistics.
                    const float ga = S.gamma[min(cg, S.C - 1)], be = S.beta[min(cg, S.C - 1)];
                    float *rowmean = lds;                 // [nb][cc]
                    float *rowm2 = lds + rows;            // [nb][cc]
                    float *gstat = lds + 2 * rows;        // [nb * ng][2]
                    const int ng = cc / gs;
                    const int gq = c / gs;                // this thread's group within the chunk
                    const float inv_len = 1.f / (float)S.Lin;
                    const float inv_cnt = 1.f / (float)(gs * S.Lin);
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        if (i < nb && c < cc) {
                            float sacc = 0.f;
#pragma unroll
                            for (int j = 0; j < LV; ++j) sacc += (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
                            const float rm = sacc * inv_len;
                            float m2 = 0.f;
#pragma unroll
                            for (int j = 0; j < LV; ++j)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float d = v[i][j][q] - rm;
                                    if (!PARTIAL || 4 * j + q < S.Lin) m2 += d * d;
                                }
                            rowmean[i * cc + c] = rm; rowm2[i * cc + c] = m2;
                        }
                    }
                    lds_barrier();
                    // group statistics: equal-size rows combine exactly (Chan et al.):
                    //   mean = avg(row means),  M2 = sum(row M2) + Lin * sum((row mean - mean)^2)
                    // one team of 8 lanes per (batch entry, group): strided partial sums + 3 shuffle steps
                    // instead of one thread walking up to 56 dependent LDS reads
                    for (int q0 = 0; q0 < nb * ng; q0 += 32) {
                        const int q = q0 + (tid >> 3), lt = tid & 7;
                        const bool qok = q < nb * ng;
                        const int off = qok ? (q / ng) * cc + (q % ng) * gs : 0;
                        float sm = 0.f;
                        if (qok) for (int j = lt; j < gs; j += 8) sm += rowmean[off + j];
                        sm += __shfl_xor(sm, 4); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 1);
                        const float gm = sm / (float)gs;
                        float m2 = 0.f;
                        if (qok) for (int j = lt; j < gs; j += 8) { const float d = rowmean[off + j] - gm; m2 += rowm2[off + j] + (float)S.Lin * (d * d); }
                        m2 += __shfl_xor(m2, 4); m2 += __shfl_xor(m2, 2); m2 += __shfl_xor(m2, 1);
                        if (qok && lt == 0) {
                            gstat[2 * q] = gm;
                            gstat[2 * q + 1] = 1.f / sqrtf(m2 * inv_cnt + 1e-5f);
                        }
                    }
                    lds_barrier();
                    float scal[RPT], gmean[RPT];
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int q = (i < nb && c < cc) ? i * ng + gq : 0;
                        gmean[i] = gstat[2 * q]; scal[i] = gstat[2 * q + 1];
                    }
                    lds_barrier();      // exchange arrays are dead: the slab may be written now
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        if (i < nb && cok) {
                            const float gsc = ga * scal[i];
#pragma unroll
                            for (int j = 0; j < LV; ++j)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    float w = (v[i][j][q] - gmean[i]) * gsc + be;
                                    if (S.act) w = silu(w);
                                    v[i][j][q] = w;
                                }
                        }
                    }
                } else if (S.act && A.ablate < 2) {
#pragma unroll
                    for (int i = 0; i < RPT; ++i)
#pragma unroll
                        for (int j = 0; j < LV; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[i][j][q] = silu(v[i][j][q]);
                }
                DBG_STAMP(3);
                // ---- write the slab, transposed to [b][position][channel]; zero the halo positions -----------
                if (c < cc) {
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        if (i < nb) {
                            float *dst = lds + (i * A.Lsl + pad) * cs + c;
#pragma unroll
                            for (int j = 0; j < LV; ++j)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int l = 4 * j + q;
                                    if (PARTIAL && l >= S.Lin) continue;
                                    if (S.ups) { dst[(2 * l) * cs] = v[i][j][q]; dst[(2 * l + 1) * cs] = v[i][j][q]; }
                                    else dst[l * cs] = v[i][j][q];
                                }
                            float *row0 = lds + (i * A.Lsl) * cs + c;
                            for (int p = 0; p < pad; ++p) row0[p * cs] = 0.f;
                            for (int p = pad + Lcov; p < A.Lsl; ++p) row0[p * cs] = 0.f;
                        }
                    }
                }
                lds_barrier();
            }
            DBG_STAMP(4);
            // ---- MFMA over (tap, 8-channel group); weight fragments prefetched one stage ahead ------------
            if (active && A.ablate < 1) {
                int lbase[NTW];       // per-lane LDS offset of each column tile's operand rows
#pragma unroll
                for (int i = 0; i < NTW; ++i) lbase[i] = (colb[i] * A.Lsl + coll[i] * S.stride) * cs + 4 * (lane >> 5);
                auto operand = [&](int i, int it) -> f32x4 {
                    const int tap = (it >= nkg) + (it >= 2 * nkg);
                    return *reinterpret_cast<const f32x4 *>(lds + lbase[i] + tap * cs + (it - tap * nkg) * 8);
                };
                auto compute_group = [&](const f32x4 (&a)[CONV_U], int g) {
                    if (g >= ngroups) return;
                    const int it0 = it_beg + g * CONV_U;
                    // operand fragments are fetched one iteration ahead of the MFMAs that consume them
                    f32x4 bq[NTW], bn[NTW];
#pragma unroll
                    for (int i = 0; i < NTW; ++i)
                        if (ct0 + i * ct_step < nct) bq[i] = operand(i, min(it0, it_end - 1));
#pragma unroll
                    for (int u = 0; u < CONV_U; ++u) {
                        const int it = it0 + u;
                        const int itn = min(it + 1, it_end - 1);
#pragma unroll
                        for (int i = 0; i < NTW; ++i)
                            if (ct0 + i * ct_step < nct) bn[i] = operand(i, itn);
                        if (it < it_end) {
#pragma unroll
                            for (int i = 0; i < NTW; ++i) {
                                if (ct0 + i * ct_step < nct) {
#pragma unroll
                                    for (int q = 0; q < 4; ++q)
                                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q], bq[i][q], acc[i], 0, 0, 0);
                                }
                            }
                        }
#pragma unroll
                        for (int i = 0; i < NTW; ++i) bq[i] = bn[i];
                    }
                };
                for (int g = 0; g < ngroups; g += 3) {
                    compute_group(aA, g);
                    load_group(aA, g + 3);
                    compute_group(aB, g + 1);
                    load_group(aB, g + 4);
                    compute_group(aC, g + 2);
                    load_group(aC, g + 5);
                }
            }
        }
    }
    DBG_STAMP(5);
    // ---- cross-wave K reduction (only when spare waves split K) --------------------------------
    if (KP > 1) {
        lds_barrier();
        if (active && kpart > 0) {
            float *dst = red + ((size_t)(kpart - 1) * nct + ct0) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r * 64 + lane] = acc[0][r];
        }
        lds_barrier();
        if (active && kpart == 0) {
            for (int kp = 1; kp < KP; ++kp) {
                const float *srcp = red + ((size_t)(kp - 1) * nct + ct0) * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += srcp[r * 64 + lane];
            }
        }
    }
    DBG_STAMP(6);
    // ---- cross-workgroup K reduction: every slice publishes its partial tiles; the last arriver
    //      sums them in slice order (deterministic) and runs the epilogue.  Publication follows the
    //      agent-scope hand-off recipe R1 (cdna_hip_programming.md §6 G16): write-through (sc1) stores ->
    //      vmcnt(0) in every storing wave -> barrier -> one-lane relaxed ticket; last arriver: acquire.
    if (A.KS > 1) {
        const size_t slot = ((size_t)blockIdx.y * gridDim.x + tile);
        float *mine = A.part + (((size_t)kz * gridDim.y + blockIdx.y) * gridDim.x + tile) * A.part_stride;
        if (active && kpart == 0) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int ct = ct0 + i * ct_step;
                if (ct < nct) {
                    // write-through (sc1) 16-byte stores: the partial tile leaves this XCD's L2 as it is
                    // written, so no buffer_wbl2 release fence is needed before the ticket
                    // (MI355X_MICROARCH.md "publish-large": 3.0 us vs 8.2 us)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 val = {acc[i][4 * r4], acc[i][4 * r4 + 1], acc[i][4 * r4 + 2], acc[i][4 * r4 + 3]};
                        float *dst = mine + ((size_t)(ct * 4 + r4) * 64 + lane) * 4;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(val) : "memory");
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *flag = reinterpret_cast<int *>(red + 6 * 1024 - 4);
        if (tid == 0) {
            const int prev = __hip_atomic_fetch_add(A.counters + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (prev == A.KS - 1) ? 1 : 0;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(A.counters + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            *flag = last;
        }
        __syncthreads();
        DBG_STAMP(7);
        if (*flag == 0) return;
        if (active && kpart == 0) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int ct = ct0 + i * ct_step;
                if (ct >= nct) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
                for (int z = 0; z < A.KS; ++z) {
                    const float *src = A.part + (((size_t)z * gridDim.y + blockIdx.y) * gridDim.x + tile) * A.part_stride;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 pv = *reinterpret_cast<const f32x4 *>(src + ((size_t)(ct * 4 + r4) * 64 + lane) * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[i][4 * r4 + q] += pv[q];
                    }
                }
            }
        }
    }
    DBG_STAMP(8);
    // ---- epilogue: bias + embedding + residual, coalesced along l ------------------------------------
    if (active && kpart == 0) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int ct = ct0 + i * ct_step;
            if (ct >= nct) continue;
            const int m = ct * 32 + (lane & 31);
            const bool mok = m < M;
            const int b = b0 + (m >> A.log2Lout), l = m & (A.Lout - 1);
            // all epilogue operands are requested first (one memory round trip), then combined and stored
            float add[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (NTW == 1) {
                    add[r] = pre_add[r];
                } else {
                    const int co = tile * 32 + frag_row(r, lane);
                    const bool ok = mok && co < A.Cout;
                    float t = 0.f;
                    if (ok) {
                        if (A.bias) t = A.bias[co];
                        if (embp) t += embp[b * A.emb_bstride + co];
                        if (A.res) t += A.res[b * A.res_bstride + (long)co * A.Lout + l];
                    }
                    add[r] = t;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = tile * 32 + frag_row(r, lane);
                if (mok && co < A.Cout) A.out[b * A.out_bstride + (long)co * A.Lout + l] = acc[i][r] + add[r];
            }
        }
    }
    if (dbg_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); DBG_STAMP(9); }
#undef DBG_STAMP
}

// ---------------------------------------------------------------------------------------------
// attention core on the matrix pipe: one workgroup per (sample, head, 32-query tile)
// ---------------------------------------------------------------------------------------------
// QKVAttentionLegacy (openaimodel.py:356-372): w = softmax_s((q * d^-1/4)^T (k * d^-1/4)), a = v w^T, per head, T <= 64,
// d = C / 8 in {28, 56, 112}.  q, k, v of the head are staged in LDS as [channel][T]; all three contractions run on
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation):
//   S^T[s][t] = sum_c k[c][s] q[c][t]     keys along the accumulator ROWS, queries along the columns: a query's 32 (or 64)
//                                          scores then live in the 16 (32) accumulator registers of two lanes, so the
//                                          softmax over keys is register-local plus ONE cross-lane exchange;
//   a[c][t]   = sum_s v[c][s] P[s][t]      the probabilities are consumed as the MFMA B operand straight from those
//                                          registers: k-step r contracts keys frag_row(r, 0) and frag_row(r, 1), which is
//                                          exactly what lane halves 0 / 1 hold in register r — P never touches LDS.
__global__ __launch_bounds__(256) void attn_kernel(const float *qkv, long qkv_bstride, float *out, long out_bstride,
                                                   int heads, int d, int T, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int qt_n = (T + 31) >> 5;                       // query tiles = key tiles (1, or 2 when T = 64)
    const int bh = blockIdx.x / qt_n, qt = blockIdx.x - bh * qt_n;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Tp = T + 1;                                 // row stride: conflict-free column reads for the P*V operand
    float *q = lds, *k = q + d * Tp, *v = k + d * Tp;
    float *xch = v + d * Tp;                              // [4 waves][qt_n * 16 registers][64 lanes]: partial scores
    // ---- stage the head's q, k, v: all four waves, float4 along the sequence (scalar for odd tiny sequences) -------
    if (T & 3) {
        const float *src = qkv + b * qkv_bstride + (long)h * 3 * d * T;
        for (int e = tid; e < 3 * d * T; e += 256) {
            const int c = e / T, t = e - c * T;
            lds[c * Tp + t] = c < 2 * d ? src[e] * scale : src[e];
        }
    } else {
        const f32x4 *src4 = reinterpret_cast<const f32x4 *>(qkv + b * qkv_bstride + (long)h * 3 * d * T);   // head-major [q | k | v]
        const int vpr = T >> 2, n4 = 3 * d * vpr;
        for (int e = tid; e < n4; e += 256) {
            const int c = e / vpr, t4 = e - c * vpr;
            f32x4 x = src4[e];
            if (c < 2 * d) { x[0] *= scale; x[1] *= scale; x[2] *= scale; x[3] *= scale; }   // (q * scale), (k * scale) as the reference forms them
            float *dstp = lds + c * Tp + 4 * t4;
            dstp[0] = x[0]; dstp[1] = x[1]; dstp[2] = x[2]; dstp[3] = x[3];
        }
    }
    __syncthreads();
    const int t_glob = qt * 32 + col;                     // this lane's query
    const bool t_ok = t_glob < T;
    f32x16 sc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
    // ---- scores, transposed (rows = keys, columns = queries); the channel contraction is split over the waves ----
    const int cper = ((d / 2 + 3) / 4) * 2;               // channels per wave (even)
    const int c_beg = wave * cper, c_end = min(d, c_beg + cper);
    for (int c = c_beg; c < c_end; c += 2) {
        const float qv = t_ok ? q[(c + half) * Tp + t_glob] : 0.f;                 // B operand: q[c + half][t]
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
            if (kt < qt_n) {
                const int s_ = kt * 32 + col;
                const float kv = s_ < T ? k[(c + half) * Tp + s_] : 0.f;         // A operand: k[c + half][s]
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qv, sc[kt], 0, 0, 0);
            }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
        if (kt < qt_n)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[((wave * qt_n + kt) * 16 + r) * 64 + lane] = sc[kt][r];
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
        if (kt < qt_n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) t += xch[((w * qt_n + kt) * 16 + r) * 64 + lane];       // fixed order: every wave gets the same bits
                sc[kt][r] = t;
            }
    // ---- softmax over keys: registers of this lane + the partner lane (the other half of the rows) ---------------
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
        if (kt < qt_n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int s_ = kt * 32 + frag_row(r, lane);
                if (s_ >= T) sc[kt][r] = -INFINITY;          // keys beyond the sequence do not exist
                mx = fmaxf(mx, sc[kt][r]);
            }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
        if (kt < qt_n)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float ex = expf(sc[kt][r] - mx); sc[kt][r] = ex; sum += ex; }
    sum += __shfl_xor(sum, 32);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[kt][r] = sc[kt][r] / sum;
    // ---- a = v P^T: one 32-channel tile per wave, probabilities from registers ---------------------------------
    float *dst = out + b * out_bstride + (long)h * d * T;
    for (int c0 = wave * 32; c0 < d; c0 += 128) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int c = c0 + col;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
            if (kt < qt_n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int s_ = kt * 32 + frag_row(r, lane);                  // keys frag_row(r, 0) / frag_row(r, 1)
                    const float vv = (c < d && s_ < T) ? v[c * Tp + s_] : 0.f;    // A operand: v[c][s]
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sc[kt][r], acc, 0, 0, 0);
                }
        if (t_ok)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = c0 + frag_row(r, lane);
                if (cr < d) dst[(long)cr * T + t_glob] = acc[r];
            }
    }
}

// timestep_embedding: out[r][0:half] = cos(t*f_k), out[r][half:] = sin(t*f_k), f_k = exp(-ln(1e4) k / half)
__global__ void temb_kernel(const int64_t *t, int rows, int dim, float *out) {
    const int half = dim / 2;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rows * half; e += gridDim.x * blockDim.x) {
        const int r = e / half, kk = e % half;
        const float freq = expf(__fdiv_rn(__fmul_rn(-9.210340371976184f, (float)kk), (float)half));
        const float a = __fmul_rn((float)t[r], freq);
        out[(long)r * dim + kk] = cosf(a);
        out[(long)r * dim + half + kk] = sinf(a);
    }
}

__global__ void add_label_kernel(float *emb, int rows, int dim, const float *table, const int64_t *cls, int B) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < (long)rows * dim; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / dim), c = (int)(e % dim);
        emb[e] += table[cls[r % B] * dim + c];
    }
}

__global__ void vec_add_kernel(float *dst, const float *a, const float *b, int n) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) dst[e] = a[e] + (b ? b[e] : 0.f);
}

}  // namespace surfd

// =============================================================================================
// host: plan
// =============================================================================================
using namespace surfd;

namespace {

void add_param(surfd_unet *u, const std::string &k, std::vector<int64_t> shape) {
    u->pindex[k] = (int)u->params.size();
    u->params.push_back({k, std::move(shape)});
}

int new_buf(surfd_unet *u, int C, int ds) { u->bufs.push_back({C, ds}); return (int)u->bufs.size() - 1; }

void add_res_params(surfd_unet *u, const std::string &p, int cin, int cout) {
    const int64_t ted = u->ted;
    add_param(u, p + ".in_layers.0.weight", {cin}); add_param(u, p + ".in_layers.0.bias", {cin});
    add_param(u, p + ".in_layers.2.weight", {cout, cin, 3}); add_param(u, p + ".in_layers.2.bias", {cout});
    add_param(u, p + ".emb_layers.1.weight", {cout, ted}); add_param(u, p + ".emb_layers.1.bias", {cout});
    add_param(u, p + ".out_layers.0.weight", {cout}); add_param(u, p + ".out_layers.0.bias", {cout});
    add_param(u, p + ".out_layers.3.weight", {cout, cout, 3}); add_param(u, p + ".out_layers.3.bias", {cout});
    if (cin != cout) { add_param(u, p + ".skip_connection.weight", {cout, cin, 1}); add_param(u, p + ".skip_connection.bias", {cout}); }
}

void add_attn_params(surfd_unet *u, const std::string &p, int c) {
    add_param(u, p + ".norm.weight", {c}); add_param(u, p + ".norm.bias", {c});
    add_param(u, p + ".qkv.weight", {3 * c, c, 1}); add_param(u, p + ".qkv.bias", {3 * c});
    add_param(u, p + ".proj_out.weight", {c, c, 1}); add_param(u, p + ".proj_out.bias", {c});
}

// ResBlock: dst <- skip(src) + conv2(silu(gn(conv1(silu(gn(src))) + emb)))
void emit_res(surfd_unet *u, const std::string &p, View src, int cin, int cout, int ds, int scratch1, View dst) {
    Op a; a.kind = 0;
    ConvPlan &c1 = a.conv;
    c1.nseg = 1; c1.Cout = cout; c1.ds_out = ds;
    c1.seg[0].src = src; c1.seg[0].C = cin; c1.seg[0].taps = 3; c1.seg[0].gn = 1; c1.seg[0].act = 1; c1.seg[0].ds = ds;
    c1.seg[0].wkey = p + ".in_layers.2.weight"; c1.seg[0].gnkey = p + ".in_layers.0";
    c1.bias_keys = {p + ".in_layers.2.bias"};
    c1.emb_off = u->emb_total;
    u->emb_layers.push_back({p + ".emb_layers.1", cout});
    u->emb_total += cout;
    c1.dst = View{scratch1, 0};
    u->ops.push_back(a);
    Op b; b.kind = 0;
    ConvPlan &c2 = b.conv;
    c2.Cout = cout; c2.ds_out = ds;
    c2.seg[0].src = View{scratch1, 0}; c2.seg[0].C = cout; c2.seg[0].taps = 3; c2.seg[0].gn = 1; c2.seg[0].act = 1; c2.seg[0].ds = ds;
    c2.seg[0].wkey = p + ".out_layers.3.weight"; c2.seg[0].gnkey = p + ".out_layers.0";
    c2.bias_keys = {p + ".out_layers.3.bias"};
    if (cin != cout) {
        c2.nseg = 2;
        c2.seg[1].src = src; c2.seg[1].C = cin; c2.seg[1].taps = 1; c2.seg[1].ds = ds;
        c2.seg[1].wkey = p + ".skip_connection.weight";
        c2.bias_keys.push_back(p + ".skip_connection.bias");
    } else {
        c2.nseg = 1;
        c2.res = src;
    }
    c2.dst = dst;
    u->ops.push_back(b);
}

// AttentionBlock: dst <- src + proj(attn(qkv(gn(src))))
void emit_attn(surfd_unet *u, const std::string &p, View src, int c, int ds, int qkv_buf, int att_buf, View dst) {
    Op a; a.kind = 0;
    ConvPlan &q = a.conv;
    q.Cout = 3 * c; q.ds_out = ds;
    q.seg[0].src = src; q.seg[0].C = c; q.seg[0].taps = 1; q.seg[0].gn = 1; q.seg[0].act = 0; q.seg[0].ds = ds;
    q.seg[0].wkey = p + ".qkv.weight"; q.seg[0].gnkey = p + ".norm";
    q.bias_keys = {p + ".qkv.bias"};
    q.dst = View{qkv_buf, 0};
    u->ops.push_back(a);
    Op m; m.kind = 1;
    m.attn.qkv = View{qkv_buf, 0}; m.attn.out = View{att_buf, 0}; m.attn.C = c; m.attn.ds = ds;
    u->ops.push_back(m);
    Op b; b.kind = 0;
    ConvPlan &pr = b.conv;
    pr.Cout = c; pr.ds_out = ds;
    pr.seg[0].src = View{att_buf, 0}; pr.seg[0].C = c; pr.seg[0].taps = 1; pr.seg[0].ds = ds;
    pr.seg[0].wkey = p + ".proj_out.weight";
    pr.bias_keys = {p + ".proj_out.bias"};
    pr.res = src; pr.dst = dst;
    u->ops.push_back(b);
}

void emit_plain_conv(surfd_unet *u, const std::string &wprefix, View src, int cin, int cout, int ds_src, int ds_out,
                     int stride, int ups, View dst) {
    Op a; a.kind = 0;
    ConvPlan &c = a.conv;
    c.Cout = cout; c.ds_out = ds_out;
    c.seg[0].src = src; c.seg[0].C = cin; c.seg[0].taps = 3; c.seg[0].stride = stride; c.seg[0].ups = ups; c.seg[0].ds = ds_src;
    c.seg[0].wkey = wprefix + ".weight";
    c.bias_keys = {wprefix + ".bias"};
    c.dst = dst;
    u->ops.push_back(a);
}

bool in_attn(const surfd_unet_cfg &c, int ds) {
    for (int i = 0; i < c.n_attn; ++i) if (c.attention_resolutions[i] == ds) return true;
    return false;
}

// Mirrors UNetModel.__init__ (openaimodel.py:516-686): parameters in registration order and
// the op list.  Skip tensors are produced directly inside the concat buffer of the output
// block that consumes them (zero-copy torch.cat, :742-744).
int build_plan(surfd_unet *u) {
    const surfd_unet_cfg &c = u->cfg;
    const int mc = c.model_channels;
    u->ted = 4 * mc;
    const int64_t ted = u->ted;
    add_param(u, "time_embed.0.weight", {ted, mc}); add_param(u, "time_embed.0.bias", {ted});
    add_param(u, "time_embed.2.weight", {ted, ted}); add_param(u, "time_embed.2.bias", {ted});
    if (c.num_classes > 0) add_param(u, "label_emb.weight", {c.num_classes, ted});
    if (c.context_dim > 0) { add_param(u, "sketch_emb.weight", {ted, c.context_dim}); add_param(u, "sketch_emb.bias", {ted}); }

    // ---- structure pass ----
    struct InBlk { std::string prefix; int kind; int cin, cout, ds_in, ds_out; bool attn; };   // kind 0 conv,1 res,2 down
    std::vector<InBlk> in;
    in.push_back({"input_blocks.0", 0, c.in_channels, mc, 1, 1, false});
    std::vector<int> chans{mc};
    std::vector<int> chan_ds{1};
    int ch = mc, ds = 1;
    for (int level = 0; level < c.n_mult; ++level) {
        for (int r = 0; r < c.num_res_blocks; ++r) {
            const int co = c.channel_mult[level] * mc;
            in.push_back({"input_blocks." + std::to_string(in.size()), 1, ch, co, ds, ds, in_attn(c, ds)});
            ch = co; chans.push_back(ch); chan_ds.push_back(ds);
        }
        if (level != c.n_mult - 1) {
            in.push_back({"input_blocks." + std::to_string(in.size()), 2, ch, ch, ds, ds * 2, false});
            ds *= 2; chans.push_back(ch); chan_ds.push_back(ds);
        }
    }
    const int mid_ch = ch, mid_ds = ds;
    struct OutBlk { std::string prefix; int ch_h, ich, cout, ds; bool attn, up; };
    std::vector<OutBlk> out;
    {
        std::vector<int> st = chans;
        int ch2 = ch, ds2 = ds;
        for (int level = c.n_mult - 1; level >= 0; --level) {
            for (int i = 0; i <= c.num_res_blocks; ++i) {
                const int ich = st.back(); st.pop_back();
                const int co = mc * c.channel_mult[level];
                const bool up = level && i == c.num_res_blocks;
                out.push_back({"output_blocks." + std::to_string(out.size()), ch2, ich, co, ds2, in_attn(c, ds2), up});
                ch2 = co;
                if (up) ds2 /= 2;
            }
        }
    }
    const int nin = (int)in.size(), nout = (int)out.size();
    if (nin != nout) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "unet plan: %d input blocks vs %d output blocks", nin, nout);
    // ---- buffers ----
    std::vector<int> cat(nout);
    int SC = mc;              // widest activation any scratch buffer has to hold
    for (int k = 0; k < nout; ++k) {
        cat[k] = new_buf(u, out[k].ch_h + out[k].ich, out[k].ds);
        SC = std::max(SC, out[k].cout);
    }
    for (auto &b : in) SC = std::max(SC, b.cout);
    // scratch buffers: SC channels at full length (ds = 1) — large enough at every level; a
    // view's batch stride is (buffer channels) x (length at the level it is used at)
    const int s1 = new_buf(u, SC, 1), s2 = new_buf(u, SC, 1), s5 = new_buf(u, SC, 1);
    const int sq = new_buf(u, 3 * SC, 1), sa = new_buf(u, SC, 1);
    const int fin = new_buf(u, mc, 1);
    auto hs_view = [&](int j) { const int k = nout - 1 - j; return View{cat[k], out[k].ch_h}; };

    // ---- input blocks ----
    for (int i = 0; i < nin; ++i) {
        const InBlk &b = in[i];
        const View src = (i == 0) ? View{-2, 0} : hs_view(i - 1);
        const View dst = hs_view(i);
        const std::string p = b.prefix;
        if (b.kind == 0) {
            add_param(u, p + ".0.weight", {b.cout, b.cin, 3}); add_param(u, p + ".0.bias", {b.cout});
            emit_plain_conv(u, p + ".0", src, b.cin, b.cout, 1, 1, 1, 0, dst);
        } else if (b.kind == 1) {
            add_res_params(u, p + ".0", b.cin, b.cout);
            if (b.attn) add_attn_params(u, p + ".1", b.cout);
            emit_res(u, p + ".0", src, b.cin, b.cout, b.ds_out, s1, b.attn ? View{s2, 0} : dst);
            if (b.attn) emit_attn(u, p + ".1", View{s2, 0}, b.cout, b.ds_out, sq, sa, dst);
        } else {
            add_param(u, p + ".0.op.weight", {b.cout, b.cin, 3}); add_param(u, p + ".0.op.bias", {b.cout});
            emit_plain_conv(u, p + ".0.op", src, b.cin, b.cout, b.ds_in, b.ds_out, 2, 0, dst);
        }
    }
    // ---- middle ----
    {
        add_res_params(u, "middle_block.0", mid_ch, mid_ch);
        add_attn_params(u, "middle_block.1", mid_ch);
        add_res_params(u, "middle_block.2", mid_ch, mid_ch);
        emit_res(u, "middle_block.0", hs_view(nin - 1), mid_ch, mid_ch, mid_ds, s1, View{s2, 0});
        emit_attn(u, "middle_block.1", View{s2, 0}, mid_ch, mid_ds, sq, sa, View{s5, 0});
        emit_res(u, "middle_block.2", View{s5, 0}, mid_ch, mid_ch, mid_ds, s1, View{cat[0], 0});
    }
    // ---- output blocks ----
    for (int k = 0; k < nout; ++k) {
        const OutBlk &b = out[k];
        const std::string p = b.prefix;
        const View src{cat[k], 0};
        const View dst = (k + 1 < nout) ? View{cat[k + 1], 0} : View{fin, 0};
        const int cin = b.ch_h + b.ich;
        add_res_params(u, p + ".0", cin, b.cout);
        int j = 1;
        if (b.attn) { add_attn_params(u, p + "." + std::to_string(j), b.cout); ++j; }
        if (b.up) { add_param(u, p + "." + std::to_string(j) + ".conv.weight", {b.cout, b.cout, 3}); add_param(u, p + "." + std::to_string(j) + ".conv.bias", {b.cout}); }
        View cur = (b.attn || b.up) ? View{s2, 0} : dst;
        emit_res(u, p + ".0", src, cin, b.cout, b.ds, s1, cur);
        j = 1;
        if (b.attn) {
            const View nxt = b.up ? View{s5, 0} : dst;
            emit_attn(u, p + "." + std::to_string(j), cur, b.cout, b.ds, sq, sa, nxt);
            cur = nxt; ++j;
        }
        if (b.up) emit_plain_conv(u, p + "." + std::to_string(j) + ".conv", cur, b.cout, b.cout, b.ds, b.ds / 2, 1, 1, dst);
    }
    // ---- head ----
    add_param(u, "out.0.weight", {mc}); add_param(u, "out.0.bias", {mc});
    add_param(u, "out.2.weight", {c.out_channels, mc, 3}); add_param(u, "out.2.bias", {c.out_channels});
    {
        Op a; a.kind = 0;
        ConvPlan &h = a.conv;
        h.Cout = c.out_channels; h.ds_out = 1;
        h.seg[0].src = View{fin, 0}; h.seg[0].C = mc; h.seg[0].taps = 3; h.seg[0].gn = 1; h.seg[0].act = 1; h.seg[0].ds = 1;
        h.seg[0].wkey = "out.2.weight"; h.seg[0].gnkey = "out.0";
        h.bias_keys = {"out.2.bias"};
        h.dst = View{-3, 0};
        u->ops.push_back(a);
    }
    // ---- embedding path (length-1 operands: ds = 0) ----
    u->lin1.Cout = (int)ted; u->lin1.seg[0].C = mc; u->lin1.seg[0].ds = 0; u->lin1.seg[0].wkey = "time_embed.0.weight";
    u->lin1.bias_keys = {"time_embed.0.bias"};
    u->lin2.Cout = (int)ted; u->lin2.seg[0].C = (int)ted; u->lin2.seg[0].act = 1; u->lin2.seg[0].ds = 0; u->lin2.seg[0].wkey = "time_embed.2.weight";
    u->lin2.bias_keys = {"time_embed.2.bias"};
    if (c.context_dim > 0) {
        u->lin2.nseg = 2;
        u->lin2.seg[1].C = c.context_dim; u->lin2.seg[1].ds = 0; u->lin2.seg[1].wkey = "sketch_emb.weight";
        u->lin2.bias_keys.push_back("sketch_emb.bias");
    }
    u->lin3.Cout = u->emb_total; u->lin3.seg[0].C = (int)ted; u->lin3.seg[0].act = 1; u->lin3.seg[0].ds = 0;
    return SURFD_OK;
}

int seg_kgroups(const SegPlan &s) { return s.taps * (ceil_div(s.C, 8)); }

}  // namespace

// =============================================================================================
// host: device state
// =============================================================================================
namespace {

int unet_alloc(surfd_unet *u) {
    if (u->allocated) return SURFD_OK;
    HIP_TRY(hipGetDevice(&u->device));
    // ---- packed-weight arena layout ----
    size_t off = 0;
    auto place = [&](ConvPlan &c) {
        int kg = 0;
        for (int s = 0; s < c.nseg; ++s) { c.kg_off[s] = kg; kg += seg_kgroups(c.seg[s]); }
        c.KGtot = kg;
        c.w_off = off;
        off += (size_t)ceil_div(c.Cout, 32) * kg * 256;
    };
    for (auto &op : u->ops) if (op.kind == 0) place(op.conv);
    place(u->lin1); place(u->lin2); place(u->lin3);
    u->wpack_floats = off;
    { int rc2 = conv2_plan_layout(u); if (rc2) return rc2; }
    { int rc2 = conv2_set_attributes(); if (rc2) return rc2; }
    if (const char *pe = getenv("SURFD_UNET_PRECISION")) u->precision = !strcmp(pe, "fp32") ? 0 : 1;
    HIP_TRY(hipMalloc((void **)&u->wpack, off * sizeof(float)));
    HIP_TRY(hipMemset(u->wpack, 0, off * sizeof(float)));
    // ---- vectors: every 1-D parameter raw, then combined biases ----
    size_t voff = 0;
    for (auto &p : u->params)
        if (p.shape.size() == 1) { u->vec_off[p.key] = voff; voff += (size_t)ceil_div<int64_t>(p.shape[0], 4) * 4; }
    auto place_bias = [&](ConvPlan &c) { c.bias_off = voff; voff += (size_t)ceil_div(c.Cout, 4) * 4; };
    for (auto &op : u->ops) if (op.kind == 0) place_bias(op.conv);
    place_bias(u->lin1); place_bias(u->lin2); place_bias(u->lin3);
    u->vec_floats = voff;
    HIP_TRY(hipMalloc((void **)&u->vecs, voff * sizeof(float)));
    HIP_TRY(hipMemset(u->vecs, 0, voff * sizeof(float)));
    if (u->cfg.num_classes > 0) HIP_TRY(hipMalloc((void **)&u->label_table, (size_t)u->cfg.num_classes * u->ted * sizeof(float)));
    const int max_lds = 160 * 1024;
#define SURFD_SET_LDS(K) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds))
    SURFD_SET_LDS((conv_kernel<-1, false, 1>)); SURFD_SET_LDS((conv_kernel<-1, false, 4>));
    SURFD_SET_LDS((conv_kernel<0, false, 1>));  SURFD_SET_LDS((conv_kernel<0, false, 4>));
    SURFD_SET_LDS((conv_kernel<0, true, 1>));   SURFD_SET_LDS((conv_kernel<0, true, 4>));
    SURFD_SET_LDS((conv_kernel<1, false, 1>));  SURFD_SET_LDS((conv_kernel<1, false, 4>));
    SURFD_SET_LDS((conv_kernel<2, false, 1>));  SURFD_SET_LDS((conv_kernel<2, false, 4>));
    SURFD_SET_LDS((conv_kernel<3, false, 1>));  SURFD_SET_LDS((conv_kernel<3, false, 4>));
    SURFD_SET_LDS((conv_kernel<4, false, 1>));  SURFD_SET_LDS((conv_kernel<4, false, 4>));
#undef SURFD_SET_LDS
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    u->part_floats = (size_t)16 << 20;                      // 64 MB of partial tiles
    HIP_TRY(hipMalloc((void **)&u->part, u->part_floats * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&u->counters, 8192 * sizeof(int)));
    HIP_TRY(hipMemset(u->counters, 0, 8192 * sizeof(int)));
    if (getenv("SURFD_CONV_DEBUG")) { HIP_TRY(hipMalloc((void **)&u->dbg, 4096 * 16 * sizeof(long long))); HIP_TRY(hipMemset(u->dbg, 0, 4096 * 16 * sizeof(long long))); }
    u->allocated = true;
    return SURFD_OK;
}

// packs conv/linear weight `src` ([Cout][C][taps] row-major) into segment s of plan c
int pack_segment(surfd_unet *u, ConvPlan &c, int s, const float *src, int row_off, int rows, hipStream_t st) {
    // rows [row_off, row_off + rows) of the plan's output channels come from this tensor
    const SegPlan &sp = c.seg[s];
    const int Cp = ceil_div(sp.C, 8) * 8;
    if (row_off % 32) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "pack_segment: row offset %d not a multiple of 32", row_off);
    PackDesc pd;
    pd.src = src;
    pd.dst = u->wpack + c.w_off + (size_t)(row_off / 32) * c.KGtot * 256;
    pd.N = rows; pd.K = sp.taps * Cp;
    pd.Npad = ceil_div(rows, 32) * 32; pd.Kpad = sp.taps * Cp;
    pd.rs = (long)sp.C * sp.taps;
    pd.inner = Cp; pd.inner_valid = sp.C;
    pd.os = 1;          // outer index = tap: stride 1 in [C][taps]
    pd.is = sp.taps;    // inner index = channel: stride taps
    pd.KGtot = c.KGtot; pd.kg_off = c.kg_off[s];
    return launch_pack(pd, st);
}

struct WeightSite { ConvPlan *plan; int seg; int row_off; };

}  // namespace

extern "C" {

int surfd_unet_create(const surfd_unet_cfg *cfg, surfd_unet **out) {
    if (!cfg || !out) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_create: null argument");
    if (cfg->n_mult < 1 || cfg->n_mult > 8 || cfg->n_attn < 0 || cfg->n_attn > 8 || cfg->num_heads < 1 ||
        cfg->model_channels % 32 || cfg->model_channels % cfg->num_heads || cfg->num_res_blocks < 1)
        SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "surfd_unet_create: unsupported configuration");
    auto *u = new surfd_unet();
    u->cfg = *cfg;
    int rc = build_plan(u);
    if (rc) { delete u; return rc; }
    *out = u;
    return SURFD_OK;
}

void surfd_unet_destroy(surfd_unet *u) {
    if (!u) return;
    if (u->wpack) (void)hipFree(u->wpack);
    if (u->vecs) (void)hipFree(u->vecs);
    if (u->label_table) (void)hipFree(u->label_table);
    for (float *p : u->buf_ptr) if (p) (void)hipFree(p);
    for (float *p : {u->temb, u->h1, u->emb, u->emb_table, u->emb_ctx}) if (p) (void)hipFree(p);
    if (u->t_dev) (void)hipFree(u->t_dev);
    if (u->part) (void)hipFree(u->part);
    if (u->counters) (void)hipFree(u->counters);
    for (void *p : {(void *)u->whf, (void *)u->whf2, (void *)u->wsc, (void *)u->sat}) if (p) (void)hipFree(p);
    if (u->loop.exec) (void)hipGraphExecDestroy(u->loop.exec);
    if (u->loop.graph) (void)hipGraphDestroy(u->loop.graph);
    if (u->loop.cap_stream) (void)hipStreamDestroy(u->loop.cap_stream);
    if (u->loop.poll_stream) (void)hipStreamDestroy(u->loop.poll_stream);
    for (void *p : {(void *)u->loop.step_ctr, (void *)u->loop.x, (void *)u->loop.x0, u->loop.params, (void *)u->loop.tab}) if (p) (void)hipFree(p);
    delete u;
}

int surfd_unet_num_params(const surfd_unet *u) { return u ? (int)u->params.size() : 0; }

int surfd_unet_param_info(const surfd_unet *u, int i, const char **key, int64_t shape[4], int *ndim) {
    if (!u || i < 0 || i >= (int)u->params.size()) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_param_info: bad index %d", i);
    *key = u->params[i].key.c_str();
    *ndim = (int)u->params[i].shape.size();
    for (int j = 0; j < *ndim; ++j) shape[j] = u->params[i].shape[j];
    return SURFD_OK;
}

int surfd_unet_set_param(surfd_unet *u, const char *key, const void *dev_ptr, const int64_t *shape, int ndim, surfd_stream s) {
    if (!u || !key || !dev_ptr) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_set_param: null argument");
    std::string k(key);
    if (k.rfind("Unet.", 0) == 0) k = k.substr(5);
    auto it = u->pindex.find(k);
    if (it == u->pindex.end()) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_set_param: unexpected key '%s'", key);
    ParamInfo &p = u->params[it->second];
    if (ndim != (int)p.shape.size()) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_set_param: '%s' rank %d, expected %zu", key, ndim, p.shape.size());
    size_t numel = 1;
    for (int j = 0; j < ndim; ++j) {
        if (shape[j] != p.shape[j]) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_set_param: '%s' dim %d is %lld, expected %lld", key, j, (long long)shape[j], (long long)p.shape[j]);
        numel *= shape[j];
    }
    int rc = unet_alloc(u);
    if (rc) return rc;
    hipStream_t st = as_stream(s);
    const float *src = static_cast<const float *>(dev_ptr);
    if (ndim == 1) {
        HIP_TRY(hipMemcpyAsync(u->vecs + u->vec_off[k], src, numel * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (k == "label_emb.weight") {
        HIP_TRY(hipMemcpyAsync(u->label_table, src, numel * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        // find the plan segment(s) fed by this weight tensor
        bool placed = false;
        auto try_plan = [&](ConvPlan &c) -> int {
            for (int sidx = 0; sidx < c.nseg; ++sidx)
                if (c.seg[sidx].wkey == k) { placed = true; return pack_segment(u, c, sidx, src, 0, c.Cout, st); }
            return SURFD_OK;
        };
        for (auto &op : u->ops) if (op.kind == 0 && !placed) if ((rc = try_plan(op.conv))) return rc;
        if (!placed) if ((rc = try_plan(u->lin1))) return rc;
        if (!placed) if ((rc = try_plan(u->lin2))) return rc;
        if (!placed) {
            int row = 0;
            for (auto &el : u->emb_layers) {
                if (el.first + ".weight" == k) { placed = true; if ((rc = pack_segment(u, u->lin3, 0, src, row, el.second, st))) return rc; break; }
                row += el.second;
            }
        }
        if (!placed) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_set_param: no plan site for '%s'", key);
    }
    p.is_set = true;
    u->finalized = false;
    return SURFD_OK;
}

int surfd_unet_finalize(surfd_unet *u, surfd_stream s) {
    if (!u) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_finalize: null handle");
    for (auto &p : u->params)
        if (!p.is_set) SURFD_FAIL(SURFD_ERR_STATE, "surfd_unet_finalize: parameter '%s' was never set", p.key.c_str());
    hipStream_t st = as_stream(s);
    // combined biases (e.g. out_layers.3.bias + skip_connection.bias; time_embed.2.bias + sketch_emb.bias)
    auto combine = [&](ConvPlan &c) -> int {
        const float *a = u->vecs + u->vec_off[c.bias_keys[0]];
        const float *b = c.bias_keys.size() > 1 ? u->vecs + u->vec_off[c.bias_keys[1]] : nullptr;
        hipLaunchKernelGGL(vec_add_kernel, dim3(ceil_div(c.Cout, 256)), dim3(256), 0, st, u->vecs + c.bias_off, a, b, c.Cout);
        LAUNCH_CHECK();
        return SURFD_OK;
    };
    int rc;
    for (auto &op : u->ops) if (op.kind == 0) if ((rc = combine(op.conv))) return rc;
    if ((rc = combine(u->lin1))) return rc;
    if ((rc = combine(u->lin2))) return rc;
    // lin3 bias = concatenation of the 22 emb_layers biases (row offsets are multiples of 32 -> 4)
    int row = 0;
    for (auto &el : u->emb_layers) {
        HIP_TRY(hipMemcpyAsync(u->vecs + u->lin3.bias_off + row, u->vecs + u->vec_off[el.first + ".bias"],
                               (size_t)el.second * sizeof(float), hipMemcpyDeviceToDevice, st));
        row += el.second;
    }
    if ((rc = conv2_finalize(u, st))) return rc;
    u->finalized = true;
    u->ws_gen++;
    return SURFD_OK;
}

}  // extern "C"

// =============================================================================================
// host: execution
// =============================================================================================
namespace {

int ensure_workspace(surfd_unet *u, int B, int L) {
    if (u->ws_B >= B && u->ws_L >= L && !u->buf_ptr.empty()) return SURFD_OK;
    const int nB = std::max(B, u->ws_B), nL = std::max(L, u->ws_L);
    for (float *p : u->buf_ptr) if (p) HIP_TRY(hipFree(p));
    u->buf_ptr.assign(u->bufs.size(), nullptr);
    for (size_t i = 0; i < u->bufs.size(); ++i) {
        const size_t n = (size_t)nB * u->bufs[i].C * ceil_div(nL, u->bufs[i].ds);
        HIP_TRY(hipMalloc((void **)&u->buf_ptr[i], n * sizeof(float)));
    }
    u->ws_B = nB; u->ws_L = nL;
    u->ws_gen++;                 // pointers baked into a cached loop graph are stale now
    return SURFD_OK;
}

struct Resolved { float *ptr; long bstride; };

// Launch one planned convolution.  `B` batch entries, operand length Lseg(ds) = ds ? L/ds : 1.
int launch_conv(surfd_unet *u, const ConvPlan &c, int B, int L, const float *ext_in[2], const long ext_in_bs[2],
                const int ext_bmod[2], float *ext_out, long ext_out_bs, const float *emb, long emb_bs, hipStream_t st,
                const int *step_ptr = nullptr, const LoopFuse *lf = nullptr, bool *lf_done = nullptr,
                const float *lin_add = nullptr, long lin_add_step_stride = 0) {
    // dbg_only: -1 = every op; otherwise first | (last << 16): the ops [first, last] (last = 0: `first` alone) run on the f16x2 kernel
    const int only_first = u->dbg_only & 0xffff, only_last = (u->dbg_only >> 16) ? (u->dbg_only >> 16) : only_first;
    if (u->precision == 1 && c.f16_ok && (u->dbg_only < 0 || (c.id >= only_first && c.id <= only_last))) {
        const ConvLaunchIO io{ext_in[0], ext_in_bs[0], ext_out, ext_out_bs, emb, emb_bs, step_ptr, lf, lf_done};
        const int r2 = launch_conv2(u, c, B, L, io, st);
        if (r2 <= 0) return r2;          // launched (0) or failed (< 0); 1 = shape not covered -> fp32 kernel below
    }
    ConvArgs A;
    memset(&A, 0, sizeof(A));
    A.nseg = c.nseg; A.Cout = c.Cout; A.B = B;
    A.Lout = c.ds_out ? L / c.ds_out : 1;
    A.log2Lout = 0;
    while ((1 << A.log2Lout) < A.Lout) ++A.log2Lout;
    if ((1 << A.log2Lout) != A.Lout) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "conv: output length %d is not a power of two", A.Lout);
    A.wp = u->wpack + c.w_off; A.KGtot = c.KGtot;
    A.bias = u->vecs + c.bias_off;
    auto resolve = [&](const View &v, int ds, int which, bool is_out) -> Resolved {
        const int len = ds ? L / ds : 1;
        if (v.buf >= 0) {
            const BufInfo &bi = u->bufs[v.buf];
            const long bs = (long)bi.C * len;
            return {u->buf_ptr[v.buf] + (long)v.choff * len, bs};
        }
        if (is_out) return {ext_out, ext_out_bs};
        return {const_cast<float *>(ext_in[which]), ext_in_bs[which]};
    };
    int max_lsl = 1;
    for (int s = 0; s < c.nseg; ++s) {
        const SegPlan &sp = c.seg[s];
        SegArgs &S = A.seg[s];
        const Resolved r = resolve(sp.src, sp.ds, s, false);
        S.x = r.ptr; S.bstride = r.bstride;
        S.C = sp.C; S.Cp = ceil_div(sp.C, 8) * 8;
        S.Lin = sp.ds ? L / sp.ds : 1;
        S.taps = sp.taps; S.stride = sp.stride; S.ups = sp.ups; S.gn = sp.gn; S.act = sp.act;
        S.bmod = (sp.src.buf < 0) ? ext_bmod[s] : 0;
        if (sp.gn) { S.gamma = u->vecs + u->vec_off[sp.gnkey + ".weight"]; S.beta = u->vecs + u->vec_off[sp.gnkey + ".bias"]; }
        S.kg_off = c.kg_off[s];
        if (s == 0 && lin_add) { S.add = lin_add; S.add_step_stride = lin_add_step_stride; A.step_ptr = step_ptr; }
        const int lsl = sp.taps == 3 ? (sp.stride == 2 ? 2 * A.Lout + 1 : A.Lout + 2) : A.Lout;
        max_lsl = std::max(max_lsl, lsl);
    }
    A.Lsl = max_lsl;
    // ---- tiling: batch entries per workgroup and channels per staged chunk -----------------
    static const int budget_env = getenv("SURFD_CONV_BUDGET") ? atoi(getenv("SURFD_CONV_BUDGET")) : 0;   // debug knob
    const int budget = budget_env > 0 ? budget_env : 24576;   // floats of LDS for the slab (96 KB)
    auto min_cc = [&](const SegPlan &sp) {
        if (!sp.gn) return 8;
        const int gs = sp.C / 32;
        int g = 1;
        while ((g * gs) % 8) ++g;
        return g * gs;
    };
    int need = 8;
    for (int s = 0; s < c.nseg; ++s) need = std::max(need, min_cc(c.seg[s]));
    const int Lin0 = A.seg[0].Lin;
    for (int s = 1; s < c.nseg; ++s)
        if (A.seg[s].Lin != Lin0) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "conv: segments with different operand lengths");
    int log2lv = -2;
    const bool linear = c.seg[0].ds == 0;                      // embedding-path Linear layers
    switch (Lin0) { case 1: log2lv = linear ? -1 : 5; break; case 2: log2lv = 5; break; case 4: log2lv = 0; break;
                    case 8: log2lv = 1; break; case 16: log2lv = 2; break; case 32: log2lv = 3; break; case 64: log2lv = 4; break; }
    if (log2lv == -2) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "conv: operand length %d (supported: 1, 2, 4, 8, 16, 32, 64)", Lin0);
    const int lin_regs = std::max(Lin0, 4);
    // a thread stages at most CONV_VEC_MAX float4 in registers: nb * cc * Lin <= 256 * 128 floats
    const long reg_cap = 256L * CONV_VEC_MAX * 4;   // floats a workgroup can hold in registers while staging
    auto fits = [&](int bc, int cc) {
        const int lv = lin_regs / 4;                                     // float4 per row
        const long rpt = std::max(1, CONV_VEC_MAX / lv);                  // rows a thread can hold
        if (!linear) return (long)bc * A.Lsl * (cc + 4) <= budget && bc <= rpt && cc <= 256 && reg_cap > 0;   // thread <-> channel, row <-> batch entry
        return (long)bc * A.Lsl * (cc + 4) <= budget && (long)bc * cc / 4 <= 256 * rpt;                           // vectors for Linear layers
    };
    int bchunk = std::min(B, std::max(1, 512 / A.Lout));
    // wide form (surfd_unet_set_wide): the Linear layers of the embedding path chunk their rows and their K axis as for
    // eight rows whatever B is, so that an embedding row's summation order does not depend on the batch it is computed in
    const bool fixed_rows = linear && u->wide_batch > 0;
    const int chunk_rows = fixed_rows ? 8 : 0;
    if (fixed_rows) bchunk = std::min(B, chunk_rows);
    while (bchunk > 1 && !fits(bchunk, need)) --bchunk;
    if (!fits(bchunk, need))
        SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "conv: operand does not fit LDS/registers (L=%d, C=%d)", L, c.seg[0].C);
    // more, smaller workgroups while the chip is under-filled (weights are then re-read through L2)
    const int ntiles = ceil_div(c.Cout, 32);
    while (!fixed_rows && bchunk > 1 && ntiles * ceil_div(B, bchunk) < 128 && ((bchunk + 1) / 2) * A.Lout >= 32) bchunk = (bchunk + 1) / 2;
    A.bchunk = bchunk;
    // split K over workgroups when (channel tiles x batch chunks) cannot fill the chip: the heavy
    // low-resolution layers have 28 channel tiles and ONE batch chunk but stream 19 MB of weights
    const int nby = ceil_div(B, bchunk);
    int ks_target = 1;
    long work = 0;
    for (int s = 0; s < c.nseg; ++s) work += (long)c.seg[s].taps * (ceil_div(c.seg[s].C, 8) * 8);
    // the publish/acquire hand-off costs ~8 us: only worth it when a workgroup would otherwise stream
    // more than ~192 KB of weights on its own
    static const int nosplit_env = getenv("SURFD_CONV_NOSPLIT") ? 1 : 0;     // developer aid
    // (wide form, Linear layers: the K chunking follows the handle's design batch, never B — an embedding row's summation
    //  order must not depend on the batch it is computed in)
    const int nby_k = fixed_rows ? ceil_div(u->wide_batch, chunk_rows) : nby;
    if (!nosplit_env && ntiles * nby_k < 192 && work * 128 > 192 * 1024) ks_target = std::min(16, ceil_div(256, ntiles * nby_k));
    const long work_per_slice = ceil_div<long>(work, ks_target);
    int cs_max = 0, nchunks = 0;
    for (int s = 0; s < c.nseg; ++s) {
        const SegPlan &sp = c.seg[s];
        const int unit = min_cc(sp);
        const int Cp = ceil_div(sp.C, 8) * 8;
        int cc = std::min(ceil_div(Cp, unit) * unit, 4096);
        if (ks_target > 1) cc = std::min<long>(cc, std::max<long>(unit, ceil_div<long>(ceil_div<long>(work_per_slice, sp.taps), unit) * unit));
        while (cc > unit && !fits(fixed_rows ? chunk_rows : bchunk, cc)) cc -= unit;
        if (linear) { int p2 = 8; while (p2 * 2 <= cc) p2 *= 2; cc = p2; }   // power-of-two vectors per row (shift addressing)
        A.seg[s].cc = cc;
        nchunks += ceil_div(Cp, cc);
        cs_max = std::max(cs_max, cc + 4);
    }
    A.KS = std::max(1, std::min(ks_target, nchunks));
    A.part_stride = ceil_div(bchunk * A.Lout, 32) * 1024;
    if (A.KS > 1) {
        if ((size_t)A.KS * nby * ntiles * A.part_stride > u->part_floats || nby * ntiles > 8192) A.KS = 1;
        A.part = u->part; A.counters = u->counters;
    }
    A.cs_max = cs_max;
    if (c.emb_off >= 0 && emb) {
        A.emb = emb + c.emb_off; A.step_ptr = step_ptr;
        A.emb_bstride = u->emb_shared ? 0 : emb_bs; A.emb_step_stride = u->emb_shared ? emb_bs : (long)B * emb_bs;
        if (u->emb_ingraph) { A.step_ptr = nullptr; A.emb_step_stride = 0; }      // the iteration's own [B][14112] rows
    }
    if (c.res.buf != -1) { const Resolved r = resolve(c.res, c.ds_out, 0, false); A.res = r.ptr; A.res_bstride = r.bstride; }
    const Resolved o = resolve(c.dst, c.ds_out, 0, true);
    A.out = o.ptr; A.out_bstride = o.bstride;
    // slab, or (GroupNorm exchange arrays that alias it: 2 floats per row + 2 per (batch, group))
    bool any_gn = false;
    for (int s = 0; s < c.nseg; ++s) any_gn |= c.seg[s].gn != 0;
    A.red_off = bchunk * A.Lsl * cs_max;
    if (any_gn) A.red_off = std::max(A.red_off, 2 * bchunk * (cs_max - 4) + 2 * bchunk * 32 + 16);
    A.red_off = (A.red_off + 3) & ~3;
    const size_t lds_bytes = ((size_t)A.red_off + 3 * 1024 * 2) * sizeof(float);
    static const int ablate_env = getenv("SURFD_CONV_ABLATE") ? atoi(getenv("SURFD_CONV_ABLATE")) : 0;
    A.ablate = ablate_env;
    dim3 grid(ceil_div(c.Cout, 32), ceil_div(B, bchunk), A.KS);
    A.dbg = nullptr;
    if (u->dbg && u->dbg_launch < 4096) {
        A.dbg = u->dbg + (size_t)(u->dbg_launch++) * 16;
        long long meta[6] = {c.Cout, c.seg[0].C, A.Lout, grid.x * grid.y * grid.z, A.KS, bchunk};
        HIP_TRY(hipMemcpyAsync(A.dbg + 10, meta, sizeof(meta), hipMemcpyHostToDevice, st));
    }
    const bool one = ceil_div(bchunk * A.Lout, 32) <= 4;     // every wave owns at most one column tile
#define SURFD_LAUNCH(LV, P) do { if (one) hipLaunchKernelGGL((conv_kernel<LV, P, 1>), grid, dim3(256), lds_bytes, st, A); \
                                 else hipLaunchKernelGGL((conv_kernel<LV, P, 4>), grid, dim3(256), lds_bytes, st, A); } while (0)
    switch (log2lv) {
        case -1: SURFD_LAUNCH(-1, false); break;
        case 0: SURFD_LAUNCH(0, false); break;
        case 1: SURFD_LAUNCH(1, false); break;
        case 2: SURFD_LAUNCH(2, false); break;
        case 3: SURFD_LAUNCH(3, false); break;
        case 5: SURFD_LAUNCH(0, true); break;
        default: SURFD_LAUNCH(4, false); break;
    }
#undef SURFD_LAUNCH
    LAUNCH_CHECK();
    return SURFD_OK;
}

}  // namespace

namespace surfd {

// rows = one per (step, sample), or — `shared`: nothing but the timestep enters the embedding (no context, no labels) —
// one per step, read by every sample with a batch stride of 0
int unet_prepare_embeddings_dev(surfd_unet *u, const int64_t *t_dev, int rows, const float *ctx, const int64_t *cls,
                                int B, hipStream_t st, bool shared) {
    if (shared && (ctx || cls)) SURFD_FAIL(SURFD_ERR_ARG, "unet: shared embedding rows need an unconditional evaluation");
    if ((u->emb_shared != 0) != shared) { u->emb_shared = shared; u->ws_gen++; }       // captured graphs hold the strides
    if (u->emb_ingraph) { u->emb_ingraph = 0; u->ws_gen++; }
    if (!u->finalized) SURFD_FAIL(SURFD_ERR_STATE, "unet: parameters not finalized");
    if ((u->cfg.num_classes > 0) != (cls != nullptr))
        SURFD_FAIL(SURFD_ERR_ARG, "unet: class labels must be given if and only if the model is class-conditional");
    if (ctx && u->cfg.context_dim <= 0) SURFD_FAIL(SURFD_ERR_ARG, "unet: model has no context embedding");
    if (rows > u->emb_rows_cap) {
        for (float **p : {&u->temb, &u->h1, &u->emb, &u->emb_table}) { if (*p) HIP_TRY(hipFree(*p)); *p = nullptr; }
        HIP_TRY(hipMalloc((void **)&u->temb, (size_t)rows * u->cfg.model_channels * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&u->h1, (size_t)rows * u->ted * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&u->emb, (size_t)rows * u->ted * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&u->emb_table, (size_t)rows * u->emb_total * sizeof(float)));
        u->emb_rows_cap = rows;
        u->ws_gen++;
    }
    const int mc = u->cfg.model_channels;
    hipLaunchKernelGGL(temb_kernel, dim3((unsigned)std::min<long>(ceil_div<long>((long)rows * mc / 2, 256), 1024)), dim3(256), 0, st, t_dev, rows, mc, u->temb);
    LAUNCH_CHECK();
    int rc;
    {
        const float *in[2] = {u->temb, nullptr}; const long bs[2] = {mc, 0}; const int bm[2] = {0, 0};
        if ((rc = launch_conv(u, u->lin1, rows, 1, in, bs, bm, u->h1, u->ted, nullptr, 0, st))) return rc;
    }
    {
        // without a context the second K-segment is skipped (weights stay packed; nseg is forced to 1)
        ConvPlan l2 = u->lin2;
        if (!ctx) {
            l2.nseg = 1;
            // bias without sketch_emb.bias: plain time_embed.2.bias
            const float *in[2] = {u->h1, nullptr}; const long bs[2] = {u->ted, 0}; const int bm[2] = {0, 0};
            // temporarily point the combined-bias slot at the raw bias
            const size_t keep = l2.bias_off;
            l2.bias_off = u->vec_off["time_embed.2.bias"];
            rc = launch_conv(u, l2, rows, 1, in, bs, bm, u->emb, u->ted, nullptr, 0, st);
            l2.bias_off = keep;
            if (rc) return rc;
        } else {
            const float *in[2] = {u->h1, ctx}; const long bs[2] = {u->ted, u->cfg.context_dim}; const int bm[2] = {0, B};
            if ((rc = launch_conv(u, l2, rows, 1, in, bs, bm, u->emb, u->ted, nullptr, 0, st))) return rc;
        }
    }
    if (cls) {
        hipLaunchKernelGGL(add_label_kernel, dim3((unsigned)std::min<long>(ceil_div<long>((long)rows * u->ted, 256), 2048)), dim3(256), 0, st, u->emb, rows,
                           u->ted, (const float *)u->label_table, cls, B);
        LAUNCH_CHECK();
    }
    {
        const float *in[2] = {u->emb, nullptr}; const long bs[2] = {u->ted, 0}; const int bm[2] = {0, 0};
        if ((rc = launch_conv(u, u->lin3, rows, 1, in, bs, bm, u->emb_table, u->emb_total, nullptr, 0, st))) return rc;
    }
    u->emb_rows = rows; u->emb_B = B;
    return SURFD_OK;
}

static int ensure_t_dev(surfd_unet *u, int rows) {
    if (rows > u->t_cap) {
        if (u->t_dev) HIP_TRY(hipFree(u->t_dev));
        u->t_dev = nullptr;
        HIP_TRY(hipMalloc((void **)&u->t_dev, (size_t)rows * sizeof(int64_t)));
        u->t_cap = rows;
    }
    return SURFD_OK;
}

int unet_prepare_loop_embeddings(surfd_unet *u, const int64_t *t_steps_host, int T, const float *ctx, const int64_t *cls, int B,
                                 hipStream_t st) {
    if (!ctx && !cls) SURFD_FAIL(SURFD_ERR_ARG, "unet: in-loop embedding rows are for conditioned loops (context and / or labels)");
    if (!u->finalized) SURFD_FAIL(SURFD_ERR_STATE, "unet: parameters not finalized");
    if ((u->cfg.num_classes > 0) != (cls != nullptr))
        SURFD_FAIL(SURFD_ERR_ARG, "unet: class labels must be given if and only if the model is class-conditional");
    if (ctx && u->cfg.context_dim <= 0) SURFD_FAIL(SURFD_ERR_ARG, "unet: model has no context embedding");
    int rc = ensure_t_dev(u, T);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(u->t_dev, t_steps_host, (size_t)T * sizeof(int64_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // the host vector may go away after we return
    const int rows = std::max(T, B);      // temb / h1 / emb hold T rows, emb_table B rows of 14112
    if (rows > u->emb_rows_cap) {
        for (float **p : {&u->temb, &u->h1, &u->emb, &u->emb_table}) { if (*p) HIP_TRY(hipFree(*p)); *p = nullptr; }
        HIP_TRY(hipMalloc((void **)&u->temb, (size_t)rows * u->cfg.model_channels * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&u->h1, (size_t)rows * u->ted * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&u->emb, (size_t)rows * u->ted * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&u->emb_table, (size_t)rows * u->emb_total * sizeof(float)));
        u->emb_rows_cap = rows;
        u->ws_gen++;
    }
    if (B > u->emb_ctx_cap) {
        if (u->emb_ctx) HIP_TRY(hipFree(u->emb_ctx));
        u->emb_ctx = nullptr;
        HIP_TRY(hipMalloc((void **)&u->emb_ctx, (size_t)B * u->ted * sizeof(float)));
        u->emb_ctx_cap = B;
        u->ws_gen++;
    }
    if (!u->emb_ingraph || u->emb_shared) { u->emb_ingraph = 1; u->emb_shared = 0; u->ws_gen++; }
    const int mc = u->cfg.model_channels;
    // ---- per step: time_embed(timestep_embedding(t_k)) -> u->emb[T][ted] --------------------------------------------
    hipLaunchKernelGGL(temb_kernel, dim3((unsigned)std::min<long>(ceil_div<long>((long)T * mc / 2, 256), 1024)), dim3(256), 0, st, u->t_dev, T, mc, u->temb);
    LAUNCH_CHECK();
    {
        const float *in[2] = {u->temb, nullptr}; const long bs[2] = {mc, 0}; const int bm[2] = {0, 0};
        if ((rc = launch_conv(u, u->lin1, T, 1, in, bs, bm, u->h1, u->ted, nullptr, 0, st))) return rc;
        ConvPlan l2 = u->lin2;           // time_embed.2 alone: first K segment, its own bias
        l2.nseg = 1;
        l2.bias_off = u->vec_off["time_embed.2.bias"];
        const float *in2[2] = {u->h1, nullptr}; const long bs2[2] = {u->ted, 0};
        if ((rc = launch_conv(u, l2, T, 1, in2, bs2, bm, u->emb, u->ted, nullptr, 0, st))) return rc;
    }
    // ---- per sample: sketch_emb(ctx) + label_emb[cls] -> emb_ctx[B][ted] ------------------------------------------------
    if (ctx) {
        ConvPlan l2 = u->lin2;           // sketch_emb alone: the second K segment of the same packed rows, its own bias
        l2.nseg = 1;
        l2.seg[0] = u->lin2.seg[1];
        l2.kg_off[0] = u->lin2.kg_off[1];
        l2.bias_off = u->vec_off["sketch_emb.bias"];
        const float *in[2] = {ctx, nullptr}; const long bs[2] = {u->cfg.context_dim, 0}; const int bm[2] = {0, 0};
        if ((rc = launch_conv(u, l2, B, 1, in, bs, bm, u->emb_ctx, u->ted, nullptr, 0, st))) return rc;
    } else {
        HIP_TRY(hipMemsetAsync(u->emb_ctx, 0, (size_t)B * u->ted * sizeof(float), st));
    }
    if (cls) {
        hipLaunchKernelGGL(add_label_kernel, dim3((unsigned)std::min<long>(ceil_div<long>((long)B * u->ted, 256), 2048)), dim3(256), 0, st, u->emb_ctx, B,
                           u->ted, (const float *)u->label_table, cls, B);
        LAUNCH_CHECK();
    }
    u->emb_rows = B; u->emb_B = B;
    return SURFD_OK;
}

int unet_prepare_embeddings(surfd_unet *u, const int64_t *t_rows_host, int rows, const float *ctx, const int64_t *cls,
                            int B, hipStream_t st, bool shared) {
    if (rows > u->t_cap) {
        if (u->t_dev) HIP_TRY(hipFree(u->t_dev));
        u->t_dev = nullptr;
        HIP_TRY(hipMalloc((void **)&u->t_dev, (size_t)rows * sizeof(int64_t)));
        u->t_cap = rows;
    }
    HIP_TRY(hipMemcpyAsync(u->t_dev, t_rows_host, (size_t)rows * sizeof(int64_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // the host vector may go away after we return
    return unet_prepare_embeddings_dev(u, u->t_dev, rows, ctx, cls, B, st, shared);
}

LoopState *unet_loop_state(surfd_unet *u) { return &u->loop; }
int unet_device(surfd_unet *u) { return u->device; }
long unet_workspace_generation(surfd_unet *u) { return u->ws_gen; }

// one op of the denoiser body: x / out are the external input / output of the whole network
static int run_op(surfd_unet *u, const Op &op, const float *x, float *out, int B, int L, const float *emb, hipStream_t st,
                  const int *step_ptr, const LoopFuse *lf = nullptr, bool *lf_done = nullptr) {
    if (op.kind == 0) {
        const float *in[2] = {x, x}; const long bs[2] = {(long)u->cfg.in_channels * L, (long)u->cfg.in_channels * L};
        const int bm[2] = {0, 0};
        return launch_conv(u, op.conv, B, L, in, bs, bm, out, (long)u->cfg.out_channels * L, emb, u->emb_total, st, step_ptr, lf, lf_done);
    }
    const AttnPlan &a = op.attn;
    const int T = L / a.ds, heads = u->cfg.num_heads, d = a.C / heads;
    const float *qkv = u->buf_ptr[a.qkv.buf];
    float *o = u->buf_ptr[a.out.buf];
    const long qbs = (long)u->bufs[a.qkv.buf].C * T, obs = (long)u->bufs[a.out.buf].C * T;
    const float scale = 1.f / sqrtf(sqrtf((float)d));
    if (T > 64) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "attention: sequence length %d (at most 64 positions per attention block)", T);
    const int qtn = (T + 31) / 32;
    const size_t lds_bytes = ((size_t)3 * d * (T + 1) + (size_t)4 * qtn * 16 * 64) * sizeof(float);
    hipLaunchKernelGGL(attn_kernel, dim3(B * heads * qtn), dim3(256), lds_bytes, st, qkv, qbs, o, obs, heads, d, T, scale);
    LAUNCH_CHECK();
    return SURFD_OK;
}

int unet_forward_prepared(surfd_unet *u, const float *x, int row0, float *out, int B, int L, hipStream_t st,
                          const int *step_ptr, const LoopFuse *lf, bool *lf_done) {
    if (!u->finalized) SURFD_FAIL(SURFD_ERR_STATE, "unet: parameters not finalized");
    if (row0 < 0 || row0 + (u->emb_shared ? 1 : B) > u->emb_rows) SURFD_FAIL(SURFD_ERR_STATE, "unet: embedding rows [%d,%d) not prepared", row0, row0 + B);
    if (u->emb_ingraph && (row0 != 0 || B != u->emb_B)) SURFD_FAIL(SURFD_ERR_STATE, "unet: in-loop embedding rows were prepared for %d samples", u->emb_B);
    int max_ds = 1;
    for (auto &b : u->bufs) max_ds = std::max(max_ds, b.ds);
    if (L % max_ds || L < max_ds || L > 64 || (L & (L - 1))) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "unet: latent length %d must be a power of two in [%d, 64]", L, max_ds);
    int rc = ensure_workspace(u, B, L);
    if (rc) return rc;
    const float *emb = u->emb_table + (size_t)row0 * u->emb_total;
    if (u->emb_ingraph) {
        // emb_table[b] = emb_layers(SiLU(time_embed(t_k) + sample part[b])) for THIS iteration (openaimodel.py:724-735, 218-224)
        const float *in[2] = {u->emb_ctx, nullptr}; const long bs[2] = {u->ted, 0}; const int bm[2] = {0, 0};
        if ((rc = launch_conv(u, u->lin3, B, 1, in, bs, bm, u->emb_table, u->emb_total, nullptr, 0, st, step_ptr, nullptr, nullptr, u->emb, u->ted))) return rc;
    }
    // every convolution learns which convolution follows it (attention launches in between do not count; after the head the
    // next evaluation's first one): it requests that layer's weights while it finishes (conv_f16x2.hip, SURFD_C2_PFN)
    const ConvPlan *first_conv = nullptr;
    for (auto &op : u->ops) if (op.kind == 0) { first_conv = &op.conv; break; }
    for (size_t i = 0; i < u->ops.size(); ++i) {
        u->pf_next = first_conv;
        for (size_t j = i + 1; j < u->ops.size(); ++j) if (u->ops[j].kind == 0) { u->pf_next = &u->ops[j].conv; break; }
        rc = run_op(u, u->ops[i], x, out, B, L, emb, st, step_ptr, lf, lf_done);
        u->pf_next = nullptr;
        if (rc) return rc;
    }
    return SURFD_OK;
}

}  // namespace surfd

extern "C" int surfd_unet_debug_read(surfd_unet *u, long long *out, int max_launches) {
    if (!u || !u->dbg) return 0;
    const int n = std::min(u->dbg_launch, max_launches);
    if (hipMemcpy(out, u->dbg, (size_t)n * 16 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    u->dbg_launch = 0;
    return n;
}

// Test tap: runs ONLY the ops of one module (e.g. "input_blocks.1.0", "middle_block.1", "out") on a given input
// activation, with the embedding rows prepared by the preceding surfd_unet_forward (same t / conditioning, same B, L).
// in[B, Cin, Lin] -> out[B, Cout, Lout]; shapes are checked against the plan.
extern "C" int surfd_unet_debug_run_module(surfd_unet *u, const char *module, const float *in, int Cin, int Lin,
                                           float *out, int Cout, int Lout, int B, int L, surfd_stream s) {
    if (!u || !module || !in || !out) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_debug_run_module: null argument");
    if (!u->finalized || u->emb_rows < B || u->ws_B < B || u->ws_L < L)
        SURFD_FAIL(SURFD_ERR_STATE, "surfd_unet_debug_run_module: call surfd_unet_forward with the same B, L first");
    hipStream_t st = as_stream(s);
    const std::string pre = std::string(module) + ".";
    std::vector<const Op *> sel;
    bool prev = false;
    for (auto &op : u->ops) {
        const bool in_mod = op.kind == 0 ? op.conv.seg[0].wkey.rfind(pre, 0) == 0 : prev;
        if (in_mod) sel.push_back(&op);
        prev = in_mod;
    }
    while (!sel.empty() && sel.back()->kind != 0) sel.pop_back();      // "input_blocks.1.1.qkv": the projection alone, without the attention core behind it
    if (sel.empty() || sel.front()->kind != 0 || sel.back()->kind != 0)
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_debug_run_module: no ops for module '%s'", module);
    const ConvPlan &first = sel.front()->conv, &last = sel.back()->conv;
    const int lin = L / first.seg[0].ds, lout = L / last.ds_out;
    if (first.seg[0].C != Cin || lin != Lin || last.Cout != Cout || lout != Lout)
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_debug_run_module: '%s' maps [%d,%d] -> [%d,%d]", module, first.seg[0].C, lin, last.Cout, lout);
    const View src = first.seg[0].src, dst = last.dst;
    const float *ext_in = in;
    if (src.buf >= 0) {
        float *p = u->buf_ptr[src.buf] + (long)src.choff * lin;
        HIP_TRY(hipMemcpy2DAsync(p, (size_t)u->bufs[src.buf].C * lin * sizeof(float), in, (size_t)Cin * lin * sizeof(float),
                                 (size_t)Cin * lin * sizeof(float), B, hipMemcpyDeviceToDevice, st));
    }
    int rc;
    for (const Op *op : sel)
        if ((rc = run_op(u, *op, ext_in, out, B, L, u->emb_table, st, nullptr))) return rc;
    if (dst.buf >= 0) {
        const float *p = u->buf_ptr[dst.buf] + (long)dst.choff * lout;
        HIP_TRY(hipMemcpy2DAsync(out, (size_t)Cout * lout * sizeof(float), p, (size_t)u->bufs[dst.buf].C * lout * sizeof(float),
                                 (size_t)Cout * lout * sizeof(float), B, hipMemcpyDeviceToDevice, st));
    }
    return SURFD_OK;
}

extern "C" int surfd_unet_set_precision(surfd_unet *u, int mode) {
    if (!u || (mode != 0 && mode != 1)) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_set_precision: mode must be 0 (fp32) or 1 (f16x2)");
    if (u->precision != mode) { u->precision = mode; u->ws_gen++; }
    return SURFD_OK;
}

extern "C" int surfd_unet_set_cu_budget(surfd_unet *u, int cus) {
    if (!u || cus < 1 || cus > 256) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_set_cu_budget: need 1 <= cus <= 256");
    if (u->cu_budget != cus) { u->cu_budget = cus; u->ws_gen++; }      // captured loop graphs hold the old launch shapes
    return SURFD_OK;
}

extern "C" int surfd_unet_set_wide(surfd_unet *u, int design_batch) {
    if (!u || design_batch < 0 || design_batch > 4096) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_set_wide: need 0 <= design_batch <= 4096");
    if (u->wide_batch != design_batch) { u->wide_batch = design_batch; u->ws_gen++; }      // captured loop graphs hold the old launch shapes
    return SURFD_OK;
}

// developer aid: op >= 0 restricts the f16x2 kernel to that one conv op (all others run exact fp32), -1 lifts it
extern "C" int surfd_unet_debug_only_op(surfd_unet *u, int op) {
    if (!u) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_debug_only_op: null handle");
    u->dbg_only = op; u->ws_gen++;
    return SURFD_OK;
}

extern "C" int surfd_unet_saturation_count(surfd_unet *u, int reset, int64_t *count, surfd_stream s) {
    if (!u || !count) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_saturation_count: null argument");
    *count = 0;
    if (!u->sat) return SURFD_OK;
    unsigned v = 0;
    hipStream_t st = as_stream(s);
    HIP_TRY(hipMemcpyAsync(&v, u->sat, sizeof(v), hipMemcpyDeviceToHost, st));
    if (reset) HIP_TRY(hipMemsetAsync(u->sat, 0, sizeof(v), st));
    HIP_TRY(hipStreamSynchronize(st));
    *count = v;
    return SURFD_OK;
}

extern "C" int surfd_unet_forward(surfd_unet *u, const float *x, const int64_t *t, const float *ctx, const int64_t *cls,
                                  float *out, int B, int L, surfd_stream s) {
    if (!u || !x || !t || !out || B < 1 || L < 1) SURFD_FAIL(SURFD_ERR_ARG, "surfd_unet_forward: bad argument");
    hipStream_t st = as_stream(s);
    int rc = surfd::unet_prepare_embeddings_dev(u, t, B, ctx, cls, B, st, false);
    if (rc) return rc;
    return surfd::unet_forward_prepared(u, x, 0, out, B, L, st);
}
