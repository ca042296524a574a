// Sixteen-wave latency form of the split-fp16 convolution of the latent denoiser (reference models/openaimodel.py:255-275
// ResBlock, :318-324 AttentionBlock projections, :134-160 Downsample, :91-119 Upsample, :682-686 head) for gfx950 — what a
// narrow reverse loop (sample/generate_*: 8 latents; bench.py's strict_c3) spends its time in.  Round 6.
//
// Same arithmetic and the same packed weights as conv2_kernel (conv_f16x2.hip): Out[Cout x (b,l)] = W[Cout x K] * im2col(act(GN(x))),
// operands split into two fp16 terms, three products on v_mfma_f32_32x32x16_f16, fp32 accumulation, GroupNorm + SiLU fused into
// the LDS staging, skip convolution as a second K segment, bias / embedding / residual (and the fused posterior update of the
// loop's head) in the epilogue.  What differs is who does what.
//
// One evaluation of a narrow loop is a chain of 100 dependent launches, each of which lives exactly as long as ONE workgroup:
// 8-21 us in the four-wave form (profiles/r06_latency_form.md: kernel-argument fetch and address arithmetic 2.0, operand wait +
// GroupNorm statistics 1.4, GroupNorm apply / SiLU / split / 64 ds_write_b16 per thread 2.1-3.6, K loop 1.6 + 0.4-6,
// k-part reduction 0.5, split-K hand-off 2-4, epilogue 0.9) while the chip is mostly idle (56-512 workgroups of four waves).
// The four-wave form gives every thread 32 operand values to normalise, activate, split and store, 8-11 k16 steps of weights
// to stream per wave and a sixteenth of a tile's epilogue per lane — all of it on the critical path of the launch.
//
// Here a workgroup is SIXTEEN waves (1024 threads, one workgroup per CU, 128 registers per lane):
//  * staging: thread <-> (channel, four consecutive positions) — at most four float4 per thread instead of eight or sixteen.
//    A GroupNorm group of one sample occupies 2^k consecutive lanes of ONE wave (channel slot x position chunk), so its
//    statistics are the same in-wave DPP / permlane butterflies with padded cross-lane reads as in conv2_kernel
//    (conv2_dev.h; profiles/r06_conv2_instability.md), now as a plain two-pass mean / variance over the unit's lanes;
//  * matrix work: the sixteen waves are (row tiles RT) x (column tiles) x (k-parts KP), RT * nct * KP = 16: with one row tile
//    and one column tile a wave runs 3 of the 42 k16 steps of a three-tap 224-channel block — all of its weight fragments
//    are requested by its first instructions;
//  * k-part reduction and epilogue are DISTRIBUTED: every wave writes its partial tile to LDS, then wave kp sums registers
//    [kp * 16 / KP, (kp + 1) * 16 / KP) of its tile over the k-parts in k-part order and finishes exactly those rows — bias,
//    embedding, residual and store of a tile are sixteen waves' work instead of one's, and the split-K hand-off reads
//    16 / KP registers per lane and slice, all slices in flight at once;
//  * the epilogue operands are requested by the first instructions (there are registers to spare).
// Decomposition over workgroups (XCD-aware block decode, K slices of whole K blocks, split-K through write-through partial tiles
// and an arrival ticket, summed in slice order by the last arriver) is conv2_kernel's.
#include "common.h"
#include "unet_api.h"
#include "unet_plan.h"
#include "conv2_dev.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace surfd {

struct SegX {
    const float *x;        // source view (channel offset applied)
    long bstride;          // floats between batch entries
    const float *gamma, *beta;
    int C, Lin, log2Lin;
    int taps, stride, ups, gn, act;
    int blk, blkp, nblk;   // channels per K block (= staged chunk), padded to 16, number of blocks
    int k16_off;           // first k16 step of the segment
    // staging map: unit = (channel slot g of the K block, batch row i); a unit takes 2^(lp + ql) lanes of one wave:
    // lane in unit = (channel in slot) * 2^ql + position chunk; a thread holds 2^lfu consecutive float4 of its chunk per pass
    int gs;                // channels per slot (GroupNorm group size; 8 for segments without GroupNorm)
    int lp, ql, lfu;       // log2: lanes per slot along the channels, position chunks across lanes, float4 per thread and unit
    int ng;                // slots per K block
    unsigned magic_ng;     // ceil(2^32 / ng): u / ng == umulhi(u, magic) for u < 2^16 (ng > 1)
    int npass;             // units per thread (passes over the block); npass << lfu <= 4
    float inv_cnt;         // 1 / (gs * Lin)
};

struct ConvXArgs {
    SegX seg[2];
    int nseg;
    const _Float16 *whf;   // [ntiles][KS16][2 planes][64 lanes][8]
    int KS16;
    float inv_sc;
    const float *bias;     // [Cout]
    const float *emb;      // emb[b * emb_bstride + co]; unused operands point at the bias vector with zero strides
    long emb_bstride;
    const float *res;      // res[b * res_bstride + co * res_cstride + l * res_lstride]
    long res_bstride;
    int res_cstride, res_lstride, has_res, has_emb;
    float *out;
    long out_bstride;
    int Cout, Lout, log2Lout, B;
    int bchunk, Lsl, cs;   // batch entries per workgroup, slab positions per batch entry, slab row stride (halfs)
    int off_flag;          // byte offset of the split-K flag in LDS
    int ntiles, nby, KS, nrt;
    int RT, log2RT, log2nct, log2KP;     // waves = RT row tiles x nct column tiles x KP k-parts (= 16)
    unsigned magic_nby, magic_ks, magic_g;
    float *part;           // split-K partial tiles [KS][nby][ntiles][2048]
    int *counters;         // [nby][nrt], zero between launches
    const int *step_ptr;   // device loop counter (embedding rows advance by emb_step_stride per step) or null
    long emb_step_stride;
    unsigned *sat;         // saturation counter
    const LoopFuse *lf;    // head convolution inside the graph-replayed loop (LF instantiation)
    long long *dbg;        // -DSURFD_C2_STAMPS builds: phase stamps of workgroup 0
    long pad_[7];          // the argument block fills its eight 64-byte lines (kernel-argument prefetch)
};
static_assert(sizeof(ConvXArgs) > 0x1c0 && sizeof(ConvXArgs) <= 0x200, "ConvXArgs: eight lines of kernel arguments");

constexpr int CX_PLANE = 16384;     // halfs per fp16 plane of the slab; the two planes (64 KB) are also the k-part reduction area
constexpr int CX_D = 4;             // weight ring: stages of one k16 step (8 registers each)

template <bool LF>
__global__ __launch_bounds__(1024) void conv2x_kernel(ConvXArgs A) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    _Float16 *slab = reinterpret_cast<_Float16 *>(lds_raw);
    float *red = reinterpret_cast<float *>(lds_raw);                   // [16 waves][16 registers][64 lanes], after the last matrix instruction
    int *flag = reinterpret_cast<int *>(lds_raw + A.off_flag);
    c2_kernarg_prefetch<(int)sizeof(ConvXArgs)>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SURFD_C2_STAMPS
    long long stamp_[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) stamp_[i] = 0;
#define CX_STAMP(i) do { stamp_[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define CX_STAMP_FIRST(i) do { if (ch == kz) CX_STAMP(i); } while (0)
    const long long cyc0_ = (long long)__builtin_readcyclecounter();
#else
#define CX_STAMP(i) do { } while (0)
#define CX_STAMP_FIRST(i) do { } while (0)
#endif
    CX_STAMP(0);
    // ---- XCD-aware decode (conv2_kernel's): group g = (row group, K slice); every batch chunk of a group on XCD g % 8 ----
    int by, kz, rg;
    {
        const int bid = blockIdx.x;
        const int G = A.nrt * A.KS;
        int g;
        if (G < 8) {
            by = G == 1 ? bid : (int)__umulhi((unsigned)bid, A.magic_g);
            g = bid - by * G;
        } else {
            const int xcd = bid & 7, idx = bid >> 3;
            const int j = A.nby == 1 ? idx : (int)__umulhi((unsigned)idx, A.magic_nby);
            by = idx - j * A.nby;
            g = xcd + 8 * j;
            if (g >= G) return;
        }
        rg = A.KS == 1 ? g : (int)__umulhi((unsigned)g, A.magic_ks);
        kz = g - rg * A.KS;
    }
    // ---- the wave's role: (row tile of the group, column tile, k-part) ----
    const int log2KP = A.log2KP, KP = 1 << log2KP;
    const int kpart = wave & (KP - 1);
    const int ct = (wave >> log2KP) & ((1 << A.log2nct) - 1);
    const int rt = wave >> (log2KP + A.log2nct);
    int tile = (rg << A.log2RT) + rt;
    const bool tile_ok = tile < A.ntiles;          // the last row group may have fewer than RT tiles (wave-uniform)
    tile = min(tile, A.ntiles - 1);                // an idle wave streams the last tile's weights (valid addresses, same schedule) and stores nothing
    const int b0 = by * A.bchunk;
    const int nb = min(A.bchunk, A.B - b0);
    const int M = nb * A.Lout;
    int colb, coll;
    {
        int m = ct * 32 + (lane & 31);
        if (m >= M) m = 0;
        colb = m >> A.log2Lout;
        coll = m & (A.Lout - 1);
    }
    const int nblk0 = A.seg[0].nblk;
    const int nch = nblk0 + (A.nseg > 1 ? A.seg[1].nblk : 0);
    constexpr int PLANE = CX_PLANE;
    const int cs = A.cs;
    const float inv_sc = A.inv_sc;

    // ONE accumulator (small terms first): a wave runs a handful of k16 steps, the dependent issue of its nine matrix instructions
    // costs less than the sixteen registers of a second accumulator (128 per lane at four waves per SIMD)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // ---- weight stream state of one K block for this wave (conv2_kernel's scheme with one k16 step per ring stage): every wave
    //      runs the same static schedule, all weight loads are unconditional, addresses clamped into the block ----
    struct WS { const _Float16 *base; int it_beg, it_end, ngroups, nk; };
    auto make_ws = [&](int ch) -> WS {
        const int s = ch >= nblk0 ? 1 : 0;
        const int bi = ch - (s ? nblk0 : 0);
        const int nk = A.seg[s].blkp >> 4;
        const int iters = A.seg[s].taps * nk;
        const int per = (iters + KP - 1) >> log2KP;
        WS w;
        w.nk = nk;
        w.it_beg = min(kpart * per, iters);
        w.it_end = min(iters, w.it_beg + per);
        w.ngroups = per;
        w.base = A.whf + ((size_t)tile * A.KS16 + A.seg[s].k16_off + (size_t)bi * iters) * 1024 + lane * 8;
        return w;
    };
    f16x8 ring[CX_D][2];
    auto load_step = [&](f16x8 (&dst)[2], const _Float16 *base, int it0, int it_last) {
        const int it = max(min(it0, it_last), 0);
        const gf16x8 *p = (const gf16x8 *)(base + (size_t)it * 1024);
        dst[0] = p[0];
        dst[1] = p[64];          // low plane: +512 halfs
    };

    // ---- staging map of this thread for segment s and pass p: batch row, channel of the block, validity, first float4 ----
    struct Unit { int i, c, f4; bool ok; };
    auto unit_of = [&](int s, int p) -> Unit {
        const SegX &S = A.seg[s];
        const int lu = S.lp + S.ql;
        const int li = lane & ((1 << lu) - 1);
        const int pq = li & ((1 << S.ql) - 1), sl = li >> S.ql;
        const int pc = min(p, S.npass - 1);
        const int u = ((pc * 16 + wave) << (6 - lu)) + (lane >> lu);
        Unit r;
        r.i = S.ng == 1 ? u : (int)__umulhi((unsigned)u, S.magic_ng);
        const int g = u - r.i * S.ng;
        r.c = g * S.gs + sl;
        r.ok = p < S.npass && r.i < nb && sl < S.gs && r.c < S.blk;
        r.f4 = pq << S.lfu;
        return r;
    };
    // ---- raw operand of one K block: four float4 per thread (pass p = j >> lfu, float4 f = j & (2^lfu - 1) of the thread's
    //      chunk); unconditional loads, addresses clamped into the tensor ----
    auto issue_operand = [&](int ch, f32x4 (&v)[4], float (&ga)[4], float (&be)[4]) {
        const int s = ch >= nblk0 ? 1 : 0;
        const int bi = ch - (s ? nblk0 : 0);
        const SegX &S = A.seg[s];
        const int lfu = S.lfu;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Unit un = unit_of(s, j >> lfu);
            const int cg = min(bi * S.blk + min(un.c, S.blk - 1), S.C - 1);
            const int i = min(un.i, nb - 1);
            const float *src = S.x + (long)cg * S.Lin + (b0 + i) * S.bstride + 4 * (un.f4 + (j & ((1 << lfu) - 1)));
            v[j] = *reinterpret_cast<const f32x4 *>(src);
            ga[j] = S.gamma[cg]; be[j] = S.beta[cg];          // segments without GroupNorm point these at the bias vector
        }
    };

    int ch = kz;
    WS cur = make_ws(ch);
    f32x4 v[4];
    float ga[4], be[4];
    issue_operand(ch, v, ga, be);
#pragma unroll
    for (int d = 0; d < CX_D; ++d) load_step(ring[d], cur.base, cur.it_beg + d, cur.it_end - 1);

    // ---- epilogue operands of the rows this wave will finish (registers kpart * NR + rr, rr < NR = 16 / KP <= 4), requested now ----
    const int nr = 16 >> log2KP;
    float pre_b[4], pre_e[4], pre_r[4];
    const int m_ep = min(ct * 32 + (lane & 31), M - 1);
    const int b_ep = b0 + (m_ep >> A.log2Lout), l_ep = m_ep & (A.Lout - 1);
    LoopFuse lfv;
    int lfk = 0;
    float lfrow[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    auto request_epilogue = [&]() {
        const float *embp = A.emb;
        if (A.step_ptr) embp += (long)(*reinterpret_cast<const __attribute__((address_space(4))) int *>((unsigned long)A.step_ptr)) * A.emb_step_stride;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = kpart * nr + min(rr, nr - 1);
            const int co = min(tile * 32 + frag_row(r, lane), A.Cout - 1);
            pre_b[rr] = A.bias[co];
            pre_e[rr] = embp[b_ep * A.emb_bstride + co];
            pre_r[rr] = A.res[b_ep * A.res_bstride + (long)co * A.res_cstride + l_ep * A.res_lstride];
        }
        if constexpr (LF) {
            lfv = *A.lf;
            lfk = *lfv.step;
#pragma unroll
            for (int q = 0; q < 5; ++q) lfrow[q] = lfv.tab[(long)lfk * 8 + q];
        }
    };
    request_epilogue();
    bool saturated = false;
    CX_STAMP(1);

    while (true) {
        // =========================== stage K block `ch` into the slab ===============================
        {
            const int s = ch >= nblk0 ? 1 : 0;
            const SegX &S = A.seg[s];
            const int lfu = S.lfu, lu = S.lp + S.ql;
            const int blk = S.blk, blkp = S.blkp;
            const int pad = S.taps == 3 ? 1 : 0;
            const int ups = S.ups, act = S.act;
            Unit un[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) un[j] = unit_of(s, j >> lfu);
            if (S.gn) {
                // ---- GroupNorm statistics of the unit (one group of one sample), two-pass, inside the wave: the thread's sums go
                //      up a binary tree over its 2^lfu float4 (levels enabled by the wave-uniform lfu, no dynamic register index),
                //      then through the butterfly over the unit's 2^lu lanes; every lane ends with the unit's total ----
                auto tree = [&](float (&t)[4]) {
                    if (lfu > 0) {
                        const float a = t[0] + t[1], b = t[2] + t[3];
                        t[0] = t[1] = a; t[2] = t[3] = b;
                    }
                    if (lfu > 1) {
                        const float a = t[0] + t[2];
                        t[0] = t[1] = t[2] = t[3] = a;
                    }
                };
                float a[4], mean[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = un[j].ok ? (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]) : 0.f;
                tree(a);
                c2_slot_sum<4>(a, lu);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mean[j] = a[j] * S.inv_cnt;
                    float m2 = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float d = v[j][q] - mean[j]; m2 += d * d; }
                    a[j] = un[j].ok ? m2 : 0.f;
                }
                tree(a);
                c2_slot_sum<4>(a, lu);
                CX_STAMP_FIRST(2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gsc = ga[j] * (1.f / sqrtf(a[j] * S.inv_cnt + 1e-5f));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float w = (v[j][k] - mean[j]) * gsc + be[j];
                        if (act) w = silu2(w);
                        v[j][k] = w;
                    }
                }
            } else if (act) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[j][k] = silu2(v[j][k]);
            }
            CX_STAMP_FIRST(3);
            // ---- split and write the slab [batch row][position][channel] (conv2_kernel's layout): one packed conversion per pair
            //      of positions, the halves stored with ds_write_b16 / ds_write_b16_d16_hi, low plane at a compile-time offset ----
            const int rstep = ups ? 2 : 1;
            const int Lcov = ups ? 2 * S.Lin : S.Lin;
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (un[j].ok) {
                    const int f4 = un[j].f4 + (j & ((1 << lfu) - 1));
                    _Float16 *row = slab + (un[j].i * A.Lsl + pad + rstep * 4 * f4) * cs + un[j].c;
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        amax = fmaxf(amax, fmaxf(fabsf(v[j][k]), fabsf(v[j][k + 1])));
                        const f32x2 w = {__builtin_amdgcn_fmed3f(v[j][k], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[j][k + 1], -65504.f, 65504.f)};
                        const f16x2 h = __builtin_convertvector(w, f16x2);
                        const f32x2 hf = __builtin_convertvector(h, f32x2);
                        const f16x2 lo = __builtin_convertvector(w - hf, f16x2);
                        _Float16 *d0 = row + (k * rstep) * cs, *d1 = d0 + rstep * cs;
                        d0[0] = h[0]; d0[PLANE] = lo[0];
                        d1[0] = h[1]; d1[PLANE] = lo[1];
                        if (ups) { d0[cs] = h[0]; d0[cs + PLANE] = lo[0]; d1[cs] = h[1]; d1[cs + PLANE] = lo[1]; }
                    }
                }
            }
            saturated |= amax > 65504.f;
            // channels that only exist as padding of the K block, and the halo positions of the real ones: zeros.  (row, position)
            // pairs over the waves, channels over the lanes
            {
                const int nhalo = A.Lsl - Lcov;
                for (int rp = wave; rp < nb * A.Lsl; rp += 16)
                    for (int c = blk + lane; c < blkp; c += 64) { slab[rp * cs + c] = (_Float16)0.f; slab[rp * cs + c + PLANE] = (_Float16)0.f; }
                for (int hp = wave; hp < nb * nhalo; hp += 16) {
                    const int i = hp / nhalo, h = hp - i * nhalo;
                    const int p = h < pad ? h : Lcov + h;          // halo positions: [0, pad) and [pad + Lcov, Lsl)
                    for (int c = lane; c < blk; c += 64) { slab[(i * A.Lsl + p) * cs + c] = (_Float16)0.f; slab[(i * A.Lsl + p) * cs + c + PLANE] = (_Float16)0.f; }
                }
            }
        }
        // the raw operand of the next K block, requested now: its registers are free (unconditional: re-reads this block at the end)
        const int chn = ch + A.KS;
        const bool has_next = chn < nch;
        const WS nxt = has_next ? make_ws(chn) : cur;
        issue_operand(has_next ? chn : ch, v, ga, be);
        lds_bar();
        CX_STAMP_FIRST(4);
        // =========================== MFMAs of this K block ============================================
        {
            const int s = ch >= nblk0 ? 1 : 0;
            const int lbase = (colb * A.Lsl + coll * A.seg[s].stride) * cs + 8 * (lane >> 5);
            const int nk = cur.nk;
            auto compute = [&](const f16x8 (&a)[2], int g) {
                const int it = cur.it_beg + g;
                if (it < cur.it_end) {          // wave-uniform; only LDS reads and MFMAs inside
                    const int tap = (it >= nk) + (it >= 2 * nk);
                    const _Float16 *bp = slab + lbase + tap * cs + (it - tap * nk) * 16;
                    const f16x8 bh = *reinterpret_cast<const f16x8 *>(bp);
                    const f16x8 bl = *reinterpret_cast<const f16x8 *>(bp + PLANE);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bh, acc, 0, 0, 0);
                }
            };
            const int P = ((cur.ngroups + CX_D - 1) / CX_D) * CX_D;       // ngroups >= 1
            int g0 = 0;
#pragma unroll 1
            do {
#pragma unroll
                for (int d = 0; d < CX_D; ++d) {
                    const int g = g0 + d;
                    compute(ring[d], g);
                    // refill this stage: step g + D of this block, or (last pass) step d of the next block, or a harmless re-load
                    const int vg = g + CX_D;
                    const bool into_next = has_next && vg >= P;
                    const _Float16 *rb = into_next ? nxt.base : cur.base;
                    const int it0 = into_next ? nxt.it_beg + (vg - P) : cur.it_beg + vg;
                    const int itl = into_next ? nxt.it_end - 1 : cur.it_end - 1;
                    load_step(ring[d], rb, it0, itl);
                }
                g0 += CX_D;
            } while (g0 < P);
        }
        CX_STAMP_FIRST(5);
        lds_bar();                   // every wave is done reading the slab (next block's staging, or the reduction area that aliases it)
        if (!has_next) break;
        ch = chn;
        cur = nxt;
    }
    if (saturated) atomicAdd(A.sat, 1u);
    CX_STAMP(6);

    // ---- k-part reduction through LDS, distributed: every wave leaves its partial tile, wave kpart then owns NR = 16 / KP
    //      accumulator registers of its tile (rows frag_row(kpart * NR + rr, lane)) and sums them over the k-parts in k-part order ----
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    lds_bar();
    const int w0 = wave - kpart;             // the tile's first k-part
    float val[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = kpart * nr + min(rr, nr - 1);
        float t = 0.f;
        for (int k = 0; k < KP; ++k) t += red[((w0 + k) * 16 + r) * 64 + lane];
        val[rr] = t;
    }
    CX_STAMP(7);
    // ---- cross-workgroup K reduction (conv2_kernel's hand-off: write-through partial tiles -> vmcnt(0) -> barrier -> relaxed
    //      ticket; the last arriver acquires and sums in slice order) — every wave publishes and later sums only its own rows ----
    if (A.KS > 1) {
        const size_t slot = (size_t)by * A.nrt + rg;
        const size_t toff = ((size_t)ct * 16) * 64 + lane;
        float *mine = A.part + (((size_t)kz * A.nby + by) * A.ntiles + tile) * 2048 + toff;
        if (tile_ok) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (rr < nr) {
                    float *dst = mine + (size_t)(kpart * nr + rr) * 64;
                    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(val[rr]) : "memory");
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int prev = __hip_atomic_fetch_add(A.counters + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (prev == A.KS - 1) ? 1 : 0;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(A.counters + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
        // every wave reads through its own L1: the acquire above invalidated the L1 of this CU (one per CU: all sixteen waves share it)
        const float *src0 = A.part + ((size_t)by * A.ntiles + tile) * 2048 + toff;
        const size_t zstride = (size_t)A.nby * A.ntiles * 2048;
        auto sum_slices = [&](auto nrc) {
            constexpr int NR = decltype(nrc)::value, ZB = 16 / NR;        // slices in flight together
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) val[rr] = 0.f;
            for (int z0 = 0; z0 < A.KS; z0 += ZB) {
                float pv[ZB][NR];
#pragma unroll
                for (int zz = 0; zz < ZB; ++zz) {
                    const int z = min(z0 + zz, A.KS - 1);
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) pv[zz][rr] = src0[(size_t)z * zstride + (size_t)(kpart * NR + rr) * 64];
                }
#pragma unroll
                for (int zz = 0; zz < ZB; ++zz)
                    if (z0 + zz < A.KS) {
#pragma unroll
                        for (int rr = 0; rr < NR; ++rr) val[rr] += pv[zz][rr];
                    }
            }
        };
        if (nr == 1) sum_slices(std::integral_constant<int, 1>());
        else if (nr == 2) sum_slices(std::integral_constant<int, 2>());
        else sum_slices(std::integral_constant<int, 4>());
    }
    CX_STAMP(8);
    // ---- epilogue of this wave's rows ----
    {
        const int ml = ct * 32 + (lane & 31);
        const bool mok = ml < M && tile_ok;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int co = tile * 32 + frag_row(kpart * nr + rr, lane);
            if (rr < nr && mok && co < A.Cout) {
                const float o = val[rr] * inv_sc + ((pre_b[rr] + (A.has_emb ? pre_e[rr] : 0.f)) + (A.has_res ? pre_r[rr] : 0.f));
                A.out[b_ep * A.out_bstride + (long)co * A.Lout + l_ep] = o;
                if constexpr (LF) {
                    // x0 prediction -> x_{t-1}, in place (this element of x is read and written by this thread only)
                    const long n = (long)A.B * A.Cout * A.Lout, e = ((long)b_ep * A.Cout + co) * A.Lout + l_ep;
                    const float xn = loop_update(lfv.sampler, lfv.clip, lfv.eta, lfrow, o, lfv.x[e], lfv.lp->noise[(long)(1 + lfk) * n + e]);
                    lfv.x[e] = xn;
                    if (lfv.lp->traj) lfv.lp->traj[(long)lfk * n + e] = xn;
                }
            }
        }
    }
    if constexpr (LF) {
        // the last workgroup to get here advances the loop counter: every workgroup of this launch that reads it has done so before
        __syncthreads();
        if (tid == 0) {
            const int prev = __hip_atomic_fetch_add(lfv.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == A.nby * A.nrt - 1) {
                __hip_atomic_store(lfv.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(lfv.step, lfk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#ifdef SURFD_C2_STAMPS
    CX_STAMP(9);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CX_STAMP(10);
    if (A.dbg && blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 11; ++i) A.dbg[i] = stamp_[i];
        A.dbg[15] = (long long)__builtin_readcyclecounter() - cyc0_;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
static int envx(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
static int lg2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Launches planned convolution `c` in the sixteen-wave latency form; returns 1 when the layer / shape is not covered (the caller
// continues with the four-wave form), 0 when launched, < 0 on error.  Same operand resolution as launch_conv2.
int launch_conv2x(surfd_unet *u, const ConvPlan &c, int B, int L, const ConvLaunchIO &io, hipStream_t st) {
    if (!c.f16_ok || !u->whf) return 1;
    ConvXArgs A;
    memset(&A, 0, sizeof(A));
    A.nseg = c.nseg; A.Cout = c.Cout; A.B = B;
    A.Lout = c.ds_out ? L / c.ds_out : 1;
    if (A.Lout < 1 || (A.Lout & (A.Lout - 1))) return 1;
    A.log2Lout = lg2(A.Lout);
    auto resolve = [&](const View &v, int ds, bool is_out, float *&ptr, long &bs) {
        const int len = ds ? L / ds : 1;
        if (v.buf >= 0) {
            ptr = u->buf_ptr[v.buf] + (long)v.choff * len;
            bs = (long)u->bufs[v.buf].C * len;
        } else if (is_out) { ptr = io.ext_out; bs = io.ext_out_bs; }
        else { ptr = const_cast<float *>(io.ext_in); bs = io.ext_in_bs; }
    };
    int Lin0 = 0, max_lsl = 1, max_blkp = 16;
    for (int s = 0; s < c.nseg; ++s) {
        const SegPlan &sp = c.seg[s];
        if (!sp.ds) return 1;
        SegX &S = A.seg[s];
        float *p; long bs;
        resolve(sp.src, sp.ds, false, p, bs);
        S.x = p; S.bstride = bs;
        S.C = sp.C; S.Lin = L / sp.ds;
        if (S.Lin < 4 || S.Lin > 64 || (S.Lin & (S.Lin - 1))) return 1;
        S.log2Lin = lg2(S.Lin);
        if (s == 0) Lin0 = S.Lin; else if (S.Lin != Lin0) return 1;
        S.taps = sp.taps; S.stride = sp.stride; S.ups = sp.ups; S.gn = sp.gn; S.act = sp.act;
        S.gamma = S.beta = u->vecs;
        if (sp.gn) { S.gamma = u->vecs + u->vec_off[sp.gnkey + ".weight"]; S.beta = u->vecs + u->vec_off[sp.gnkey + ".bias"]; }
        S.blk = c.blk[s]; S.blkp = c.blkp[s]; S.nblk = c.nblk[s]; S.k16_off = c.k16_off[s];
        S.gs = sp.gn ? sp.C / 32 : 8;
        if (S.gs > 64) return 1;
        S.lp = lg2(S.gs);
        S.ng = (S.blk + S.gs - 1) / S.gs;
        S.magic_ng = (unsigned)((0x100000000ULL + S.ng - 1) / S.ng);
        S.inv_cnt = 1.f / (float)(S.gs * S.Lin);
        const int q = S.log2Lin - 2;                 // log2 float4 per row
        S.ql = std::min(q, 6 - S.lp);
        S.lfu = q - S.ql;
        if (S.lfu > 2) return 1;                     // a group's row does not fit four float4 per lane (42 channels x 32 positions, 21 x 64): four-wave form
        const int lsl = sp.taps == 3 ? (sp.stride == 2 ? 2 * A.Lout + 1 : A.Lout + 2) : A.Lout;
        max_lsl = std::max(max_lsl, lsl);
        max_blkp = std::max(max_blkp, S.blkp);
    }
    A.Lsl = max_lsl;
    A.cs = max_blkp + 8;
    const int nch = c.nblk[0] + (c.nseg > 1 ? c.nblk[1] : 0);
    A.ntiles = ceil_div(c.Cout, 32);
    // ---- decomposition.  Batch rows per workgroup: the staged positions (nb * Lin) and the columns (nb * Lout) are both <= 64,
    //      every segment's units fit four float4 per thread, the slab fits its planes.  One column tile (<= 32 columns) where that
    //      keeps the launch within one workgroup per CU; K slices to fill the CUs; more row tiles per workgroup (fewer k-parts)
    //      where even one slice per workgroup has more workgroups than CUs. ----
    static const int nct_env = envx("SURFD_CONV2X_NCT", 0), rt_env = envx("SURFD_CONV2X_RT", 0), ks_env = envx("SURFD_CONV2X_KS", 0);
    static const int ks_max = envx("SURFD_CONV2_KSMAX", 16);
    const int cus = u->cu_budget;
    auto fits = [&](int nb) {
        if (nb < 1 || nb * Lin0 > 64 || nb * A.Lout > 64) return false;
        if ((size_t)nb * A.Lsl * A.cs > (size_t)CX_PLANE) return false;
        for (int s = 0; s < c.nseg; ++s) {
            const SegX &S = A.seg[s];
            const int upw = 64 >> (S.lp + S.ql);
            const int npass = ceil_div(S.ng * nb, 16 * upw);
            if ((npass << S.lfu) > 4) return false;
        }
        return true;
    };
    int nb_hi = std::min(B, 8);
    while (nb_hi > 1 && !fits(nb_hi)) --nb_hi;
    if (!fits(nb_hi)) return 1;
    int nb_lo = nb_hi;                               // the largest batch chunk of <= 32 columns
    while (nb_lo > 1 && nb_lo * A.Lout > 32) --nb_lo;
    const bool lo_ok = nb_lo * A.Lout <= 32;
    struct Cand { int nb, nct, RT, KS; long wgs; };
    auto make = [&](int nb, int RT) -> Cand {
        Cand k;
        k.nb = nb; k.nct = nb * A.Lout > 32 ? 2 : 1; k.RT = RT;
        const long base = (long)ceil_div(A.ntiles, RT) * ceil_div(B, nb);
        k.KS = (int)std::min<long>({(long)nch, (long)ks_max, std::max<long>(1, cus / base)});
        if (ks_env) k.KS = std::min(ks_env, nch);
        k.wgs = base * k.KS;
        return k;
    };
    Cand pick = make(lo_ok ? nb_lo : nb_hi, 1);
    if (pick.wgs > cus && nb_hi > pick.nb) pick = make(nb_hi, 1);
    if (pick.wgs > cus && pick.nct == 1) pick = make(pick.nb, 2);
    if (pick.wgs > cus && pick.nct * 2 <= 4 && pick.RT == 2) pick = make(pick.nb, 4);
    if (nct_env == 2 && nb_hi * A.Lout > 32) pick = make(nb_hi, pick.RT);
    if (nct_env == 1 && lo_ok) pick = make(nb_lo, pick.RT);
    if (rt_env && rt_env * pick.nct <= 4) pick = make(pick.nb, rt_env);
    if (pick.RT * pick.nct > 4) return 1;
    const int nb = pick.nb;
    A.bchunk = nb;
    A.RT = pick.RT; A.log2RT = lg2(pick.RT); A.log2nct = lg2(pick.nct); A.log2KP = 4 - A.log2RT - A.log2nct;
    A.nby = ceil_div(B, nb);
    A.nrt = ceil_div(A.ntiles, pick.RT);
    int KS = pick.KS;
    if ((size_t)KS * A.ntiles * A.nby * 2048 > u->part_floats || (long)A.nby * A.nrt > 8192) KS = 1;
    A.KS = KS;
    for (int s = 0; s < c.nseg; ++s) {
        SegX &S = A.seg[s];
        S.npass = ceil_div(S.ng * nb, 16 * (64 >> (S.lp + S.ql)));
    }
    size_t lds = (size_t)CX_PLANE * 2 * sizeof(_Float16);      // = 16 partial tiles of 4 KB
    A.off_flag = (int)lds;
    lds += 256;
    A.whf = u->whf + c.whf_off; A.KS16 = c.KS16;
    A.inv_sc = u->wsc_host[(size_t)c.sc_idx * 4 + 1];
    A.bias = u->vecs + c.bias_off;
    A.emb = A.bias; A.emb_bstride = 0; A.res = A.bias; A.res_bstride = 0; A.res_cstride = 1; A.res_lstride = 0;
    if (c.emb_off >= 0 && io.emb) {
        A.emb = io.emb + c.emb_off; A.step_ptr = io.step_ptr;
        A.emb_bstride = u->emb_shared ? 0 : io.emb_bs; A.emb_step_stride = u->emb_shared ? io.emb_bs : (long)B * io.emb_bs;
        if (u->emb_ingraph) { A.step_ptr = nullptr; A.emb_step_stride = 0; }
        A.has_emb = 1;
    }
    if (c.res.buf != -1) {
        float *p; long bs; resolve(c.res, c.ds_out, false, p, bs);
        A.res = p; A.res_bstride = bs; A.res_cstride = A.Lout; A.res_lstride = 1; A.has_res = 1;
    }
    { float *p; long bs; resolve(c.dst, c.ds_out, true, p, bs); A.out = p; A.out_bstride = bs; }
    A.part = u->part; A.counters = u->counters;
    A.sat = u->sat;
    const bool fuse_head = io.lf && c.dst.buf == -3;
    if (fuse_head) {
        if (io.ext_out_bs != (long)c.Cout * A.Lout) SURFD_FAIL(SURFD_ERR_ARG, "conv: fused posterior update needs a contiguous head output");
        A.lf = io.lf;
        if (io.lf_done) *io.lf_done = true;
    }
    A.dbg = nullptr;
    if (u->dbg && u->dbg_launch < 4096) {
        A.dbg = u->dbg + (size_t)(u->dbg_launch++) * 16;
        long long meta[4] = {c.Cout, c.seg[0].C + (c.nseg > 1 ? c.seg[1].C : 0), A.Lout * 100000LL + (long long)A.nrt * A.nby * KS, KS * 100 + nch};
        HIP_TRY(hipMemcpyAsync(A.dbg + 11, meta, sizeof(meta), hipMemcpyHostToDevice, st));
    }
    A.magic_nby = (unsigned)((0x100000000ULL + A.nby - 1) / A.nby);
    A.magic_ks = (unsigned)((0x100000000ULL + KS - 1) / KS);
    const int G = A.nrt * KS;
    A.magic_g = (unsigned)((0x100000000ULL + G - 1) / G);
    if ((long)G * A.nby >= 65536) return 1;
    dim3 grid((unsigned)(G < 8 ? G * A.nby : 8 * ceil_div(G, 8) * A.nby));
    // this launch's shape for the launch before it (weight prefetch ahead of the wide form, conv_f16x2.hip): a sixteen-wave launch
    // leaves no record a wide launch could use
    c.rec.gen = -1;
    if (A.lf) hipLaunchKernelGGL((conv2x_kernel<true>), grid, dim3(1024), lds, st, A);
    else hipLaunchKernelGGL((conv2x_kernel<false>), grid, dim3(1024), lds, st, A);
    LAUNCH_CHECK();
    return SURFD_OK;
}

int conv2x_set_attributes() {
    const int max_lds = 160 * 1024;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2x_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2x_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    return SURFD_OK;
}

}  // namespace surfd
