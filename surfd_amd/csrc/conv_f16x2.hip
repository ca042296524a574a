// Split-fp16 ("f16x2") convolution kernel of the latent denoiser (reference models/openaimodel.py:255-275
// ResBlock, :318-324 AttentionBlock projections, :134-160 Downsample, :91-119 Upsample, :682-686 head) for
// gfx950 — the kernel the reverse loop spends its time in.
//
// Same implicit GEMM as conv_kernel (unet.hip): Out[Cout x (b,l)] = W[Cout x K] * im2col(act(GN(x))), K ordered
// (segment, K block, tap, channel), with GroupNorm32 + SiLU fused into the LDS staging, the 1x1 skip conv as a
// second K segment and bias / timestep-embedding / residual in the epilogue.  What differs:
//
//  * arithmetic: every fp32 operand is split into two fp16 terms (x = xh + xl, |x - xh - xl| <= 2^-22 |x|);
//    the three significant products xh*wh + xh*wl + xl*wh run on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate)
//    with fp32 accumulation in three independent accumulators.  Weights are multiplied by one power of two per
//    layer (max |W| * SC in [256, 512)) before the split so the low terms stay normal; SC is folded back exactly
//    in the epilogue.  Operands are clamped to +-65504 (fp16 range); a device counter records every workgroup
//    that had to clamp (surfd_unet_saturation_count) — precision mode 0 runs the exact fp32 kernel instead.
//  * ONE small kernel for every layer shape (operand length 4..32; a second instantiation for 64): the fp32
//    family is 7 x 25-50 KB of once-executed straight-line code that misses the instruction cache on every
//    launch of the latency-bound loop.
//  * the weight stream is a register ring that runs ACROSS K blocks (weights of the next block are requested
//    during the last MFMAs of the current one), and the raw operand of the next K block is requested before the
//    MFMAs of the current one.
//  * blockIdx -> (tile, batch chunk, K slice) is XCD-aware: all batch chunks that stream the same weight slice
//    run on one XCD (block b runs on XCD b % 8), so a slice leaves HBM once and is re-read from that XCD's L2.
//  * two work decompositions of the same arithmetic.  Latency form (WT = false; one narrow loop alone on the chip):
//    a workgroup owns ONE 32-row tile, its four waves split the K slice and meet in LDS.  Wide form (WT = true;
//    surfd_unet_set_wide, loops over tens of latents): a workgroup owns FOUR row tiles, one per wave, each wave
//    running the whole K slice against the one staged operand — the GroupNorm / SiLU / split staging, which is
//    what a workgroup spends its life on, is done once for 128 output rows instead of once per 32, and the K
//    split depends on the layer only, so a latent's result does not depend on the batch it rides in.
#include "common.h"
#include "unet_api.h"
#include "unet_plan.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace surfd {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16x8 __attribute__((address_space(1))) gf16x8;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Seg2 {
    const float *x;        // source view (channel offset applied)
    long bstride;          // floats between batch entries
    const float *gamma, *beta;
    int C, Lin, log2Lin;
    int taps, stride, ups, gn, act;
    int blk, blkp, nblk;   // channels per K block (= staged chunk), padded to 16, number of blocks
    int k16_off;           // first k16 step of the segment
    int gs;                // channels per GroupNorm group
    int log2P, gsm;        // thread <-> channel map of the staging: lane = group slot * 2^log2P + position, gsm channels per slot
                           // (segments with GroupNorm: gsm = gs, 2^log2P = gs rounded up to a power of two; others: one slot of 64)
};

struct Conv2Args {
    Seg2 seg[2];
    int nseg;
    const _Float16 *whf;   // [ntiles][KS16][2 planes][64 lanes][8]
    int KS16;
#if defined(SURFD_C2_PROBE) || defined(SURFD_C2_DBG_POISON)      // developer builds: the two fields the kernel never reads carry the probe buffer / LDS size (the struct fills its eight lines)
    unsigned long long *probe;   // [workgroup][8] phase checksums of this launch (tools/probe_phases.py) or null
#define C2_LDS_BYTES plane
#else
    const float *sc;       // {SC, 1/SC} on the device (weight preparation); the kernel takes 1/SC by value:
#endif
    float inv_sc;          // a dependent scalar load in front of everything else cost ~1 us per workgroup
    const float *bias;     // [Cout]
    const float *emb;      // emb[b * emb_bstride + co] or null
    long emb_bstride;
    const float *res;      // res[b * res_bstride + co * res_cstride + l * res_lstride]
    long res_bstride;
    int res_cstride, res_lstride, has_res, has_emb;   // unused emb / res point at the bias vector with zero strides
    float *out;
    long out_bstride;
    int Cout, Lout, log2Lout, B;
    int bchunk, Lsl, cs;   // batch entries per workgroup, slab positions per batch entry, slab row stride (halfs)
    int plane;             // halfs between the high and the low plane of the slab
    int off_ex, off_red;   // byte offsets of the GroupNorm exchange area / reduction scratch in LDS
    int ntiles, nby, KS;
    int nrt;               // row groups: ntiles (latency form) or ceil(ntiles / 4) (wide form)
    int nhalf;             // wide form, 64 output positions: 2 = a workgroup owns one 32-column half of ONE sample (nby counts halves)
    unsigned magic_nby, magic_ks, magic_g;   // ceil(2^32 / d): x / d == umulhi(x, magic) for x < 2^16 (d = nby, KS, nrt * KS)
    float *part;           // split-K partial tiles [KS][nby][ntiles][part_stride]
    int part_stride;
    int *counters;         // [nby][ntiles], zero between launches
    const int *step_ptr;   // device loop counter (embedding rows advance by emb_step_stride per step) or null
    long emb_step_stride;
    unsigned *sat;         // saturation counter
    const LoopFuse *lf;    // device; head convolution inside the graph-replayed loop (LF instantiations): posterior update +
                           // loop-counter advance in the epilogue.  A pointer, and its own instantiation: the other 83 launches of an
                           // evaluation must not carry a byte or a register of it
    long long *dbg;        // -DSURFD_C2_STAMPS builds: 16 phase stamps (100 MHz ticks) of workgroup 0
    // weight prefetch ahead (SURFD_C2_PFN): the NEXT convolution's packed weights and how its launch will cut them up
    const char *pf_w;      // null: nothing to request
    int pf_ntiles, pf_KS16, pf_KS, pf_G, pf_nblk0, pf_it0, pf_nch, pf_it1;
    int pf_log2tpg, pf_log2bps, pf_log2lpb;   // tiles per group (1 / 4), K blocks per slice and 128-byte lines per K block, rounded up to powers of two
    unsigned pf_magic_ks;
    int off_pf;            // unused since round 6 (was: landing area of an LDS-DMA form of the request); kept so that the argument block keeps its eight lines
};

#ifndef SURFD_C2_EPI_LATE
#define SURFD_C2_EPI_LATE 1
#endif
#ifndef SURFD_C2_BPIPE
#define SURFD_C2_BPIPE 0
#endif
#ifndef SURFD_C2_LEAN_U
#define SURFD_C2_LEAN_U 2
#endif
#ifndef SURFD_C2_ZB
#define SURFD_C2_ZB 4                  // split-K partial tiles in flight together in the last arriver's reduction
#endif
#ifndef SURFD_C2_LAT_D
#define SURFD_C2_LAT_D 2
#endif
#ifndef SURFD_C2_DIST
#define SURFD_C2_DIST 1                // latency form: k-part reduction, split-K hand-off and epilogue distributed over the k-part waves (see the kernel)
#endif
#ifndef SURFD_C2_DEEP_D
#define SURFD_C2_DEEP_D 3
#endif
// ring stages: 2 (with 4 k16 steps per stage 16 x 1 KB fragments in flight per wave).  With the epilogue operands requested after
// the K loop the two-workgroups-per-CU forms have room for a third stage (232 instead of 200 registers): measured SLOWER in the
// latency form (SURFD_C2_LAT_D=3: 1.385 against 1.352 ms per evaluation at 8 latents — a wave's k-part there is 10-11 k16 steps,
// 8 of them are in flight with two stages, and the third stage's 8 loads sit in front of the wait for the operand); the wide
// form's two-per-CU kernel (one wave = the whole K slice, 42 k16 steps per 224-channel three-tap block) runs SURFD_C2_DEEP_D
#ifndef SURFD_C2_LEAN_WAVES
#define SURFD_C2_LEAN_WAVES 3          // waves per SIMD (= workgroups per CU) the lean form is compiled for: 3 -> 168 VGPRs, no spills
#endif
constexpr int C2_PLANE_NT2 = 13056;    // two column tiles per wave: 96 positions x (128 + 8) halfs; 2 planes + flag = 52 240 B, three workgroups per CU
#ifndef SURFD_C2_PLANE_LEAN
#define SURFD_C2_PLANE_LEAN 11264
#endif
// halfs per fp16 plane of the slab in the lean form.  Round 4: 10 112 (2 planes + flag = 40 464 B: room for a fourth workgroup per CU
// that the 168-register build never uses); 11 264 holds EIGHT samples of the 4-position level (8 x 6 x 232 = 11 136 halfs)
// instead of seven: 10 full 32-column chunks per 80 latents instead of 11 + a tail of three; 3 x 45 072 B of the CU's 160 KB
constexpr int C2_PLANE_LEAN = SURFD_C2_PLANE_LEAN;

// SiLU of the operand staging: x * 1 / (1 + e^-x) with the hardware reciprocal (v_rcp_f32, 1 ulp).  __frcp_rn is an IEEE
// division on this target — v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup: ten instructions per value, a
// quarter of everything the staging executes (read back from the code object) — for half an ulp nobody downstream can see:
// the value is split into two fp16 terms with 2^-22 relative error right after.
#ifndef SURFD_C2_FAST_RCP
#define SURFD_C2_FAST_RCP 1
#endif
__device__ __forceinline__ float silu2(float v) {
#if SURFD_C2_FAST_RCP
    return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
#else
    return v * __frcp_rn(1.f + __expf(-v));
#endif
}

// four values at once: the multiplications and the addition two per instruction, the four transcendental pairs independent of
// each other (the scalar form compiled to one dependent chain per value on one pair of registers: 11 instructions, two of them
// quarter-rate, nothing to overlap them with on a SIMD that holds one wave).  Same operations per element as silu2: same bits.
__device__ __forceinline__ void silu2_x4(f32x4 &v) {
#if SURFD_C2_FAST_RCP
    const f32x2 c = {-0x1.715476p+0f, -0x1.715476p+0f}, one = {1.f, 1.f};
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const f32x2 ta = a * c, tb = b * c;
    f32x2 ea = {__builtin_amdgcn_exp2f(ta[0]), __builtin_amdgcn_exp2f(ta[1])}, eb = {__builtin_amdgcn_exp2f(tb[0]), __builtin_amdgcn_exp2f(tb[1])};
    ea = one + ea; eb = one + eb;
    const f32x2 ra = {__builtin_amdgcn_rcpf(ea[0]), __builtin_amdgcn_rcpf(ea[1])}, rb = {__builtin_amdgcn_rcpf(eb[0]), __builtin_amdgcn_rcpf(eb[1])};
    const f32x2 ya = a * ra, yb = b * rb;
    v[0] = ya[0]; v[1] = ya[1]; v[2] = yb[0]; v[3] = yb[1];
#else
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = silu2(v[k]);
#endif
}

// Kernel-argument prefetch.  Conv2Args is 6-7 cache lines of kernarg segment; with 106 SGPRs the compiler fetches it in
// seven batches, each behind an s_waitcnt lgkmcnt(0) and each touching lines the scalar cache has not seen in this launch
// (the segment was last read one graph replay = 553 MB of weight stream ago): seven dependent misses before the first
// operand load can be issued.  One dword of every line requested by the first instructions of the wave turns them into one
// miss and seven hits.
#ifndef SURFD_C2_KAPF
#define SURFD_C2_KAPF 1
#endif
template <int BYTES>
__device__ __forceinline__ void c2_kernarg_prefetch() {
#if SURFD_C2_KAPF
    static_assert(BYTES > 0x1c0 && BYTES <= 0x200, "kernel-argument prefetch covers eight 64-byte lines");
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    int d0, d1, d2, d3, d4, d5, d6, d7;
    // the wait is part of the statement: the destinations are dead when it ends (no compiler-assigned value can be hit by a late
    // return), and it costs nothing — the compiler's own first batch would wait for the same miss two instructions later
    asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\ts_load_dword %3, %8, 0xc0\n\t"
                 "s_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\ts_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7) : "s"(ka));
#endif
}

// split-K partial tiles are read with plain loads behind the last arriver's agent-scope acquire (round 5 tried sc1 loads without
// the acquire: an sc1 load bypasses this CU's L1 but not a copy of the line an EARLIER launch's reduction left in this XCD's L2 —
// the sequential and the pipelined run stopped agreeing bit for bit; removed in round 6, profiles/r05_loop_experiments.md section 3)
__device__ __forceinline__ f32x4 c2_load_partial(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// GroupNorm statistics in the wave (SURFD_C2_GNW = 1, the default since round 6).  The staging maps thread <-> channel so that
// every GroupNorm group of a K block sits in ONE wave, in a slot of 2^log2P consecutive lanes (7 channels -> 8 lanes, 14 -> 16,
// 21 / 28 -> 32, 42 / 56 -> 64; the unused lanes of a slot stage nothing).  A group's sums are a butterfly over the slot's lanes:
// DPP for 2, 4, 8 and 16 lanes, v_permlane16_swap / v_permlane32_swap for 32 and 64 — every lane of a slot ends with the same
// bits (each step adds the same two numbers on both sides).  No LDS exchange, no barrier between the operand and the slab.
//
// History (profiles/r05_loop_experiments.md section 3, profiles/r06_conv2_instability.md).  Round 5 built this form, found it NOT
// bit-stable at three workgroups per CU and left it off; the same round saw the LDS-exchange form (SURFD_C2_GNW = 0: per-(row,
// channel) means through an LDS exchange area, an 8-lane ds_bpermute combine per (row, group), results through LDS again) give
// timing-dependent wrong results in the VEC = 16 wide kernel.  Round 6 identified what goes wrong in BOTH: one quantity, the
// 1/sigma of one or two groups of one sample, short by about one lane's term of the second reduction while the mean is right (a
// regression of the observed error on d out / d (1/sigma_g) explains 100.00 % of it) — a lane of the all-reduce worked with a
// neighbour's value from before that neighbour's last add.  In this form one wait state between every add and the cross-lane
// read of its result (SURFD_C2_GNPAD, below) removes it completely (0 differing evaluations in > 400 where every evaluation
// differed before, two and three workgroups per CU, both column-tile forms, with the weight prefetch compiled into every
// instantiation); in the LDS-exchange form the same padding does NOT, so that form is retired: it only compiles with
// -DSURFD_ALLOW_UNSAFE_VARIANTS.
#ifndef SURFD_C2_GNW
#define SURFD_C2_GNW 1
#endif
template <int CTRL>
__device__ __forceinline__ float c2_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}
// N independent sums, step by step (the N chains of dependent DPP / bpermute operations overlap); x must be zero in the lanes
// of a slot that hold no channel
// SURFD_C2_GNPAD=n (round 6): n wait states between every add of the GroupNorm reductions and the cross-lane read (DPP,
// ds_bpermute, v_permlane*_swap) of its result.  The statistics are the one quantity that goes wrong in the timing-dependent
// failures of round 5 (profiles/r06_conv2_instability.md: 1/sigma of one or two groups short by about one lane's term, mean
// right): a lane of the all-reduce worked with a neighbour's value from BEFORE that neighbour's last add.  The compiler's own
// padding (s_nop 1 in front of a DPP read, nothing in front of ds_bpermute's data read) assumes the add's result reaches the
// register file a fixed number of cycles after issue.
#ifndef SURFD_C2_GNPAD
#define SURFD_C2_GNPAD 4           // measured: 1, 2, 4 and 8 are all bit-stable where 0 is not (profiles/r06_conv2_instability.md); 4 costs nothing measurable
#endif
__device__ __forceinline__ void c2_gnpad(float &x) {
#if SURFD_C2_GNPAD > 0
    asm volatile(".rept %1\n\ts_nop 0\n\t.endr" : "+v"(x) : "i"(SURFD_C2_GNPAD));
#else
    (void)x;
#endif
}
// x[lane] + x[lane ^ W] for W = 16 / 32 on the vector ALU (gfx950: v_permlane16_swap / v_permlane32_swap exchange the odd
// 16-lane rows / the upper half of the first operand with the even rows / the lower half of the second: with both operands x
// the two results hold, in every lane, the two partners' values) — no LDS pipe, no lgkmcnt
template <int W>
__device__ __forceinline__ float c2_swap_sum(float x) {
    const unsigned u = __float_as_uint(x);
    if constexpr (W == 16) { const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); return __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    else { const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); return __uint_as_float(r[0]) + __uint_as_float(r[1]); }
}
template <int N>
__device__ __forceinline__ void c2_slot_sum(float (&x)[N], int log2P) {
#if SURFD_C2_GNPAD > 0
#pragma unroll
    for (int i = 0; i < N; ++i) c2_gnpad(x[i]);          // the leaves were written by the instruction before
#endif
    if (log2P > 0) {         // quad_perm [1,0,3,2]: lane ^ 1
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] += c2_dpp<0xB1>(x[i]); c2_gnpad(x[i]); }
    }
    if (log2P > 1) {         // quad_perm [2,3,0,1]: lane ^ 2
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] += c2_dpp<0x4E>(x[i]); c2_gnpad(x[i]); }
    }
    if (log2P > 2) {         // row_half_mirror: the other quad of the 8
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] += c2_dpp<0x141>(x[i]); c2_gnpad(x[i]); }
    }
    if (log2P > 3) {         // row_mirror: the other half of the 16
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] += c2_dpp<0x140>(x[i]); c2_gnpad(x[i]); }
    }
    if (log2P > 4) {
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] = c2_swap_sum<16>(x[i]); c2_gnpad(x[i]); }
    }
    if (log2P > 5) {
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] = c2_swap_sum<32>(x[i]); c2_gnpad(x[i]); }
    }
}

__device__ __forceinline__ void lds_bar() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Weight prefetch ahead (SURFD_C2_PFN).  A wave of the wide form runs its whole K slice against a ring of four k16 steps; every
// step waits for a round trip, and the weights come from HBM every time (553 MB per evaluation against 32 MB of L2): 214 ns per
// k16 step against 40 ns of matrix time — and 136 ns when the same launch is repeated and finds its weights in the L2s
// (profiles/r05_loop_experiments.md, SURFD_CONV2_TWICE).  So every workgroup requests its share of the NEXT convolution's
// weights — the slices that launch's workgroups on THIS XCD will stream (block b runs on XCD b % 8; the next launch's group
// g = (row group, K slice) runs on XCD g % 8, or everywhere when it has fewer than 8 groups): C2_PFN_N four-byte loads per
// thread, one per 128-byte line, issued right behind this block's own first requests and "used" by an empty statement at the
// very end.  Ordinary, unconditional loads: the compiler's vmcnt arithmetic stays exact (a request it does not know about in
// front of the ring would make every ring wait stricter; requests at the exit — the first form of this — made every workgroup
// end 1.5 us later: S_ENDPGM waits for them).  The XCD's line set is indexed uniformly: (group of this XCD, tile of the group,
// K block of the slice, line) with every block padded to the longer segment's length; padding and missing blocks re-read the
// first line.
#ifndef SURFD_C2_PFN
#define SURFD_C2_PFN 1
#endif
// SURFD_C2_PFN_FORMS: 1 (default since round 6) = every wide instantiation carries the request; 0 = only the lean ones (three
// workgroups per CU).  Round 5 compiled it out of the VEC = 16 wide kernel (the 64-position level of the L = 64 models: C4 / C5)
// because that kernel's results were then not bit-stable and left the 1e-4 bound — which round 6 traced to the GroupNorm
// statistics of the LDS-exchange form (above), not to the request: with the in-wave statistics the same build is exact and
// repeatable (profiles/r06_conv2_instability.md) and the 64-position level gets its share of the prefetch (L = 64, two loops of
// 80: 34.6 -> 33.0 us per evaluation and latent).
#ifndef SURFD_C2_PFN_FORMS
#define SURFD_C2_PFN_FORMS 1
#endif
#ifndef SURFD_C2_PFN_N
#define SURFD_C2_PFN_N 1
#endif
constexpr int C2_PFN_N = SURFD_C2_PFN_N;
// (Round 5 also tried the request as an LDS-DMA load — no destination register: slower, and M0's 16-bit LDS base wraps in the
// two-per-CU forms — and developer variants without the load / with the word awaited at once; removed in round 6.)
__device__ __forceinline__ void c2_prefetch_next(const Conv2Args &A, int tid, unsigned (&sink)[C2_PFN_N]) {
#if SURFD_C2_PFN
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int x = bid & 7, i = bid >> 3, nx = (nwg + 7 - x) >> 3;            // this workgroup is number i of nx on XCD x
    const bool all = A.pf_G < 8;                                              // fewer than 8 groups: every XCD streams every group
    const int ngx = all ? A.pf_G : (A.pf_G + 7 - x) >> 3;                     // groups whose weights this XCD will read
    // index space, powers of two throughout (shifts, no divisions in front of the operand wait):
    //   ((group of this XCD * tiles per group + tile) * blocks per slice + block) * lines per block + line
    const int sl = A.pf_log2lpb, sb = A.pf_log2bps, st = A.pf_log2tpg;
    const char *w = A.pf_w;                                                   // never null: the host points it at this launch's own weights when there is nothing to request
    long off[C2_PFN_N];
#pragma unroll
    for (int k = 0; k < C2_PFN_N; ++k) off[k] = 0;
    if (A.pf_ntiles > 0) {                                                    // wave-uniform; the loads below are issued either way
#pragma unroll
        for (int k = 0; k < C2_PFN_N; ++k) {
            const unsigned p = (unsigned)(i + k * nx) * 256u + (unsigned)tid;
            const int l = (int)(p & ((1u << sl) - 1u));
            const unsigned r1 = p >> sl;
            const int j = (int)(r1 & ((1u << sb) - 1u));
            const unsigned r2 = r1 >> sb;
            const int tl = (int)(r2 & ((1u << st) - 1u)), q = (int)(r2 >> st);
            const int g = all ? q : x + 8 * q;
            const int rg = A.pf_KS == 1 ? g : (int)__umulhi((unsigned)g, A.pf_magic_ks), kz = g - rg * A.pf_KS;      // g / KS: host-made reciprocal, g < 2^16
            const int tile = (rg << st) + tl, ch = kz + j * A.pf_KS;
            const int it = ch < A.pf_nblk0 ? A.pf_it0 : A.pf_it1;
            const long k16 = ch < A.pf_nblk0 ? (long)ch * A.pf_it0 : (long)A.pf_nblk0 * A.pf_it0 + (long)(ch - A.pf_nblk0) * A.pf_it1;
            const bool ok = q < ngx && tile < A.pf_ntiles && ch < A.pf_nch && l < it * 16;
            off[k] = ok ? (((long)tile * A.pf_KS16 + k16) * 2048 + (long)l * 128) : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < C2_PFN_N; ++k) sink[k] = *reinterpret_cast<const unsigned *>(w + off[k]);
#else
#pragma unroll
    for (int k = 0; k < C2_PFN_N; ++k) sink[k] = 0u;
#endif
}

// Developer aids of the round-6 hunt for the "second workgroup on a CU" instability (profiles/r06_conv2_instability.md).
// -DSURFD_C2_PROBE: every workgroup adds order-sensitive checksums of its phases — raw operand, GroupNorm mean / scale, staged
// values, the slab read back from LDS, weight fragments consumed, accumulators, epilogue operands, stored values — to eight
// 64-bit words of a probe buffer (SURFD_CONV2_PROBE_PTR = device address, indexed by conv op id); two evaluations of the
// same input are compared word by word: the first (launch, phase, workgroup) that differs names where the bits change.
#ifdef SURFD_C2_PROBE
#define C2_PROBE_ADD(slot, h) do { if (A.probe) atomicAdd(A.probe + (size_t)blockIdx.x * 8 + (slot), (unsigned long long)(h)); } while (0)
#else
#define C2_PROBE_ADD(slot, h) do { } while (0)
#endif
// -DSURFD_C2_DBG_POISON=1 (zeros) / =2 (0x7fc00000, a quiet NaN): the first instructions of every wave write the pattern to
// v1..v255 and s4..s99 (v0 = thread id, s[0:1] = kernel arguments, s2 = workgroup id are live) and the whole dynamic LDS
// allocation — a value read before it is written can no longer be what another wave left behind.
#if defined(SURFD_C2_DBG_POISON) && SURFD_C2_DBG_POISON
// (v160..v167 of the lean form and v248..v255 of the others stay out: the compiler needs a register across the statement for its
//  SGPR spill lanes; s32 / s33 are reserved)
template <bool HI>
__device__ __forceinline__ void c2_poison_registers() {
    constexpr unsigned pat = SURFD_C2_DBG_POISON == 2 ? 0x7fc00000u : 0u;
    if constexpr (HI) asm volatile(
        "v_mov_b32 v160, %0\n\tv_mov_b32 v161, %0\n\tv_mov_b32 v162, %0\n\tv_mov_b32 v163, %0\n\tv_mov_b32 v164, %0\n\tv_mov_b32 v165, %0\n\t"
        "v_mov_b32 v166, %0\n\tv_mov_b32 v167, %0\n\tv_mov_b32 v168, %0\n\tv_mov_b32 v169, %0\n\tv_mov_b32 v170, %0\n\tv_mov_b32 v171, %0\n\t"
        "v_mov_b32 v172, %0\n\tv_mov_b32 v173, %0\n\tv_mov_b32 v174, %0\n\tv_mov_b32 v175, %0\n\tv_mov_b32 v176, %0\n\tv_mov_b32 v177, %0\n\t"
        "v_mov_b32 v178, %0\n\tv_mov_b32 v179, %0\n\tv_mov_b32 v180, %0\n\tv_mov_b32 v181, %0\n\tv_mov_b32 v182, %0\n\tv_mov_b32 v183, %0\n\t"
        "v_mov_b32 v184, %0\n\tv_mov_b32 v185, %0\n\tv_mov_b32 v186, %0\n\tv_mov_b32 v187, %0\n\tv_mov_b32 v188, %0\n\tv_mov_b32 v189, %0\n\t"
        "v_mov_b32 v190, %0\n\tv_mov_b32 v191, %0\n\tv_mov_b32 v192, %0\n\tv_mov_b32 v193, %0\n\tv_mov_b32 v194, %0\n\tv_mov_b32 v195, %0\n\t"
        "v_mov_b32 v196, %0\n\tv_mov_b32 v197, %0\n\tv_mov_b32 v198, %0\n\tv_mov_b32 v199, %0\n\tv_mov_b32 v200, %0\n\tv_mov_b32 v201, %0\n\t"
        "v_mov_b32 v202, %0\n\tv_mov_b32 v203, %0\n\tv_mov_b32 v204, %0\n\tv_mov_b32 v205, %0\n\tv_mov_b32 v206, %0\n\tv_mov_b32 v207, %0\n\t"
        "v_mov_b32 v208, %0\n\tv_mov_b32 v209, %0\n\tv_mov_b32 v210, %0\n\tv_mov_b32 v211, %0\n\tv_mov_b32 v212, %0\n\tv_mov_b32 v213, %0\n\t"
        "v_mov_b32 v214, %0\n\tv_mov_b32 v215, %0\n\tv_mov_b32 v216, %0\n\tv_mov_b32 v217, %0\n\tv_mov_b32 v218, %0\n\tv_mov_b32 v219, %0\n\t"
        "v_mov_b32 v220, %0\n\tv_mov_b32 v221, %0\n\tv_mov_b32 v222, %0\n\tv_mov_b32 v223, %0\n\tv_mov_b32 v224, %0\n\tv_mov_b32 v225, %0\n\t"
        "v_mov_b32 v226, %0\n\tv_mov_b32 v227, %0\n\tv_mov_b32 v228, %0\n\tv_mov_b32 v229, %0\n\tv_mov_b32 v230, %0\n\tv_mov_b32 v231, %0\n\t"
        "v_mov_b32 v232, %0\n\tv_mov_b32 v233, %0\n\tv_mov_b32 v234, %0\n\tv_mov_b32 v235, %0\n\tv_mov_b32 v236, %0\n\tv_mov_b32 v237, %0\n\t"
        "v_mov_b32 v238, %0\n\tv_mov_b32 v239, %0\n\tv_mov_b32 v240, %0\n\tv_mov_b32 v241, %0\n\tv_mov_b32 v242, %0\n\tv_mov_b32 v243, %0\n\t"
        "v_mov_b32 v244, %0\n\tv_mov_b32 v245, %0\n\tv_mov_b32 v246, %0\n\tv_mov_b32 v247, %0\n\t"
        "s_nop 0"
        :: "s"(pat)
        :
          "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183",
          "v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207",
          "v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231",
          "v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247");
    asm volatile(
        "v_mov_b32 v1, %0\n\tv_mov_b32 v2, %0\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\tv_mov_b32 v6, %0\n\t"
        "v_mov_b32 v7, %0\n\tv_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\t"
        "v_mov_b32 v13, %0\n\tv_mov_b32 v14, %0\n\tv_mov_b32 v15, %0\n\tv_mov_b32 v16, %0\n\tv_mov_b32 v17, %0\n\tv_mov_b32 v18, %0\n\t"
        "v_mov_b32 v19, %0\n\tv_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\tv_mov_b32 v24, %0\n\t"
        "v_mov_b32 v25, %0\n\tv_mov_b32 v26, %0\n\tv_mov_b32 v27, %0\n\tv_mov_b32 v28, %0\n\tv_mov_b32 v29, %0\n\tv_mov_b32 v30, %0\n\t"
        "v_mov_b32 v31, %0\n\tv_mov_b32 v32, %0\n\tv_mov_b32 v33, %0\n\tv_mov_b32 v34, %0\n\tv_mov_b32 v35, %0\n\tv_mov_b32 v36, %0\n\t"
        "v_mov_b32 v37, %0\n\tv_mov_b32 v38, %0\n\tv_mov_b32 v39, %0\n\tv_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\t"
        "v_mov_b32 v43, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\tv_mov_b32 v46, %0\n\tv_mov_b32 v47, %0\n\tv_mov_b32 v48, %0\n\t"
        "v_mov_b32 v49, %0\n\tv_mov_b32 v50, %0\n\tv_mov_b32 v51, %0\n\tv_mov_b32 v52, %0\n\tv_mov_b32 v53, %0\n\tv_mov_b32 v54, %0\n\t"
        "v_mov_b32 v55, %0\n\tv_mov_b32 v56, %0\n\tv_mov_b32 v57, %0\n\tv_mov_b32 v58, %0\n\tv_mov_b32 v59, %0\n\tv_mov_b32 v60, %0\n\t"
        "v_mov_b32 v61, %0\n\tv_mov_b32 v62, %0\n\tv_mov_b32 v63, %0\n\tv_mov_b32 v64, %0\n\tv_mov_b32 v65, %0\n\tv_mov_b32 v66, %0\n\t"
        "v_mov_b32 v67, %0\n\tv_mov_b32 v68, %0\n\tv_mov_b32 v69, %0\n\tv_mov_b32 v70, %0\n\tv_mov_b32 v71, %0\n\tv_mov_b32 v72, %0\n\t"
        "v_mov_b32 v73, %0\n\tv_mov_b32 v74, %0\n\tv_mov_b32 v75, %0\n\tv_mov_b32 v76, %0\n\tv_mov_b32 v77, %0\n\tv_mov_b32 v78, %0\n\t"
        "v_mov_b32 v79, %0\n\tv_mov_b32 v80, %0\n\tv_mov_b32 v81, %0\n\tv_mov_b32 v82, %0\n\tv_mov_b32 v83, %0\n\tv_mov_b32 v84, %0\n\t"
        "v_mov_b32 v85, %0\n\tv_mov_b32 v86, %0\n\tv_mov_b32 v87, %0\n\tv_mov_b32 v88, %0\n\tv_mov_b32 v89, %0\n\tv_mov_b32 v90, %0\n\t"
        "v_mov_b32 v91, %0\n\tv_mov_b32 v92, %0\n\tv_mov_b32 v93, %0\n\tv_mov_b32 v94, %0\n\tv_mov_b32 v95, %0\n\tv_mov_b32 v96, %0\n\t"
        "v_mov_b32 v97, %0\n\tv_mov_b32 v98, %0\n\tv_mov_b32 v99, %0\n\tv_mov_b32 v100, %0\n\tv_mov_b32 v101, %0\n\tv_mov_b32 v102, %0\n\t"
        "v_mov_b32 v103, %0\n\tv_mov_b32 v104, %0\n\tv_mov_b32 v105, %0\n\tv_mov_b32 v106, %0\n\tv_mov_b32 v107, %0\n\tv_mov_b32 v108, %0\n\t"
        "v_mov_b32 v109, %0\n\tv_mov_b32 v110, %0\n\tv_mov_b32 v111, %0\n\tv_mov_b32 v112, %0\n\tv_mov_b32 v113, %0\n\tv_mov_b32 v114, %0\n\t"
        "v_mov_b32 v115, %0\n\tv_mov_b32 v116, %0\n\tv_mov_b32 v117, %0\n\tv_mov_b32 v118, %0\n\tv_mov_b32 v119, %0\n\tv_mov_b32 v120, %0\n\t"
        "v_mov_b32 v121, %0\n\tv_mov_b32 v122, %0\n\tv_mov_b32 v123, %0\n\tv_mov_b32 v124, %0\n\tv_mov_b32 v125, %0\n\tv_mov_b32 v126, %0\n\t"
        "v_mov_b32 v127, %0\n\tv_mov_b32 v128, %0\n\tv_mov_b32 v129, %0\n\tv_mov_b32 v130, %0\n\tv_mov_b32 v131, %0\n\tv_mov_b32 v132, %0\n\t"
        "v_mov_b32 v133, %0\n\tv_mov_b32 v134, %0\n\tv_mov_b32 v135, %0\n\tv_mov_b32 v136, %0\n\tv_mov_b32 v137, %0\n\tv_mov_b32 v138, %0\n\t"
        "v_mov_b32 v139, %0\n\tv_mov_b32 v140, %0\n\tv_mov_b32 v141, %0\n\tv_mov_b32 v142, %0\n\tv_mov_b32 v143, %0\n\tv_mov_b32 v144, %0\n\t"
        "v_mov_b32 v145, %0\n\tv_mov_b32 v146, %0\n\tv_mov_b32 v147, %0\n\tv_mov_b32 v148, %0\n\tv_mov_b32 v149, %0\n\tv_mov_b32 v150, %0\n\t"
        "v_mov_b32 v151, %0\n\tv_mov_b32 v152, %0\n\tv_mov_b32 v153, %0\n\tv_mov_b32 v154, %0\n\tv_mov_b32 v155, %0\n\tv_mov_b32 v156, %0\n\t"
        "v_mov_b32 v157, %0\n\tv_mov_b32 v158, %0\n\tv_mov_b32 v159, %0\n\t"
        "s_mov_b32 s4, %0\n\ts_mov_b32 s5, %0\n\ts_mov_b32 s6, %0\n\ts_mov_b32 s7, %0\n\ts_mov_b32 s8, %0\n\ts_mov_b32 s9, %0\n\t"
        "s_mov_b32 s10, %0\n\ts_mov_b32 s11, %0\n\ts_mov_b32 s12, %0\n\ts_mov_b32 s13, %0\n\ts_mov_b32 s14, %0\n\ts_mov_b32 s15, %0\n\t"
        "s_mov_b32 s16, %0\n\ts_mov_b32 s17, %0\n\ts_mov_b32 s18, %0\n\ts_mov_b32 s19, %0\n\ts_mov_b32 s20, %0\n\ts_mov_b32 s21, %0\n\t"
        "s_mov_b32 s22, %0\n\ts_mov_b32 s23, %0\n\ts_mov_b32 s24, %0\n\ts_mov_b32 s25, %0\n\ts_mov_b32 s26, %0\n\ts_mov_b32 s27, %0\n\t"
        "s_mov_b32 s28, %0\n\ts_mov_b32 s29, %0\n\ts_mov_b32 s30, %0\n\ts_mov_b32 s31, %0\n\ts_mov_b32 s34, %0\n\ts_mov_b32 s35, %0\n\t"
        "s_mov_b32 s36, %0\n\ts_mov_b32 s37, %0\n\ts_mov_b32 s38, %0\n\ts_mov_b32 s39, %0\n\ts_mov_b32 s40, %0\n\ts_mov_b32 s41, %0\n\t"
        "s_mov_b32 s42, %0\n\ts_mov_b32 s43, %0\n\ts_mov_b32 s44, %0\n\ts_mov_b32 s45, %0\n\ts_mov_b32 s46, %0\n\ts_mov_b32 s47, %0\n\t"
        "s_mov_b32 s48, %0\n\ts_mov_b32 s49, %0\n\ts_mov_b32 s50, %0\n\ts_mov_b32 s51, %0\n\ts_mov_b32 s52, %0\n\ts_mov_b32 s53, %0\n\t"
        "s_mov_b32 s54, %0\n\ts_mov_b32 s55, %0\n\ts_mov_b32 s56, %0\n\ts_mov_b32 s57, %0\n\ts_mov_b32 s58, %0\n\ts_mov_b32 s59, %0\n\t"
        "s_mov_b32 s60, %0\n\ts_mov_b32 s61, %0\n\ts_mov_b32 s62, %0\n\ts_mov_b32 s63, %0\n\ts_mov_b32 s64, %0\n\ts_mov_b32 s65, %0\n\t"
        "s_mov_b32 s66, %0\n\ts_mov_b32 s67, %0\n\ts_mov_b32 s68, %0\n\ts_mov_b32 s69, %0\n\ts_mov_b32 s70, %0\n\ts_mov_b32 s71, %0\n\t"
        "s_mov_b32 s72, %0\n\ts_mov_b32 s73, %0\n\ts_mov_b32 s74, %0\n\ts_mov_b32 s75, %0\n\ts_mov_b32 s76, %0\n\ts_mov_b32 s77, %0\n\t"
        "s_mov_b32 s78, %0\n\ts_mov_b32 s79, %0\n\ts_mov_b32 s80, %0\n\ts_mov_b32 s81, %0\n\ts_mov_b32 s82, %0\n\ts_mov_b32 s83, %0\n\t"
        "s_mov_b32 s84, %0\n\ts_mov_b32 s85, %0\n\ts_mov_b32 s86, %0\n\ts_mov_b32 s87, %0\n\ts_mov_b32 s88, %0\n\ts_mov_b32 s89, %0\n\t"
        "s_mov_b32 s90, %0\n\ts_mov_b32 s91, %0\n\ts_mov_b32 s92, %0\n\ts_mov_b32 s93, %0\n\ts_mov_b32 s94, %0\n\ts_mov_b32 s95, %0\n\t"
        "s_mov_b32 s96, %0\n\ts_mov_b32 s97, %0\n\ts_mov_b32 s98, %0\n\ts_mov_b32 s99, %0\n\t"
        "s_nop 0"
        :: "s"(pat)
        :
          "v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24",
          "v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48",
          "v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72",
          "v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96",
          "v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120",
          "v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144",
          "v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159",
          "s4","s5","s6","s7","s8","s9","s10","s11","s12","s13","s14","s15","s16","s17","s18","s19","s20","s21","s22","s23","s24","s25","s26","s27",
          "s28","s29","s30","s31","s34","s35","s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53",
          "s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77",
          "s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99");
}
#endif

// VEC: float4 registers a thread holds while staging its channel (8: operand rows of 4..32 positions,
// nb * Lin <= 32; 16: 64 positions).  PREF: request the next K block's operand before the current MFMAs.
// LEAN (wide form only): the same arithmetic in half the registers and 40 KB of LDS, so that FOUR workgroups share a CU
//       (16 waves): a workgroup's life is a chain of latencies (operand fetch, GroupNorm exchange, weight stream, split-K
//       hand-off) around 1.7 us of matrix time, and with two workgroups per CU the CU idles through most of it (measured:
//       one workgroup per CU instead of two = 1.4-1.5x the loop time).  What it gives up: the weight ring is two k16 steps
//       per stage instead of four, the GroupNorm exchange arrays alias the (not yet written) slab — one more barrier —
//       and the epilogue operands are requested after the K loop instead of at kernel start.
//       64-position layers in the wide form (L = 64 latents: configurations C4 / C5): a workgroup stages the whole 64-position row of
//       ONE sample (GroupNorm needs it) and computes one 32-column half of it (A.nhalf = 2); with VEC = 16 the exchange arrays
//       alias the slab as in the lean form, which keeps two such workgroups on a CU.
// NT2  (lean form only; loops designed for >= 128 latents): TWO 32-column tiles per wave on K blocks of <= 128 channels.  The
//       operand of a K block is staged by two threads per channel (thread = (channel, half of the batch rows), 32 values each as
//       before), so a block of 112 channels x 64 columns costs what 224 x 32 cost — same LDS, same registers — but every
//       weight fragment now feeds six matrix instructions instead of three and the fixed phases of a workgroup's life (argument
//       fetch, split-K hand-off, epilogue) are paid once per 64 columns.  Needs its own weight packing (K blocks of <= 128
//       channels: conv2_plan_layout's second layout).
template <int VEC, bool PREF, bool WT = false, bool LEAN = false, bool LF = false, bool NT2 = false>
__global__ __launch_bounds__(256, LEAN ? SURFD_C2_LEAN_WAVES : (VEC == 16 ? (WT ? 2 : 1) : 2)) void conv2_kernel(Conv2Args A) {
    static_assert(!LEAN || (WT && VEC == 8 && !PREF), "lean form: wide decomposition, rows of <= 32 positions, no operand prefetch");
    static_assert(!NT2 || (LEAN && !LF), "two column tiles per wave: lean form only");
    constexpr int NCT = NT2 ? 2 : 1;      // column tiles a wave accumulates
    [[maybe_unused]] constexpr int EXS = NT2 ? 128 : 256;  // row stride of the GroupNorm exchange arrays = threads that own a channel
    // the register / LDS diet of the lean form, also applied to 64-position rows in the wide form (16 float4 of operand per
    // thread: without it the kernel spills 77 registers at two workgroups per CU)
    constexpr bool SLIM = LEAN || (WT && VEC == 16);
    // ONE accumulator per column tile (experiment SURFD_C2_LEAN_U=3: the 16 registers of the second one buy a third k16 step per
    // ring stage in the lean form)
    constexpr bool ONE_ACC = NT2 || (LEAN && SURFD_C2_LEAN_U == 3);
    constexpr int C2_U = (LEAN && !NT2) ? SURFD_C2_LEAN_U : SLIM ? 2 : 4;    // k16 steps per ring stage
    constexpr int C2_D = (VEC == 8 && SURFD_C2_EPI_LATE && !LEAN && !LF) ? (WT ? SURFD_C2_DEEP_D : SURFD_C2_LAT_D) : 2;
    [[maybe_unused]] constexpr bool ALIAS = SLIM;          // GroupNorm exchange arrays inside the (not yet written) slab
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    _Float16 *slab = reinterpret_cast<_Float16 *>(lds_raw);
#if !SURFD_C2_GNW
    float *ex_mean = reinterpret_cast<float *>(lds_raw + (ALIAS ? 0 : A.off_ex));     // [VEC][256]
    float *ex_m2 = ex_mean + VEC * 256;                                 // [VEC][256]  (NT2: [16 rows][128 channels])
    float *gstat = ex_m2 + VEC * 256;                                   // [nb * groups][2]
#endif
    float *red = reinterpret_cast<float *>(lds_raw + A.off_red);        // [3][1024] + flag (lean form: the flag alone)

#if defined(SURFD_C2_DBG_POISON) && SURFD_C2_DBG_POISON
    c2_poison_registers<!LEAN>();
    for (int e = threadIdx.x; e < A.C2_LDS_BYTES / 4; e += 256) reinterpret_cast<unsigned *>(lds_raw)[e] = SURFD_C2_DBG_POISON == 2 ? 0x7fc00000u : 0u;
    lds_bar();
#endif
    c2_kernarg_prefetch<(int)sizeof(Conv2Args)>();
    const int tid = threadIdx.x, lane = tid & 63;
    // provably wave-uniform: everything derived from it (k-part, column tile, iteration ranges, weight bases) stays in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SURFD_C2_STAMPS
    long long stamp_[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) stamp_[i] = 0;
#define C2_STAMP(i) do { stamp_[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define C2_STAMP_FIRST(i) do { if (ch == kz) C2_STAMP(i); } while (0)
#else
#define C2_STAMP(i) do { } while (0)
#define C2_STAMP_FIRST(i) do { } while (0)
#endif
    C2_STAMP(0);
#ifdef SURFD_C2_STAMPS
    const long long cyc0_ = (long long)__builtin_readcyclecounter();
#endif
    // ---- XCD-aware decode: group g = (tile, K slice); every batch chunk of a group on XCD g % 8 ----
    int tile, by, kz, rg;
    bool tile_ok = true;           // wide form: the last row group may have fewer than four tiles (wave-uniform)
    {
        const int bid = blockIdx.x;
        const int G = A.nrt * A.KS;
        int g;
        if (G < 8) {
            // fewer groups than XCDs (the 224-channel layers in the wide form): pinning a group to one XCD would leave most
            // of the chip idle — consecutive blocks (= consecutive XCDs) take consecutive groups, no padding blocks
            by = G == 1 ? bid : (int)__umulhi((unsigned)bid, A.magic_g);             // bid / G
            g = bid - by * G;
        } else {
            const int xcd = bid & 7, idx = bid >> 3;
            const int j = A.nby == 1 ? idx : (int)__umulhi((unsigned)idx, A.magic_nby);   // idx / nby (host-made reciprocal, idx < 2^16; 2^32 / 1 does not fit)
            by = idx - j * A.nby;
            g = xcd + 8 * j;
            if (g >= G) return;
        }
        rg = A.KS == 1 ? g : (int)__umulhi((unsigned)g, A.magic_ks);                // g / KS
        kz = g - rg * A.KS;
        tile = rg;
        if constexpr (WT) {
            tile = rg * 4 + wave;
            tile_ok = tile < A.ntiles;
            tile = min(tile, A.ntiles - 1);       // an idle wave streams the last tile's weights (valid addresses, same schedule) and discards the result
        }
    }
    // 64 output positions in the wide form: `by` counts (sample, 32-column half) pairs; m_off = first column of this half
    const bool halves = WT && A.nhalf == 2;
    const int m_off = halves ? (by & 1) * 32 : 0;
    const int b0 = (halves ? by >> 1 : by) * A.bchunk;
    const int nb = min(A.bchunk, A.B - b0);
    const int M = halves ? 32 : nb * A.Lout;
    // latency form: 1 or 2 column tiles (host guarantees M <= 64), 4 or 2 waves share a column tile as k-parts;
    // wide form: one column tile (M <= 32), every wave runs the whole K slice for its own row tile
    const int nct = WT ? 1 : (M + 31) >> 5;
    const int log2kp = WT ? 0 : (nct == 1) ? 2 : 1;
    const int KP = 1 << log2kp;
    const int ct = (WT || nct == 1) ? 0 : (wave & 1);
    const int kpart = WT ? 0 : (nct == 1) ? wave : (wave >> 1);
    // staging ownership: thread <-> channel (NT2: thread <-> (channel, half of the batch rows)); the channel of a thread is a
    // function of the segment (chan_of below): wave wc owns the group slots [wc * slots, (wc + 1) * slots) of the K block
    const int zthr = NT2 ? (tid & 127) : tid;          // identity index, used where any thread will do (zero fill of padded channels)
    const int wc = NT2 ? (wave & 1) : wave;
    const int rhalf = NT2 ? (wave >> 1) : 0;
    auto chan_of = [&](int s, bool &ok) -> int {
#if SURFD_C2_GNW
        const int lp = A.seg[s].log2P, gsm = A.seg[s].gsm;
        const int pos = lane & ((1 << lp) - 1);
        const int c = ((wc << (6 - lp)) + (lane >> lp)) * gsm + pos;
        ok = pos < gsm && c < A.seg[s].blk;
        return c;
#else
        ok = zthr < A.seg[s].blk;          // thread = channel
        return zthr;
#endif
    };
    int colb[NCT], coll[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
        int m = (NT2 ? t : ct) * 32 + (lane & 31);
        if (m >= M) m = 0;
        m += m_off;
        colb[t] = m >> A.log2Lout;
        coll[t] = m & (A.Lout - 1);
    }
    const int nblk0 = A.seg[0].nblk;
    const int nch = nblk0 + (A.nseg > 1 ? A.seg[1].nblk : 0);
    // the low fp16 plane of the slab sits PLANE halfs behind the high one: a compile-time LDS offset
    constexpr int PLANE = NT2 ? C2_PLANE_NT2 : LEAN ? C2_PLANE_LEAN : (VEC == 16 ? 18432 : 15360);
    const int cs = A.cs;
    const float inv_sc = A.inv_sc;

    // two accumulators: the two small cross terms (xl*wh, xh*wl) share one, the main term has its own; a third
    // would push the kernel over 256 VGPRs (2 workgroups per CU) and make the compiler spill a just-loaded value
    // (NT2: ONE accumulator per column tile — the two tiles' products interleave, so consecutive matrix instructions still never
    //  wait on each other — and no second set of 32 registers)
    f32x16 acc_hh[NCT], acc_sm[1];
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_hh[t][r] = 0.f;
    if constexpr (!ONE_ACC)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_sm[0][r] = 0.f;

    // ---- weight stream state of one K block for this wave ------------------------------------------
    // Every wave of the workgroup runs the SAME static schedule (ngroups is wave-uniform; a wave whose k-part is
    // shorter re-loads its last fragment and skips the MFMAs): all weight loads are unconditional, so the
    // compiler's s_waitcnt bookkeeping stays exact and a ring stage is awaited with vmcnt(16), not vmcnt(0).
    struct WS { const _Float16 *base; int it_beg, it_end, ngroups, nk; };
    auto make_ws = [&](int ch) -> WS {
        const int s = ch >= nblk0 ? 1 : 0;
        const int bi = ch - (s ? nblk0 : 0);
        const int nk = A.seg[s].blkp >> 4;
        const int iters = A.seg[s].taps * nk;
        const int per = (iters + KP - 1) >> log2kp;
        WS w;
        w.nk = nk;
        w.it_beg = min(kpart * per, iters);
        w.it_end = min(iters, w.it_beg + per);
        w.ngroups = (per + C2_U - 1) / C2_U;
        w.base = A.whf + ((size_t)tile * A.KS16 + A.seg[s].k16_off + (size_t)bi * iters) * 1024 + lane * 8;
        return w;
    };
    f16x8 ring[C2_D][C2_U][2];
    // fragments [it0, it0 + U) of K block w, clamped into the block (valid addresses even for an empty k-part)
    auto load_group = [&](f16x8 (&dst)[C2_U][2], const _Float16 *base, int it0, int it_last) {
#pragma unroll
        for (int u = 0; u < C2_U; ++u) {
#if defined(SURFD_C2_ABLATE) && SURFD_C2_ABLATE == 2      // developer aid: every fragment from one address (no weight stream)
            const int it = 0 * (it0 + it_last);
#else
            const int it = max(min(it0 + u, it_last), 0);
#endif
            const gf16x8 *p = (const gf16x8 *)(base + (size_t)it * 1024);
            dst[u][0] = p[0];
            dst[u][1] = p[64];          // low plane: +512 halfs
        }
    };

    // ---- raw operand of one K block: thread <-> channel, VEC float4 covering (batch row, position) ----
    // All global loads of this kernel are UNCONDITIONAL (addresses clamped into the tensor, validity applied to
    // the value later): a load under a lane- or wave-dependent branch makes the compiler's vmcnt bookkeeping
    // conservative, and one resulting s_waitcnt vmcnt(0) in front of the MFMAs serialises the whole weight ring.
    auto issue_operand = [&](int ch, f32x4 (&v)[VEC], float &ga, float &be) {
        const int s = ch >= nblk0 ? 1 : 0;
        const int bi = ch - (s ? nblk0 : 0);
        const int lv = A.seg[s].log2Lin - 2;          // log2(float4 per row)
        const int Lin = A.seg[s].Lin;
        bool cok_;
        const int cg = min(bi * A.seg[s].blk + min(chan_of(s, cok_), A.seg[s].blk - 1), A.seg[s].C - 1);
        const float *src = A.seg[s].x + (long)cg * Lin;
        const long bstride = A.seg[s].bstride;
        const int rowbase = rhalf * ((VEC * 4) >> A.seg[s].log2Lin);      // first batch row this thread stages (NT2: second half of the rows)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int i = min(rowbase + (j >> lv), nb - 1), jj = j & ((1 << lv) - 1);
            v[j] = *reinterpret_cast<const f32x4 *>(src + (b0 + i) * bstride + 4 * jj);
        }
        ga = A.seg[s].gamma[cg]; be = A.seg[s].beta[cg];      // segments without GroupNorm point these at the bias vector
    };

    int ch = kz;
    WS cur = make_ws(ch);
    f32x4 v[VEC];
    float ga, be;
    issue_operand(ch, v, ga, be);
#pragma unroll
    for (int d = 0; d < C2_D; ++d) load_group(ring[d], cur.base, cur.it_beg + d * C2_U, cur.it_end - 1);
    unsigned pf_sink[C2_PFN_N];
    if constexpr (SURFD_C2_PFN_FORMS || LEAN) c2_prefetch_next(A, tid, pf_sink);     // the next convolution's weights, towards this XCD's L2
    else {
#pragma unroll
        for (int k = 0; k < C2_PFN_N; ++k) pf_sink[k] = 0u;
    }

    // ---- epilogue operands (bias + per-(step, sample) embedding + residual), requested now, used at the end ----
    // kept as three separate register sets and only combined in the epilogue: combining them here would put a
    // wait for the (HBM-resident) embedding rows in front of the first K block
    f32x4 pre_b[4], pre_e[NCT][4];
    float pre_r[NCT][16];
    const float *embp = A.emb;                        // never null: the host points unused operands at the bias vector
    // scalar load, requested here, first used by request_epilogue.  Through the constant address space: behind the prefetch
    // statement above the compiler no longer proves the counter unclobbered and would fetch it with a VECTOR load — whose
    // s_waitcnt vmcnt(0) in front of the address arithmetic waits for every weight fragment requested so far
    if (A.step_ptr) embp += (long)(*reinterpret_cast<const __attribute__((address_space(4))) int *>((unsigned long)A.step_ptr)) * A.emb_step_stride;
    // head of a fused loop: the loop record, the iteration and its coefficient row are requested now — behind the operand and
    // weight requests above — and first used in the epilogue
    LoopFuse lfv;
    int lfk = 0;
    float lfrow[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (LF) {
        lfv = *A.lf;
        lfk = *lfv.step;
#pragma unroll
        for (int q = 0; q < 5; ++q) lfrow[q] = lfv.tab[(long)lfk * 8 + q];
    }
    auto request_epilogue = [&]() {
        const int cmax4 = ((A.Cout + 3) & ~3) - 4;     // last aligned float4 of the (4-padded) per-channel vectors
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            const int m = min((NT2 ? t : ct) * 32 + (lane & 31), M - 1) + m_off;
            const int b = b0 + (m >> A.log2Lout), l = m & (A.Lout - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cob = min(tile * 32 + 8 * q + 4 * (lane >> 5), cmax4);      // rows frag_row(4q .. 4q+3, lane)
                if (t == 0) pre_b[q] = *reinterpret_cast<const f32x4 *>(A.bias + cob);
                pre_e[t][q] = *reinterpret_cast<const f32x4 *>(embp + b * A.emb_bstride + cob);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int co = min(cob + k, A.Cout - 1);
                    pre_r[t][4 * q + k] = A.res[b * A.res_bstride + (long)co * A.res_cstride + l * A.res_lstride];
                }
            }
        }
    };
    // Latency form, distributed tail (SURFD_C2_DIST, round 6): the k-parts of a column tile leave their partial tiles in LDS and wave
    // kpart then owns the accumulator registers [4 q, 4 q + 4) of the quarter-blocks q in [kpart * QN, (kpart + 1) * QN), QN = 4 / KP
    // (rows 8 q + 4 (lane >> 5) + 0..3 of the tile): it sums them over the k-parts in k-part order (kp = 0 first: the same additions
    // in the same order as the one-owner form, same bits), publishes / gathers exactly those rows in a split K, and finishes them —
    // the reduction, the hand-off reads and the epilogue of a tile are four waves' work instead of one's, and a lane requests 6 or
    // 12 epilogue operands instead of 24.
#if defined(SURFD_C2_PROBE)
    constexpr bool DIST = false;          // the probe build reads the one-owner form's registers
#else
    constexpr bool DIST = !WT && SURFD_C2_DIST;
#endif
    const int QN = 4 >> log2kp;           // quarter-blocks per wave in the distributed tail (1 or 2)
    f32x4 dpre_b[2], dpre_e[2];
    float dpre_r[2][4];
    auto request_epilogue_dist = [&]() {
        const int cmax4 = ((A.Cout + 3) & ~3) - 4;
        const int m = min(ct * 32 + (lane & 31), M - 1) + m_off;
        const int b = b0 + (m >> A.log2Lout), l = m & (A.Lout - 1);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int q = kpart * QN + min(qq, QN - 1);
            const int cob = min(tile * 32 + 8 * q + 4 * (lane >> 5), cmax4);
            dpre_b[qq] = *reinterpret_cast<const f32x4 *>(A.bias + cob);
            dpre_e[qq] = *reinterpret_cast<const f32x4 *>(embp + b * A.emb_bstride + cob);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = min(cob + k, A.Cout - 1);
                dpre_r[qq][k] = A.res[b * A.res_bstride + (long)co * A.res_cstride + l * A.res_lstride];
            }
        }
    };
    // SURFD_C2_EPI_LATE (default): every form requests the epilogue operands after the K loop.  Requested here they put 24 more
    // vector-memory instructions per lane in front of the wait for the operand (the wave is busy issuing for ~2.6 us while the
    // operand is back after ~1.3); after the K loop their round trip hides behind the k-part reduction and the split-K hand-off
    constexpr bool EPI_LATE = SLIM || SURFD_C2_EPI_LATE;
    if constexpr (!EPI_LATE && !DIST) request_epilogue();
    bool saturated = false;
    C2_STAMP(1);
#ifdef SURFD_C2_PROBE
    unsigned long long probe_w = 0ull;
#endif

    while (true) {
        // =========================== stage K block `ch` into the slab ===============================
        {
            const int s = ch >= nblk0 ? 1 : 0;
            const int bi = ch - (s ? nblk0 : 0);
            const int lv = A.seg[s].log2Lin - 2, vpr = 1 << lv;
            const int Lin = A.seg[s].Lin;
            const int blk = A.seg[s].blk, blkp = A.seg[s].blkp;
            bool cok;
            const int c = chan_of(s, cok);
            const int rowbase = rhalf * ((VEC * 4) >> A.seg[s].log2Lin);
            const int pad = A.seg[s].taps == 3 ? 1 : 0;
            const int ups = A.seg[s].ups, act = A.seg[s].act;
#ifdef SURFD_C2_PROBE
            {
                unsigned long long h = 0ull;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) h += (unsigned long long)__float_as_uint(v[j][k]) * (unsigned long long)(2 * ((ch * 256 + tid) * 64 + j * 4 + k) + 1);
                h += (unsigned long long)__float_as_uint(ga) * 7ull + (unsigned long long)__float_as_uint(be) * 11ull;
                C2_PROBE_ADD(0, h);
            }
#endif
            if (A.seg[s].gn) {
                const int gs = A.seg[s].gs;
                // ---- GroupNorm statistics, two-pass, in registers first: per (batch row, channel) mean and M2 over the
                //      row's Lin positions.  A row is 2^lv consecutive float4 of this thread; the sums go up a binary
                //      tree whose levels are enabled by the (wave-uniform) lv, every member of a block ending up with the
                //      block's total — no register is ever indexed dynamically.
                constexpr int LOG2VEC = VEC == 16 ? 4 : 3;
                float rs[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) rs[j] = (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
                auto tree = [&](float (&t)[VEC]) {
#pragma unroll
                    for (int lev = 0; lev < LOG2VEC; ++lev)
                        if (lev < lv) {
#pragma unroll
                            for (int j = 0; j < VEC; j += 2 << lev) {
                                const float tot = t[j] + t[j + (1 << lev)];
#pragma unroll
                                for (int m = 0; m < (2 << lev); ++m) t[j + m] = tot;
                            }
                        }
                };
                tree(rs);
                const float inv_len = 1.f / (float)Lin;
                float rm2[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    rs[j] *= inv_len;                                  // row mean, seen from every float4 of the row
                    float m2 = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float d = v[j][q] - rs[j]; m2 += d * d; }
                    rm2[j] = m2;
                }
                tree(rm2);
#if SURFD_C2_GNW
                // ---- per (batch row, group), inside the wave: equal-size rows combine exactly (Chan et al.):
                //      mean = avg(row means),  M2 = sum(row M2) + Lin * sum((row mean - mean)^2).  The butterfly runs once per
                //      row (at the row's first float4 — a wave-uniform choice); the results then go down the same binary tree the
                //      row sums came up, so no register is indexed dynamically.
                const int lp = A.seg[s].log2P;
                const float inv_gs = 1.f / (float)gs, inv_cnt = 1.f / (float)(gs * Lin), flin = (float)Lin;
                float gmr[VEC], gscr[VEC];
                // S = float4 per row (compile time per branch of the wave-uniform dispatch below): the VEC / S rows of this thread
                // go through the two butterflies together — rows past the batch chunk (clamped re-reads of the last row) included,
                // their results are never used — and every float4 of a row receives the row's result by static index
                auto group_stats = [&](auto stride) {
                    constexpr int S = decltype(stride)::value, NR = VEC / S;
                    float a[NR], gm[NR];
#pragma unroll
                    for (int i = 0; i < NR; ++i) a[i] = cok ? rs[i * S] : 0.f;
                    c2_slot_sum<NR>(a, lp);
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        gm[i] = a[i] * inv_gs;
                        const float d = rs[i * S] - gm[i];
                        a[i] = cok ? rm2[i * S] + flin * (d * d) : 0.f;
                    }
                    c2_slot_sum<NR>(a, lp);
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const float sc = ga * (1.f / sqrtf(a[i] * inv_cnt + 1e-5f));
#pragma unroll
                        for (int m = 0; m < S; ++m) { gmr[i * S + m] = gm[i]; gscr[i * S + m] = sc; }
                    }
                };
                if (lv == 0) group_stats(std::integral_constant<int, 1>());
                else if (lv == 1) group_stats(std::integral_constant<int, 2>());
                else if (lv == 2) group_stats(std::integral_constant<int, 4>());
                else if (lv == 3 || VEC == 8) group_stats(std::integral_constant<int, 8>());
                else group_stats(std::integral_constant<int, VEC>());
                C2_STAMP_FIRST(2);
                C2_STAMP_FIRST(3);
#else
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if ((j & (vpr - 1)) == 0 && rowbase + (j >> lv) < nb) {     // first float4 of a live row (uniform per wave)
                        ex_mean[(rowbase + (j >> lv)) * EXS + c] = rs[j];
                        ex_m2[(rowbase + (j >> lv)) * EXS + c] = rm2[j];
                    }
                lds_bar();
                C2_STAMP_FIRST(2);
                // ---- per (batch row, group): equal-size rows combine exactly (Chan et al.):
                //      mean = avg(row means),  M2 = sum(row M2) + Lin * sum((row mean - mean)^2); 8 lanes per group
                const int ng = blk / gs, nq = nb * ng;
                const float inv_gs = 1.f / (float)gs, inv_cnt = 1.f / (float)(gs * Lin), flin = (float)Lin;
                for (int q0 = 0; q0 < nq; q0 += 32) {
                    const int q = q0 + (tid >> 3), lt = tid & 7;
                    const bool qok = q < nq;
                    const int i = qok ? q / ng : 0;
                    const int g = qok ? q - i * ng : 0;
                    const float *pm = ex_mean + i * EXS + g * gs;
                    const float *p2 = ex_m2 + i * EXS + g * gs;
                    float sm = 0.f;
                    if (qok)
                        for (int k = lt; k < gs; k += 8) sm += pm[k];
                    c2_gnpad(sm); sm += __shfl_xor(sm, 4); c2_gnpad(sm); sm += __shfl_xor(sm, 2); c2_gnpad(sm); sm += __shfl_xor(sm, 1); c2_gnpad(sm);
                    const float gm = sm * inv_gs;
                    float m2 = 0.f;
                    if (qok)
                        for (int k = lt; k < gs; k += 8) { const float d = pm[k] - gm; m2 += p2[k] + flin * (d * d); }
                    c2_gnpad(m2); m2 += __shfl_xor(m2, 4); c2_gnpad(m2); m2 += __shfl_xor(m2, 2); c2_gnpad(m2); m2 += __shfl_xor(m2, 1); c2_gnpad(m2);
                    if (qok && lt == 0) { gstat[2 * q] = gm; gstat[2 * q + 1] = 1.f / sqrtf(m2 * inv_cnt + 1e-5f); }
                }
                lds_bar();
                C2_STAMP_FIRST(3);
                // group mean / scale of every row this thread holds, in registers: in the lean form the exchange arrays
                // (gstat included) alias the slab, so they must be dead — one more barrier — before the slab is written
                float gmr[VEC], gscr[VEC];
                {
                    const int gq = min(c, blk - 1) / gs;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const int i = rowbase + (j >> lv);
                        const int q = (i < nb) ? i * ng + gq : 0;
                        gmr[j] = gstat[2 * q]; gscr[j] = ga * gstat[2 * q + 1];
                    }
                }
#endif
#ifdef SURFD_C2_PROBE
                {
                    unsigned long long h = 0ull;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const bool live = rowbase + (j >> lv) < nb;          // rows past the batch chunk hold values nobody uses
                        h += live ? ((unsigned long long)__float_as_uint(gmr[j]) + 3ull * __float_as_uint(gscr[j])) * (unsigned long long)(2 * ((ch * 256 + tid) * 16 + j) + 1) : 0ull;
                    }
                    C2_PROBE_ADD(1, cok ? h : 0ull);
                }
#endif
#if !SURFD_C2_GNW
                if constexpr (ALIAS) lds_bar();
#endif
                if (cok && !SLIM) {
                    // two values per instruction (v_pk_add / v_pk_mul: the same IEEE operations per element, same bits) and the
                    // wave-uniform `act` decided once, not per value.  Not in the register-lean forms (the lean kernel's 168 and the
                    // 64-position wide kernel's 256 registers have no room for the independent temporaries: 11 / 55 spills)
                    const f32x2 be2 = {be, be};
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const f32x2 gm2 = {gmr[j], gmr[j]}, gs2 = {gscr[j], gscr[j]};
#pragma unroll
                        for (int k = 0; k < 4; k += 2) {
                            const f32x2 x = {v[j][k], v[j][k + 1]};
                            const f32x2 w = (x - gm2) * gs2 + be2;
                            v[j][k] = w[0]; v[j][k + 1] = w[1];
                        }
                    }
#if !(defined(SURFD_C2_ABLATE) && SURFD_C2_ABLATE == 3)  // developer aid 3: no SiLU
                    if (act) {
#pragma unroll
                        for (int j = 0; j < VEC; ++j) silu2_x4(v[j]);
                    }
#endif
                } else if (cok) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const float gm = gmr[j], gsc = gscr[j];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float w = (v[j][k] - gm) * gsc + be;
#if !(defined(SURFD_C2_ABLATE) && SURFD_C2_ABLATE == 3)
                            if (act) w = silu2(w);
#endif
                            v[j][k] = w;
                        }
                    }
                }
            } else if (act) {
                if constexpr (!SLIM) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) silu2_x4(v[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[j][k] = silu2(v[j][k]);
                }
            }
            // ---- split and write the slab [batch row][position][channel]; zero halo positions and padded channels.
            //      Two positions at a time: one packed conversion per pair, the halves stored with ds_write_b16 /
            //      ds_write_b16_d16_hi; the low plane sits at a compile-time offset (an LDS immediate, no address math).
            const int rstep = ups ? 2 : 1;
            const int Lcov = ups ? 2 * Lin : Lin;
#ifdef SURFD_C2_PROBE
            {
                unsigned long long h = 0ull;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const bool live = rowbase + (j >> lv) < nb;
#pragma unroll
                    for (int k = 0; k < 4; ++k) h += live ? (unsigned long long)__float_as_uint(v[j][k]) * (unsigned long long)(2 * ((ch * 256 + tid) * 64 + j * 4 + k) + 1) : 0ull;
                }
                C2_PROBE_ADD(2, cok ? h : 0ull);
            }
#endif
            if (cok) {
                float amax = 0.f;
                // the (wave-uniform) upsampling decided once, not per pair of positions: a branch per pair is sixteen basic blocks the
                // scheduler cannot interleave across (every form: no register more in the lean kernels, wide loops 0.6-0.9 % faster)
                auto split_rows = [&](auto upsc) {
                    constexpr bool UPS = decltype(upsc)::value;
                    constexpr int RS = UPS ? 2 : 1;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const int i = rowbase + (j >> lv), jj = j & (vpr - 1);
                        if (i < nb) {
                            _Float16 *row = slab + (i * A.Lsl + pad + RS * 4 * jj) * cs + c;
#pragma unroll
                            for (int k = 0; k < 4; k += 2) {
                                amax = fmaxf(amax, fmaxf(fabsf(v[j][k]), fabsf(v[j][k + 1])));
                                const f32x2 w = {__builtin_amdgcn_fmed3f(v[j][k], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[j][k + 1], -65504.f, 65504.f)};
                                const f16x2 h = __builtin_convertvector(w, f16x2);
                                const f32x2 hf = __builtin_convertvector(h, f32x2);
                                const f16x2 lo = __builtin_convertvector(w - hf, f16x2);
                                _Float16 *d0 = row + (k * RS) * cs, *d1 = d0 + RS * cs;
                                d0[0] = h[0]; d0[PLANE] = lo[0];
                                d1[0] = h[1]; d1[PLANE] = lo[1];
                                if constexpr (UPS) { d0[cs] = h[0]; d0[cs + PLANE] = lo[0]; d1[cs] = h[1]; d1[cs + PLANE] = lo[1]; }
                            }
                        }
                    }
                };
                if (ups) split_rows(std::true_type{}); else split_rows(std::false_type{});
                saturated |= amax > 65504.f;
            }
            if (zthr >= blk && zthr < blkp) {      // channels that only exist as padding of the K block: zeros (any thread will do)
                for (int i = 0; i < nb; ++i)
                    for (int p = 0; p < A.Lsl; ++p) { slab[(i * A.Lsl + p) * cs + zthr] = (_Float16)0.f; slab[(i * A.Lsl + p) * cs + zthr + PLANE] = (_Float16)0.f; }
            }
            if (cok)                               // halo positions of the channel this thread owns
                for (int i = 0; i < nb; ++i) {
                    _Float16 *row0 = slab + (i * A.Lsl) * cs + c;
                    for (int p = 0; p < pad; ++p) { row0[p * cs] = (_Float16)0.f; row0[p * cs + PLANE] = (_Float16)0.f; }
                    for (int p = pad + Lcov; p < A.Lsl; ++p) { row0[p * cs] = (_Float16)0.f; row0[p * cs + PLANE] = (_Float16)0.f; }
                }
            lds_bar();
            C2_STAMP_FIRST(4);
#ifdef SURFD_C2_PROBE
            {   // the slab as the matrix instructions will read it: every (batch row, position, channel < blkp) of both planes
                unsigned long long h = 0ull;
                const int n = nb * A.Lsl * blkp;
                for (int e = tid; e < n; e += 256) {
                    const int cch = e % blkp, rp = e / blkp;
                    const unsigned short hi = __builtin_bit_cast(unsigned short, slab[rp * cs + cch]), lo = __builtin_bit_cast(unsigned short, slab[rp * cs + cch + PLANE]);
                    h += ((unsigned long long)hi + 65537ull * lo) * (unsigned long long)(2 * (ch * 65536 + e) + 1);
                }
                C2_PROBE_ADD(3, h);
                lds_bar();
            }
#endif
        }
        // =========================== MFMAs of this K block ============================================
        const int chn = ch + A.KS;
        const bool has_next = chn < nch;
        const WS nxt = has_next ? make_ws(chn) : cur;
        f32x4 vn[PREF ? VEC : 1];
        float gan = 0.f, ben = 0.f;
        if constexpr (PREF) issue_operand(has_next ? chn : ch, vn, gan, ben);      // unconditional (re-reads this block at the end)
        {
            const int s = ch >= nblk0 ? 1 : 0;
            int lbase[NCT];
#pragma unroll
            for (int t = 0; t < NCT; ++t) lbase[t] = (colb[t] * A.Lsl + coll[t] * A.seg[s].stride) * cs + 8 * (lane >> 5);
            const int nk = cur.nk;
#if SURFD_C2_BPIPE
            // B operand (the slab's two fp16 planes) of the NEXT k16 step, requested before the matrix instructions of the current
            // one (round 6, profiles/r06_conv2_instability.md): (i) the LDS round trip of a step hides behind the previous step's
            // three MFMAs instead of standing in front of them twice per step; (ii) a load never writes the source registers of a
            // matrix instruction issued just before it — the registers it targets held the operand of the step BEFORE the
            // current one.  Steps of a wave run it_beg, it_beg + 1, ... (the skipped ones are all at the tail), so "next" is
            // it + 1; the last step re-reads itself.
            f16x8 nbh, nbl;
            auto bload = [&](int it, f16x8 &h, f16x8 &l) {
                const int tap = (it >= nk) + (it >= 2 * nk);
                const _Float16 *bp = slab + lbase[0] + tap * cs + (it - tap * nk) * 16;
                h = *reinterpret_cast<const f16x8 *>(bp);
                l = *reinterpret_cast<const f16x8 *>(bp + PLANE);
            };
            if constexpr (!NT2) bload(min(cur.it_beg, max(cur.it_end - 1, 0)), nbh, nbl);
#endif
            auto compute = [&](const f16x8 (&a)[C2_U][2], int g) {
#pragma unroll
                for (int u = 0; u < C2_U; ++u) {
                    const int it = cur.it_beg + g * C2_U + u;
                    if (it < cur.it_end) {          // wave-uniform (scalar branch); only LDS reads and MFMAs inside: vmcnt bookkeeping unaffected
#ifdef SURFD_C2_PROBE
                        {
                            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                            const u32x4 wa = __builtin_bit_cast(u32x4, a[u][0]), wb = __builtin_bit_cast(u32x4, a[u][1]);
                            unsigned long long hw = 0ull;
#pragma unroll
                            for (int q = 0; q < 4; ++q) hw += ((unsigned long long)wa[q] + 5ull * wb[q]) * (unsigned long long)(2 * (((ch * 64 + it) * 256 + tid) * 4 + q) + 1);
                            probe_w += hw;
                        }
#endif
                        const int tap = (it >= nk) + (it >= 2 * nk);
                        const int koff = tap * cs + (it - tap * nk) * 16;
                        if constexpr (NT2) {
                            const _Float16 *bp0 = slab + lbase[0] + koff, *bp1 = slab + lbase[NCT - 1] + koff;
                            const f16x8 bh0 = *reinterpret_cast<const f16x8 *>(bp0), bh1 = *reinterpret_cast<const f16x8 *>(bp1);
                            const f16x8 bl0 = *reinterpret_cast<const f16x8 *>(bp0 + PLANE), bl1 = *reinterpret_cast<const f16x8 *>(bp1 + PLANE);
                            // small terms first, the two column tiles alternating
                            acc_hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][1], bh0, acc_hh[0], 0, 0, 0);
                            acc_hh[NCT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][1], bh1, acc_hh[NCT - 1], 0, 0, 0);
                            acc_hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][0], bl0, acc_hh[0], 0, 0, 0);
                            acc_hh[NCT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][0], bl1, acc_hh[NCT - 1], 0, 0, 0);
                            acc_hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][0], bh0, acc_hh[0], 0, 0, 0);
                            acc_hh[NCT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][0], bh1, acc_hh[NCT - 1], 0, 0, 0);
                        } else {
#if SURFD_C2_BPIPE
                        (void)koff;
                        const f16x8 bh = nbh, bl = nbl;
                        bload(min(it + 1, cur.it_end - 1), nbh, nbl);
#else
                        const _Float16 *bp = slab + lbase[0] + koff;
                        const f16x8 bh = *reinterpret_cast<const f16x8 *>(bp);
                        const f16x8 bl = *reinterpret_cast<const f16x8 *>(bp + PLANE);
#endif
#if defined(SURFD_C2_ABLATE) && SURFD_C2_ABLATE == 1      // developer aid: no matrix work (operands still fetched)
                        acc_sm[0][0] += (float)a[u][1][0] + (float)bh[0]; acc_hh[0][0] += (float)a[u][0][0] + (float)bl[0];
#else
                        if constexpr (ONE_ACC) {
                            acc_hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][1], bh, acc_hh[0], 0, 0, 0);
                            acc_hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][0], bl, acc_hh[0], 0, 0, 0);
                            acc_hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][0], bh, acc_hh[0], 0, 0, 0);
                        } else {
                        acc_sm[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][1], bh, acc_sm[0], 0, 0, 0);
                        acc_hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][0], bh, acc_hh[0], 0, 0, 0);
                        acc_sm[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][0], bl, acc_sm[0], 0, 0, 0);
                        }
#endif
#if SURFD_C2_BPIPE == 1
                        // this step's operand registers stay allocated until here: the loads of the next step's operand (issued
                        // above, ordered before this statement by its memory clobber) cannot be given the registers the matrix
                        // instructions of this step are still reading.  (SURFD_C2_BPIPE=2, the experiment's control: the same
                        // request order WITHOUT this statement — the compiler then sinks the loads behind the last use of bl and
                        // gives them bl's registers again.)
                        asm volatile("" :: "v"(bh), "v"(bl) : "memory");
#endif
                        }
                    }
                }
            };
            const int P = ((cur.ngroups + C2_D - 1) / C2_D) * C2_D;       // ngroups >= 1
            // do-while: the body runs at least once (P >= D), which lets the compiler prove that >= 24 weight loads
            // are younger than the prefetched operand when the next block is staged (vmcnt(33), not vmcnt(9))
            int g0 = 0;
#pragma unroll 1
            do {
#pragma unroll
                for (int d = 0; d < C2_D; ++d) {
                    const int g = g0 + d;
                    compute(ring[d], g);
                    // refill this stage: group g + D of this block, or (last pass) group d of the next block, or a
                    // harmless re-load — always issued
                    const int vg = g + C2_D;
                    const bool into_next = has_next && vg >= P;
                    const _Float16 *rb = into_next ? nxt.base : cur.base;
                    const int it0 = into_next ? nxt.it_beg + (vg - P) * C2_U : cur.it_beg + vg * C2_U;
                    const int itl = into_next ? nxt.it_end - 1 : cur.it_end - 1;
                    load_group(ring[d], rb, it0, itl);
                }
                g0 += C2_D;
            } while (g0 < P);
        }
        C2_STAMP_FIRST(5);
        if (!has_next) break;
        lds_bar();                   // every wave is done reading the slab
        ch = chn;
        cur = nxt;
        if constexpr (PREF) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = vn[j];
            ga = gan; be = ben;
        } else {
            issue_operand(ch, v, ga, be);
        }
    }
    if constexpr (DIST) request_epilogue_dist();
    else if constexpr (EPI_LATE) request_epilogue();      // round trip hidden behind the split-K hand-off (or the other workgroups of the CU)
    if (saturated) atomicAdd(A.sat, 1u);
    C2_STAMP(6);

    if constexpr (DIST) {
        // ---- distributed tail of the latency form (see SURFD_C2_DIST above) ----
        lds_bar();
        {
            float *dst = red + (kpart * nct + ct) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r * 64 + lane] = ONE_ACC ? acc_hh[0][r] : acc_sm[0][r] + acc_hh[0][r];
        }
        lds_bar();
        f32x4 val[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int q = kpart * QN + min(qq, QN - 1);
            const float *srcp = red + ct * 1024 + (4 * q) * 64 + lane;
            f32x4 t = {srcp[0], srcp[64], srcp[128], srcp[192]};
            for (int kp = 1; kp < KP; ++kp) {
                const float *sp = srcp + kp * nct * 1024;
                t[0] += sp[0]; t[1] += sp[64]; t[2] += sp[128]; t[3] += sp[192];
            }
            val[qq] = t;
        }
        C2_STAMP(7);
        if (A.KS > 1) {
            const size_t slot = (size_t)by * A.nrt + rg;
            float *mine = A.part + (((size_t)kz * A.nby + by) * A.ntiles + tile) * A.part_stride;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
                if (qq < QN) {
                    float *dst = mine + ((size_t)(ct * 4 + kpart * QN + qq) * 64 + lane) * 4;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(val[qq]) : "memory");
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int *flag = reinterpret_cast<int *>(red + 4 * 1024);
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
            // the partial tiles of DZB slices in flight together, summed in slice order (as SURFD_C2_ZB in the one-owner form)
            constexpr int DZB = 8;
            val[0] = f32x4{0.f, 0.f, 0.f, 0.f}; val[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            const size_t zstride = (size_t)A.nby * A.ntiles * A.part_stride;
            const float *src0 = A.part + ((size_t)by * A.ntiles + tile) * A.part_stride + ((size_t)(ct * 4 + kpart * QN) * 64 + lane) * 4;
            for (int z0 = 0; z0 < A.KS; z0 += DZB) {
                f32x4 pv[DZB][2];
#pragma unroll
                for (int zz = 0; zz < DZB; ++zz) {
                    const int z = min(z0 + zz, A.KS - 1);
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) pv[zz][qq] = c2_load_partial(src0 + (size_t)z * zstride + (size_t)min(qq, QN - 1) * 256);
                }
#pragma unroll
                for (int zz = 0; zz < DZB; ++zz)
                    if (z0 + zz < A.KS) {
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[qq][e] += pv[zz][qq][e];
                    }
            }
        }
        C2_STAMP(8);
        {
            const int ml = ct * 32 + (lane & 31), m = ml + m_off;
            const bool mok = ml < M;
            const int b = b0 + (m >> A.log2Lout), l = m & (A.Lout - 1);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = tile * 32 + e + 8 * (kpart * QN + qq) + 4 * (lane >> 5);
                    if (qq < QN && mok && co < A.Cout) {
                        const float o = val[qq][e] * inv_sc + ((dpre_b[qq][e] + (A.has_emb ? dpre_e[qq][e] : 0.f)) + (A.has_res ? dpre_r[qq][e] : 0.f));
                        A.out[b * A.out_bstride + (long)co * A.Lout + l] = o;
                        if constexpr (LF) {
                            const long n = (long)A.B * A.Cout * A.Lout, ee = ((long)b * A.Cout + co) * A.Lout + l;
                            const float xn = loop_update(lfv.sampler, lfv.clip, lfv.eta, lfrow, o, lfv.x[ee], lfv.lp->noise[(long)(1 + lfk) * n + ee]);
                            lfv.x[ee] = xn;
                            if (lfv.lp->traj) lfv.lp->traj[(long)lfk * n + ee] = xn;
                        }
                    }
                }
        }
    } else {
    // ---- sum the two product streams, then the k-parts of the workgroup (LDS) ------------------------
    f32x16 acc[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = ONE_ACC ? acc_hh[t][r] : acc_sm[0][r] + acc_hh[t][r];
    if constexpr (!WT) {
        lds_bar();
        if (kpart > 0) {
            float *dst = red + ((kpart - 1) * nct + ct) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r * 64 + lane] = acc[0][r];
        }
        lds_bar();
        if (kpart == 0) {
            for (int kp = 1; kp < KP; ++kp) {
                const float *srcp = red + ((kp - 1) * nct + ct) * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += srcp[r * 64 + lane];
            }
        }
    }
    const bool owner = WT ? tile_ok : kpart == 0;      // this wave holds a finished tile of the workgroup's K slice
    C2_STAMP(7);
#ifdef SURFD_C2_PROBE
    {
        C2_PROBE_ADD(4, tile_ok ? probe_w : 0ull);
        unsigned long long h = 0ull;
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h += (unsigned long long)__float_as_uint(acc[t][r]) * (unsigned long long)(2 * ((tid * NCT + t) * 16 + r) + 1);
        C2_PROBE_ADD(5, owner ? h : 0ull);
    }
#endif
    // ---- cross-workgroup K reduction (hand-off recipe R1, cdna_hip_programming.md §6 G16): write-through
    //      partial tiles -> vmcnt(0) -> barrier -> relaxed ticket; the last arriver acquires and sums in slice order
    if (A.KS > 1) {
        const size_t slot = (size_t)by * A.nrt + rg;
        float *mine = A.part + (((size_t)kz * A.nby + by) * A.ntiles + tile) * A.part_stride;
        if (owner) {
#pragma unroll
            for (int t = 0; t < NCT; ++t)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 val = {acc[t][4 * r4], acc[t][4 * r4 + 1], acc[t][4 * r4 + 2], acc[t][4 * r4 + 3]};
                    float *dst = mine + ((size_t)((NT2 ? t : ct) * 4 + r4) * 64 + lane) * 4;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(val) : "memory");
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *flag = reinterpret_cast<int *>(ALIAS ? red : red + 4 * 1024);
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
        if (owner) {
#pragma unroll
            for (int t = 0; t < NCT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            // the partial tiles of C2_ZB slices are requested together and summed in slice order (round 6).  One slice at a
            // time — what the rolled loop of rounds 3-5 compiled to: four loads, a wait, the adds, the branch — is KS
            // consecutive round trips to another XCD's write-through data (~1 us each: the split-K phase of the stamps, 3-7 us
            // of a 20-30 us workgroup life at KS = 4 ... 9).  Loads past the last slice re-read it (unconditional, as everywhere
            // in this kernel); the order of the additions, hence every bit of the result, is unchanged.
#ifndef SURFD_C2_ZB
#define SURFD_C2_ZB 4
#endif
            // Measured (one box, profiles/r06_loop_ab_split_k_batch.json; bits identical in every row): latency form 1.364 -> 1.338 ms
            // per evaluation at 8 latents with four slices in flight (8: 1.346, 2: 1.375); the lean form gains nothing (three
            // workgroups per CU already cover each other's round trips: 18.9 us either way) and at four slices spills its 168th
            // register again, so it keeps one slice at a time — as does the two-column-tile form, which has no register to spare.
            constexpr int C2_ZB = (LEAN || NT2) ? 1 : SURFD_C2_ZB;
            for (int z0 = 0; z0 < A.KS; z0 += C2_ZB) {
                f32x4 pv[C2_ZB][NCT][4];
#pragma unroll
                for (int zz = 0; zz < C2_ZB; ++zz) {
                    const int z = min(z0 + zz, A.KS - 1);
                    const float *src = A.part + (((size_t)z * A.nby + by) * A.ntiles + tile) * A.part_stride;
#pragma unroll
                    for (int t = 0; t < NCT; ++t)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) pv[zz][t][r4] = c2_load_partial(src + ((size_t)((NT2 ? t : ct) * 4 + r4) * 64 + lane) * 4);
                }
#pragma unroll
                for (int zz = 0; zz < C2_ZB; ++zz)
                    if (z0 + zz < A.KS) {          // wave-uniform
#pragma unroll
                        for (int t = 0; t < NCT; ++t)
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                                for (int q = 0; q < 4; ++q) acc[t][4 * r4 + q] += pv[zz][t][r4][q];
                    }
            }
        }
    }
    C2_STAMP(8);
    // ---- epilogue ---------------------------------------------------------------------------------------
    if (owner)
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
        const int ml = (NT2 ? t : ct) * 32 + (lane & 31), m = ml + m_off;
        const bool mok = ml < M;
        const int b = b0 + (m >> A.log2Lout), l = m & (A.Lout - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = tile * 32 + frag_row(r, lane);
            if (mok && co < A.Cout) {
                const float val = acc[t][r] * inv_sc + ((pre_b[r >> 2][r & 3] + (A.has_emb ? pre_e[t][r >> 2][r & 3] : 0.f)) + (A.has_res ? pre_r[t][r] : 0.f));
                A.out[b * A.out_bstride + (long)co * A.Lout + l] = val;
#ifdef SURFD_C2_PROBE
                C2_PROBE_ADD(6, (unsigned long long)__float_as_uint(val) * (unsigned long long)(2 * ((tid * NCT + t) * 16 + r) + 1));
                C2_PROBE_ADD(7, ((unsigned long long)__float_as_uint(pre_b[r >> 2][r & 3]) + 3ull * __float_as_uint(A.has_emb ? pre_e[t][r >> 2][r & 3] : 0.f) + 5ull * __float_as_uint(A.has_res ? pre_r[t][r] : 0.f))
                                    * (unsigned long long)(2 * ((tid * NCT + t) * 16 + r) + 1));
#endif
                if constexpr (LF) {
                    // x0 prediction -> x_{t-1}, in place (this element of x is read and written by this thread only)
                    const long n = (long)A.B * A.Cout * A.Lout, e = ((long)b * A.Cout + co) * A.Lout + l;
                    const float xn = loop_update(lfv.sampler, lfv.clip, lfv.eta, lfrow, val, lfv.x[e], lfv.lp->noise[(long)(1 + lfk) * n + e]);
                    lfv.x[e] = xn;
                    if (lfv.lp->traj) lfv.lp->traj[(long)lfk * n + e] = xn;
                }
            }
        }
    }
    }
    if constexpr (LF) {
        // the last workgroup to get here advances the loop counter: every workgroup of this launch that reads it (above) has
        // done so before it arrives
        __syncthreads();
        if (tid == 0) {
            const int prev = __hip_atomic_fetch_add(lfv.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == A.nby * A.nrt - 1) {
                __hip_atomic_store(lfv.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(lfv.step, lfk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    {   // the prefetched words' only "use": the compiler counts them as loads in flight until here
        unsigned used = 0u;
#pragma unroll
        for (int k = 0; k < C2_PFN_N; ++k) used |= pf_sink[k];
        asm volatile("" :: "v"(used));
    }
#ifdef SURFD_C2_STAMPS
    C2_STAMP(9);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    C2_STAMP(10);
    if (A.dbg && blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 11; ++i) A.dbg[i] = stamp_[i];
        A.dbg[15] = (long long)__builtin_readcyclecounter() - cyc0_;       // shader cycles over the same span
    }
#endif
}

// developer aid (SURFD_CONV_DEBUG=1 SURFD_CONV2_HASH=1): order-independent checksum of a launch's output into slot 0 of its
// stamp record — run-to-run comparison of every layer without a host synchronisation between the launches
__global__ void c2_hash_kernel(const float *p, long bstride, int B, int row, unsigned long long *out) {
    unsigned long long h = 0ull;
    const long n = (long)B * row;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long b = e / row, r = e - b * row;
        h += (unsigned long long)__float_as_uint(p[b * bstride + r]) * (unsigned long long)(2 * e + 1);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

// ---------------------------------------------------------------------------------------------
// weight preparation
// ---------------------------------------------------------------------------------------------
__global__ void c2_absmax_kernel(const float *src, size_t n, unsigned *out) {
    unsigned m = 0u;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(src[i]) & 0x7fffffffu);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// sc = {SC, 1/SC, max|W| bits}: SC = 2^(8 - floor(log2 max|W|)), i.e. max |W| * SC in [256, 512)
__global__ void c2_scale_kernel(float *sc) {
    const unsigned b = __float_as_uint(sc[2]);
    int e = (int)(b >> 23) - 127;
    if (b == 0u || e > 100 || e < -100) e = 8;         // degenerate weights: SC = 1
    sc[0] = __uint_as_float((unsigned)(127 + 8 - e) << 23);
    sc[1] = __uint_as_float((unsigned)(127 - 8 + e) << 23);
}

struct PackSeg2 { int C, Cp8, taps, blk, blkp, k16_off, kg_off8; };

// fp32 fragment-major pack of one layer (common.h) -> two fragment-major fp16 planes of SC * W in the K-block
// order of the f16x2 kernel; padded channels are zero
__global__ void c2_pack_kernel(const float *wp, int KGtot, int ntiles, PackSeg2 s0, PackSeg2 s1, int nseg, int KS16,
                               const float *sc, _Float16 *dst) {
    const float scale = sc[0];
    const long total = (long)ntiles * KS16 * 512;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int el = e & 7, lane = (e >> 3) & 63;
        const long tk = e >> 9;
        const int ks = (int)(tk % KS16), tile = (int)(tk / KS16);
        const PackSeg2 S = (nseg > 1 && ks >= s1.k16_off) ? s1 : s0;
        const int nk = S.blkp >> 4;
        const int local = ks - S.k16_off, per_blk = S.taps * nk;
        const int bi = local / per_blk, r = local - bi * per_blk;
        const int tap = r / nk, kk = r - tap * nk;
        const int cb = kk * 16 + 8 * (lane >> 5) + el, c = bi * S.blk + cb;
        float x = 0.f;
        if (cb < S.blk && c < S.C) {
            const int kg = S.kg_off8 + tap * (S.Cp8 >> 3) + (c >> 3);
            x = wp[((size_t)((size_t)tile * KGtot + kg) * 64 + (lane & 31) + 32 * ((c >> 2) & 1)) * 4 + (c & 3)] * scale;
        }
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        const size_t base = ((size_t)tile * KS16 + ks) * 1024 + (size_t)lane * 8 + el;
        dst[base] = h; dst[base + 512] = l;
    }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
static int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// K blocking of one segment: a block is what one workgroup stages at a time (thread <-> channel, <= 256),
// holds whole GroupNorm groups and is padded to a multiple of 16 channels in the packed K axis.
// Group slots of the staging's thread <-> channel map (SURFD_C2_GNW): a GroupNorm group of gs channels takes 2^log2P >= gs
// consecutive lanes of one wave, a wave holds 64 >> log2P slots, `waves` waves stage a K block.
static int slot_log2(int gs) { int lp = 0; while ((1 << lp) < gs) ++lp; return lp; }
static int max_groups_per_block(int gs, int waves) {
#if SURFD_C2_GNW
    if (gs > 64) return 0;
    return waves * (64 >> slot_log2(gs));
#else
    (void)gs; (void)waves;
    return 1 << 20;
#endif
}

static bool seg_blocking(const SegPlan &sp, int &blk, int &blkp, int &nblk) {
    const int C = sp.C;
    if (sp.gn) {
        if (C % 32) return false;
        const int gs = C / 32;
        const int unit = gs * 8 / gcd_i(gs, 8);           // lcm(gs, 8)
        const int gmax = max_groups_per_block(gs, 4);
        if (unit > 256 || C % unit || unit / gs > gmax) return false;
        int m = 1;
        while (unit * m * 2 <= 256 && (C / unit) % (m * 2) == 0 && unit * m * 2 / gs <= gmax) m *= 2;
        blk = unit * m;
    } else {
        blk = 0;
        for (int b = std::min(256, C); b >= 1; --b)
            if (C % b == 0 && (b % 16 == 0 || b == C)) { blk = b; break; }
        if (!blk) return false;
        if (blk < 64 && C > 256) return false;             // awkward channel counts: leave to the fp32 kernel
    }
    blkp = ceil_div(blk, 16) * 16;
    nblk = C / blk;
    return true;
}

// K blocking for the two-column-tile form: <= 128 channels per block (two staging threads per channel), whole GroupNorm
// groups, a power-of-two number of the 32 groups per block so that the blocks tile the channel axis
static bool seg_blocking2(const SegPlan &sp, int &blk, int &blkp, int &nblk) {
    const int C = sp.C;
    blk = 0;
    if (sp.gn) {
        if (C % 32) return false;
        const int gs = C / 32;
        const int gmax = max_groups_per_block(gs, 2);
        if (gs > 128 || gmax < 1) return false;
        int m = 32;
        while (m > 1 && (gs * m > 128 || m > gmax)) m >>= 1;
        blk = gs * m;
    } else {
        for (int b = std::min(128, C); b >= 1; --b)
            if (C % b == 0 && (b % 16 == 0 || b == C)) { blk = b; break; }
        if (!blk || (blk < 64 && C > 128)) return false;
    }
    blkp = ceil_div(blk, 16) * 16;
    nblk = C / blk;
    return true;
}

int conv2_plan_layout(surfd_unet *u) {
    size_t off = 0, off2 = 0;
    int nsc = 0, id = 0;
    for (auto &op : u->ops) {
        if (op.kind != 0) continue;
        ConvPlan &c = op.conv;
        c.id = id++;
        c.f16_ok = 1;
        int k16 = 0;
        for (int s = 0; s < c.nseg; ++s) {
            if (!seg_blocking(c.seg[s], c.blk[s], c.blkp[s], c.nblk[s])) { c.f16_ok = 0; break; }
            c.k16_off[s] = k16;
            k16 += c.nblk[s] * c.seg[s].taps * (c.blkp[s] / 16);
        }
        if (!c.f16_ok) continue;
        c.KS16 = k16;
        c.whf_off = off;
        off += (size_t)ceil_div(c.Cout, 32) * k16 * 1024;
        c.sc_idx = nsc++;
        c.f16_ok2 = 1;
        int k16b = 0;
        for (int s = 0; s < c.nseg; ++s) {
            if (!seg_blocking2(c.seg[s], c.blk2[s], c.blkp2[s], c.nblk2[s])) { c.f16_ok2 = 0; break; }
            c.k16_off2[s] = k16b;
            k16b += c.nblk2[s] * c.seg[s].taps * (c.blkp2[s] / 16);
        }
        if (c.f16_ok2) {
            c.KS16_2 = k16b;
            c.whf2_off = off2;
            off2 += (size_t)ceil_div(c.Cout, 32) * k16b * 1024;
        }
    }
    u->whf_halfs = off;
    u->whf2_halfs = off2;
    u->n_sc = nsc;
    return SURFD_OK;
}

int conv2_finalize(surfd_unet *u, hipStream_t st) {
    if (!u->whf_halfs) return SURFD_OK;
    if (!u->whf2 && u->whf2_halfs) HIP_TRY(hipMalloc((void **)&u->whf2, u->whf2_halfs * sizeof(_Float16)));
    if (!u->whf) {
        HIP_TRY(hipMalloc((void **)&u->whf, u->whf_halfs * sizeof(_Float16)));
        HIP_TRY(hipMalloc((void **)&u->wsc, (size_t)u->n_sc * 4 * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&u->sat, sizeof(unsigned)));
        HIP_TRY(hipMemsetAsync(u->sat, 0, sizeof(unsigned), st));
    }
    HIP_TRY(hipMemsetAsync(u->wsc, 0, (size_t)u->n_sc * 4 * sizeof(float), st));
    for (auto &op : u->ops) {
        if (op.kind != 0 || !op.conv.f16_ok) continue;
        const ConvPlan &c = op.conv;
        const int ntiles = ceil_div(c.Cout, 32);
        float *sc = u->wsc + (size_t)c.sc_idx * 4;
        const size_t nfl = (size_t)ntiles * c.KGtot * 256;
        hipLaunchKernelGGL(c2_absmax_kernel, dim3((unsigned)std::min<size_t>(ceil_div<size_t>(nfl, 1024), 512)), dim3(256), 0, st,
                           (const float *)(u->wpack + c.w_off), nfl, reinterpret_cast<unsigned *>(sc + 2));
        LAUNCH_CHECK();
        hipLaunchKernelGGL(c2_scale_kernel, dim3(1), dim3(1), 0, st, sc);
        LAUNCH_CHECK();
        PackSeg2 ps[2];
        memset(ps, 0, sizeof(ps));
        for (int s = 0; s < c.nseg; ++s) {
            ps[s].C = c.seg[s].C; ps[s].Cp8 = ceil_div(c.seg[s].C, 8) * 8; ps[s].taps = c.seg[s].taps;
            ps[s].blk = c.blk[s]; ps[s].blkp = c.blkp[s]; ps[s].k16_off = c.k16_off[s]; ps[s].kg_off8 = c.kg_off[s];
        }
        const long total = (long)ntiles * c.KS16 * 512;
        hipLaunchKernelGGL(c2_pack_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(total, 256), 4096)), dim3(256), 0, st,
                           (const float *)(u->wpack + c.w_off), c.KGtot, ntiles, ps[0], ps[1], c.nseg, c.KS16,
                           (const float *)sc, u->whf + c.whf_off);
        LAUNCH_CHECK();
        if (c.f16_ok2) {          // the same scaled weights in the second K blocking
            for (int s = 0; s < c.nseg; ++s) { ps[s].blk = c.blk2[s]; ps[s].blkp = c.blkp2[s]; ps[s].k16_off = c.k16_off2[s]; }
            const long total2 = (long)ntiles * c.KS16_2 * 512;
            hipLaunchKernelGGL(c2_pack_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(total2, 256), 4096)), dim3(256), 0, st,
                               (const float *)(u->wpack + c.w_off), c.KGtot, ntiles, ps[0], ps[1], c.nseg, c.KS16_2,
                               (const float *)sc, u->whf2 + c.whf2_off);
            LAUNCH_CHECK();
        }
    }
    // the kernels take 1/SC by value (launch_conv2): one read-back per finalize
    u->wsc_host.assign((size_t)u->n_sc * 4, 0.f);
    HIP_TRY(hipMemcpyAsync(u->wsc_host.data(), u->wsc, u->wsc_host.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return SURFD_OK;
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

int launch_conv2(surfd_unet *u, const ConvPlan &c, int B, int L, const ConvLaunchIO &io, hipStream_t st) {
    if (!c.f16_ok || !u->whf) return 1;
    Conv2Args A;
    memset(&A, 0, sizeof(A));
    A.nseg = c.nseg; A.Cout = c.Cout; A.B = B;
    A.Lout = c.ds_out ? L / c.ds_out : 1;
    if (A.Lout < 1 || (A.Lout & (A.Lout - 1))) return 1;
    while ((1 << A.log2Lout) < A.Lout) ++A.log2Lout;
    auto resolve = [&](const View &v, int ds, bool is_out, float *&ptr, long &bs) {
        const int len = ds ? L / ds : 1;
        if (v.buf >= 0) {
            ptr = u->buf_ptr[v.buf] + (long)v.choff * len;
            bs = (long)u->bufs[v.buf].C * len;
        } else if (is_out) { ptr = io.ext_out; bs = io.ext_out_bs; }
        else { ptr = const_cast<float *>(io.ext_in); bs = io.ext_in_bs; }
    };
    const bool wide = u->wide_batch > 0;
    // two column tiles per wave (second weight layout): loops designed for >= 128 latents, stride-1 layers of <= 32 positions
    static const int nt2_env = env_int("SURFD_CONV2_NT2", 1), nt2_min = env_int("SURFD_CONV2_NT2_MIN", 128);
    bool nt2 = wide && nt2_env && u->wide_batch >= nt2_min && c.f16_ok2 && u->whf2 && !(io.lf && c.dst.buf == -3) && A.Lout <= 32;
    for (int s = 0; s < c.nseg && nt2; ++s)
        nt2 = c.seg[s].stride == 1 && !c.seg[s].ups && c.seg[s].ds == c.ds_out && c.seg[s].ds != 0;
    int Lin0 = 0, max_lsl = 1, max_blkp = 16;
    for (int s = 0; s < c.nseg; ++s) {
        const SegPlan &sp = c.seg[s];
        if (!sp.ds) return 1;                       // Linear layers of the embedding path stay on the fp32 kernel
        Seg2 &S = A.seg[s];
        float *p; long bs;
        resolve(sp.src, sp.ds, false, p, bs);
        S.x = p; S.bstride = bs;
        S.C = sp.C; S.Lin = L / sp.ds;
        if (S.Lin < 4 || S.Lin > 64 || (S.Lin & (S.Lin - 1))) return 1;
        while ((1 << S.log2Lin) < S.Lin) ++S.log2Lin;
        if (s == 0) Lin0 = S.Lin; else if (S.Lin != Lin0) return 1;
        S.taps = sp.taps; S.stride = sp.stride; S.ups = sp.ups; S.gn = sp.gn; S.act = sp.act;
        S.gamma = S.beta = u->vecs;                  // always loaded (unconditional loads), used only with GroupNorm
        if (sp.gn) {
            S.gamma = u->vecs + u->vec_off[sp.gnkey + ".weight"]; S.beta = u->vecs + u->vec_off[sp.gnkey + ".bias"];
            S.gs = sp.C / 32;
        }
        S.blk = c.blk[s]; S.blkp = c.blkp[s]; S.nblk = c.nblk[s]; S.k16_off = c.k16_off[s];
        if (nt2) { S.blk = c.blk2[s]; S.blkp = c.blkp2[s]; S.nblk = c.nblk2[s]; S.k16_off = c.k16_off2[s]; }
        S.log2P = 6; S.gsm = 64;                     // thread = channel
#if SURFD_C2_GNW
        if (sp.gn) { S.log2P = slot_log2(S.gs); S.gsm = S.gs; }      // one group per slot of 2^log2P lanes (seg_blocking: the block's groups fit its waves)
#endif
        const int lsl = sp.taps == 3 ? (sp.stride == 2 ? 2 * A.Lout + 1 : A.Lout + 2) : A.Lout;
        max_lsl = std::max(max_lsl, lsl);
        max_blkp = std::max(max_blkp, S.blkp);
    }
    A.Lsl = max_lsl;
    const int VEC = Lin0 == 64 ? 16 : 8;
    // wide form: one column tile per workgroup (M <= 32); a layer with 64 output positions (L = 64 latents) is computed as two
    // 32-column halves of one sample per workgroup pair, each staging the sample's whole row
    static const int split_env = env_int("SURFD_CONV2_WIDE64", 1);   // 0: 64-position layers stay in the latency form (A/B timing)
    const bool split = wide && A.Lout == 64 && split_env;
    const bool wt = wide && (A.Lout <= 32 || split);
    static const int lean_env = env_int("SURFD_CONV2_LEAN", 1);      // 0: the two-workgroups-per-CU wide kernel (A/B timing)
    const bool fuse_head = io.lf && c.dst.buf == -3;                 // 80 workgroups, once per evaluation: keeps the 256-register form (no spills with the loop record live)
    // lean form (three workgroups per CU) wherever one batch entry's slab fits its 20 KB planes
    nt2 = nt2 && wt && VEC == 8 && lean_env;
    bool lean = nt2 || (wt && VEC == 8 && lean_env && !fuse_head && (size_t)A.Lsl * (max_blkp + 8) <= (size_t)C2_PLANE_LEAN);
    const int nb_cap = nt2 ? std::max(1, 64 / A.Lout) : split ? 1 : std::min({(VEC * 4) / Lin0, std::max(1, (wt ? 32 : 64) / A.Lout), 8});
    int nb = std::min(B, nb_cap);
    // one fp16 plane of the slab has a fixed size (the kernel addresses the low plane with an immediate): 30 KB
    // (36 KB for 64-long rows), i.e. <= 78 KB of LDS per workgroup so that two of them share a CU
    const size_t plane_halfs = nt2 ? C2_PLANE_NT2 : lean ? C2_PLANE_LEAN : (VEC == 16 ? 18432 : 15360);
    while (nb > 1 && (size_t)nb * A.Lsl * (max_blkp + 8) > plane_halfs) --nb;
    if ((size_t)nb * A.Lsl * (max_blkp + 8) > plane_halfs) return 1;
    if (!split && nb * A.Lout > ((wt && !nt2) ? 32 : 64)) return 1;
    A.nhalf = split ? 2 : 1;
    A.bchunk = nb;
    A.cs = max_blkp + 8;
    A.plane = (int)plane_halfs;
    size_t lds = plane_halfs * 2 * sizeof(_Float16);
    lds = (lds + 15) & ~(size_t)15;
    // GroupNorm exchange area (staging) and k-part reduction scratch (after the last MFMA) are never live together
    A.off_ex = (int)lds;
    A.off_red = (int)lds;
    if (lean || (wt && VEC == 16)) lds += 16;      // the split-K flag; the GroupNorm exchange arrays alias the slab (2 x 8 (16) KB + 2 KB <= 2 planes)
#if SURFD_C2_GNW
    else lds += (size_t)(4 * 1024 + 4) * sizeof(float);       // k-part reduction scratch (latency form, distributed tail: every k-part leaves its tile) + flag
#else
    else lds += std::max(((size_t)2 * VEC * 256 + 2 * 8 * 32) * sizeof(float), (size_t)(3 * 1024 + 4) * sizeof(float));
#endif
    static const int lds_extra = env_int("SURFD_CONV2_LDS_EXTRA", 0);   // developer aid: fewer workgroups per CU (occupancy experiments)
    lds += (size_t)lds_extra;
    lds = (lds + 255) & ~(size_t)255;
    A.off_pf = (int)lds;               // landing area of the weight prefetch (256 bytes, never read)
    lds += 256;
    if (lds > 160 * 1024) return 1;
    A.whf = u->whf + c.whf_off; A.KS16 = c.KS16;
    if (nt2) { A.whf = u->whf2 + c.whf2_off; A.KS16 = c.KS16_2; }
#if !(defined(SURFD_C2_PROBE) || defined(SURFD_C2_DBG_POISON))
    A.sc = u->wsc + (size_t)c.sc_idx * 4;
#endif
    A.inv_sc = u->wsc_host[(size_t)c.sc_idx * 4 + 1];
    A.bias = u->vecs + c.bias_off;
    A.emb = A.bias; A.emb_bstride = 0; A.res = A.bias; A.res_bstride = 0; A.res_cstride = 1; A.res_lstride = 0;
    if (c.emb_off >= 0 && io.emb) {
        A.emb = io.emb + c.emb_off; A.step_ptr = io.step_ptr;
        A.emb_bstride = u->emb_shared ? 0 : io.emb_bs; A.emb_step_stride = u->emb_shared ? io.emb_bs : (long)B * io.emb_bs;
        if (u->emb_ingraph) { A.step_ptr = nullptr; A.emb_step_stride = 0; }      // the iteration's own [B][14112] rows (unet_forward_prepared)
        A.has_emb = 1;
    }
    if (c.res.buf != -1) {
        float *p; long bs; resolve(c.res, c.ds_out, false, p, bs);
        A.res = p; A.res_bstride = bs; A.res_cstride = A.Lout; A.res_lstride = 1; A.has_res = 1;
    }
    { float *p; long bs; resolve(c.dst, c.ds_out, true, p, bs); A.out = p; A.out_bstride = bs; }
    A.ntiles = ceil_div(c.Cout, 32);
    A.nby = ceil_div(B, nb) * A.nhalf;
    // K slices over workgroups when (tiles x batch chunks) under-fills the chip: whole K blocks per slice
    const int nch = nt2 ? c.nblk2[0] + (c.nseg > 1 ? c.nblk2[1] : 0) : c.nblk[0] + (c.nseg > 1 ? c.nblk[1] : 0);
    static const int ks_fill_env = env_int("SURFD_CONV2_FILL", 0);
    // workgroups aimed at: as many as fit the chip at once — two per CU (measured: 1.555 -> 1.472 ms per evaluation), three in the
    // lean form (80 latents, design batch 80: 2.49 -> 2.43 ms per evaluation alone, 2 x 80: 21.6 -> 20.6 us per evaluation and latent)
    const int ks_fill = ks_fill_env ? ks_fill_env : (lean ? SURFD_C2_LEAN_WAVES : 2) * u->cu_budget;
    static const int ks_max = env_int("SURFD_CONV2_KSMAX", 16);
    static const int ks_min_base = env_int("SURFD_CONV2_NOSPLIT_ABOVE", 200);
    const int base = A.ntiles * A.nby;
    A.nrt = wt ? ceil_div(A.ntiles, 4) : A.ntiles;
    int KS = 1;
    A.part_stride = (wt && !nt2) ? 1024 : 2 * 1024;      // floats per partial tile: one column tile (wide form) or up to two
    if (wide) {
        // the K split is a function of the LAYER and of the handle's design batch only — never of B — so that a latent's
        // result does not depend on the width of the batch it rides in
        int nbd = std::min(u->wide_batch, nb_cap);
        while (nbd > 1 && (size_t)nbd * A.Lsl * (max_blkp + 8) > plane_halfs) --nbd;
        const int based = A.nrt * ceil_div(u->wide_batch, nbd) * A.nhalf;
        if (nch > 1) KS = std::min({nch, ks_max, std::max(1, ks_fill / based)});
        if ((size_t)KS * base * A.part_stride > u->part_floats || (long)A.nby * A.nrt > 8192)
            SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "conv (wide form): batch of %d needs more split-K scratch than the handle holds", B);
    } else {
        if (base < ks_min_base && nch > 1) KS = std::min({nch, ks_max, std::max(1, ks_fill / base)});
        if ((size_t)KS * base * A.part_stride > u->part_floats || base > 8192) KS = 1;
    }
    A.KS = KS;
    // Deep form for launches that do not fill the lean form's slots anyway.  Decomposition, batch chunks and K split stay the
    // lean form's (bits unchanged); what changes is the kernel: two workgroups per CU with 232 registers each instead of three
    // with 168, i.e. a weight ring of SURFD_C2_DEEP_D x 4 k16 steps instead of 2 x 2.  A wave of the wide form runs its whole K
    // slice against the ring, and with 4 steps in flight every step waits for a round trip (stamps: 214 ns per k16 step cold,
    // 136 ns from a warm L2, against 40 ns of matrix time) — workgroups that have a CU to themselves gain nothing from slots
    // they do not use.  SURFD_CONV2_DEEP_BELOW = workgroups per CU (x cu_budget) up to which a launch takes the deep kernel.
    static const int deep_below = env_int("SURFD_CONV2_DEEP_BELOW", 0);
    if (lean && !nt2 && deep_below > 0 && (long)A.nrt * KS * A.nby <= (long)deep_below * u->cu_budget) {
        lean = false;
        lds = (size_t)15360 * 2 * sizeof(_Float16);
        lds = (lds + 15) & ~(size_t)15;
        A.plane = 15360; A.off_ex = (int)lds; A.off_red = (int)lds;
#if SURFD_C2_GNW
        lds += (size_t)(4 * 1024 + 4) * sizeof(float);
#else
        lds += std::max(((size_t)2 * VEC * 256 + 2 * 8 * 32) * sizeof(float), (size_t)(3 * 1024 + 4) * sizeof(float));
#endif
        lds += (size_t)lds_extra;
        lds = (lds + 255) & ~(size_t)255;
        A.off_pf = (int)lds;
        lds += 256;
    }
    A.part = u->part; A.counters = u->counters;
    A.sat = u->sat;
    if (fuse_head) {
        // needs one epilogue per (batch chunk, row group) — true for every decomposition — and contiguous [B, Cout, L] output
        if (io.ext_out_bs != (long)c.Cout * A.Lout) SURFD_FAIL(SURFD_ERR_ARG, "conv: fused posterior update needs a contiguous head output");
        A.lf = io.lf;
        if (io.lf_done) *io.lf_done = true;
    }
    A.dbg = nullptr;
    if (u->dbg && u->dbg_launch < 4096) {
        A.dbg = u->dbg + (size_t)(u->dbg_launch++) * 16;
        long long meta[4] = {c.Cout, c.seg[0].C + (c.nseg > 1 ? c.seg[1].C : 0), A.Lout * 100000LL + (long long)A.ntiles * A.nby * KS, KS * 100 + nch};
        HIP_TRY(hipMemcpyAsync(A.dbg + 11, meta, sizeof(meta), hipMemcpyHostToDevice, st));
    }
    A.magic_nby = (unsigned)((0x100000000ULL + A.nby - 1) / A.nby);
    A.magic_ks = (unsigned)((0x100000000ULL + KS - 1) / KS);
    const int G = A.nrt * KS;
    A.magic_g = (unsigned)((0x100000000ULL + G - 1) / G);
    if ((long)G * A.nby >= 65536) SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "conv: %d x %d workgroups exceed the kernel's block decode", G, A.nby);
    dim3 grid((unsigned)(G < 8 ? G * A.nby : 8 * ceil_div(G, 8) * A.nby));
    static const int pref = env_int("SURFD_CONV2_PREF", 0);      // operand prefetch across K blocks: measured 1.472 (on) vs 1.442 ms (off) per evaluation
    static const int wpref = env_int("SURFD_CONV2_WIDE_PREF", 0);
    // this launch's shape for the launch before it (next evaluation on: the loop replays the same launches), and the next
    // convolution's shape for this one
    {
        ConvPlan::LaunchRec &R = c.rec;
        R.gen = u->ws_gen; R.B = B; R.L = L; R.whf = A.whf; R.ntiles = A.ntiles; R.KS16 = A.KS16; R.tpg = wt ? 4 : 1; R.KS = KS; R.G = A.nrt * KS;
        R.nblk0 = A.seg[0].nblk; R.it0 = A.seg[0].taps * (A.seg[0].blkp >> 4);
        R.nch = A.seg[0].nblk + (c.nseg > 1 ? A.seg[1].nblk : 0); R.it1 = c.nseg > 1 ? A.seg[1].taps * (A.seg[1].blkp >> 4) : 0;
    }
    // nothing to request (first evaluation, other shape, switched off): every thread re-reads the first line of this launch's own
    // weights — the loads are unconditional by design
    A.pf_w = reinterpret_cast<const char *>(A.whf);
    A.pf_ntiles = 0; A.pf_KS16 = 0; A.pf_KS = 1; A.pf_G = 1; A.pf_nblk0 = 1; A.pf_it0 = 1; A.pf_nch = 1; A.pf_it1 = 1;
    A.pf_log2tpg = 0; A.pf_log2bps = 0; A.pf_log2lpb = 4; A.pf_magic_ks = 0u;
    static const int pfn_env = env_int("SURFD_CONV2_PFN", 1);
    // Requested where it pays: the request costs this launch ~1 us of issue time (two loads of 64 different lines per wave in
    // front of the wait for the operand), the next launch's K loop gets ~35 % shorter — 2-5 us on a three-tap convolution over
    // 224-channel blocks (42 k16 steps per block), 0.9 us on a 1 x 1 convolution (14): the wide form requests when the next
    // launch runs >= SURFD_CONV2_PFN_MIN k16 steps per wave; the latency form (a wave's k-part is ~10 steps, 8 of them in
    // flight before the operand is staged) never does (1.40 against 1.38 ms per evaluation at 8 latents).
    static const int pfn_min = env_int("SURFD_CONV2_PFN_MIN", 28);
    if (pfn_env && wt && u->pf_next && u->pf_next != &c) {
        const ConvPlan::LaunchRec &R = u->pf_next->rec;
        const int steps_per_wave = ceil_div(R.nch, std::max(R.KS, 1)) * std::max(R.it0, R.it1);
        if (R.gen == u->ws_gen && R.B == B && R.L == L && R.whf && R.tpg == 4 && steps_per_wave >= pfn_min) {
            A.pf_w = reinterpret_cast<const char *>(R.whf);
            A.pf_ntiles = R.ntiles; A.pf_KS16 = R.KS16; A.pf_KS = R.KS; A.pf_G = R.G;
            A.pf_nblk0 = R.nblk0; A.pf_it0 = R.it0; A.pf_nch = R.nch; A.pf_it1 = std::max(R.it1, 1);
            auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
            A.pf_log2tpg = lg(R.tpg); A.pf_log2bps = lg(ceil_div(R.nch, R.KS)); A.pf_log2lpb = lg(16 * std::max(R.it0, R.it1));
            A.pf_magic_ks = (unsigned)((0x100000000ULL + R.KS - 1) / R.KS);
        }
    }
#if defined(SURFD_C2_PROBE) || defined(SURFD_C2_DBG_POISON)
    {
        static unsigned long long *const probe_base = [] { const char *e = getenv("SURFD_CONV2_PROBE_PTR"); return e ? reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0)) : nullptr; }();
        A.probe = (probe_base && grid.x <= 4096 && c.id < 128) ? probe_base + (size_t)c.id * 4096 * 8 : nullptr;
        A.C2_LDS_BYTES = (int)lds;
    }
#endif
    auto launch = [&]() {
        if (A.lf) {         // the head of a graph-replayed loop: same decompositions, posterior update in the epilogue
            if (wt && VEC == 16) hipLaunchKernelGGL((conv2_kernel<16, false, true, false, true>), grid, dim3(256), lds, st, A);
            else if (wt) hipLaunchKernelGGL((conv2_kernel<8, false, true, false, true>), grid, dim3(256), lds, st, A);
            else if (VEC == 16) hipLaunchKernelGGL((conv2_kernel<16, false, false, false, true>), grid, dim3(256), lds, st, A);
            else hipLaunchKernelGGL((conv2_kernel<8, false, false, false, true>), grid, dim3(256), lds, st, A);
        }
        else if (nt2) hipLaunchKernelGGL((conv2_kernel<8, false, true, true, false, true>), grid, dim3(256), lds, st, A);
        else if (lean) hipLaunchKernelGGL((conv2_kernel<8, false, true, true>), grid, dim3(256), lds, st, A);
        else if (wt && VEC == 16) hipLaunchKernelGGL((conv2_kernel<16, false, true>), grid, dim3(256), lds, st, A);
        else if (wt && wpref) hipLaunchKernelGGL((conv2_kernel<8, true, true>), grid, dim3(256), lds, st, A);
        else if (wt) hipLaunchKernelGGL((conv2_kernel<8, false, true>), grid, dim3(256), lds, st, A);
        else if (VEC == 16) hipLaunchKernelGGL((conv2_kernel<16, false>), grid, dim3(256), lds, st, A);
        else if (pref) hipLaunchKernelGGL((conv2_kernel<8, true>), grid, dim3(256), lds, st, A);
        else hipLaunchKernelGGL((conv2_kernel<8, false>), grid, dim3(256), lds, st, A);
    };
    launch();
    LAUNCH_CHECK();
    static const int hash_env = env_int("SURFD_CONV2_HASH", 0);
    if (hash_env && A.dbg) {
        HIP_TRY(hipMemsetAsync(A.dbg, 0, sizeof(long long), st));
        hipLaunchKernelGGL(c2_hash_kernel, dim3(64), dim3(256), 0, st, (const float *)A.out, (long)A.out_bstride, B, A.Cout * A.Lout,
                           reinterpret_cast<unsigned long long *>(A.dbg));
        LAUNCH_CHECK();
    }
    // developer aid (SURFD_CONV2_TWICE=1 with a -DSURFD_C2_STAMPS build): the same launch again, its stamps in the next slot — the
    // second run finds the weights it streams in the L2s the first one left them in (cold vs warm weight stream, per layer)
    static const int twice = env_int("SURFD_CONV2_TWICE", 0);
    if (twice && !A.lf) {
        if (A.dbg && u->dbg_launch < 4096) {
            long long *prev = A.dbg;
            A.dbg = u->dbg + (size_t)(u->dbg_launch++) * 16;
            HIP_TRY(hipMemcpyAsync(A.dbg + 11, prev + 11, 4 * sizeof(long long), hipMemcpyDeviceToDevice, st));
        }
        launch();
    }
    LAUNCH_CHECK();
    return SURFD_OK;
}

// ---- the compiled configuration (surfd_build_config) and the fence around variants recorded as unsafe -----------------------------
// Every experiment macro of this file with the value it was compiled with; `unsafe` counts the ones that select a variant known to
// give wrong or not bit-stable results (profiles/r05_loop_experiments.md, profiles/r06_conv2_instability.md) or a developer aid that
// changes the results.  Those do not compile unless SURFD_ALLOW_UNSAFE_VARIANTS is defined (tools/build_variants.py passes it for
// the hunt's builds; the product build never does).
#define C2_STR_(x) #x
#define C2_STR(x) C2_STR_(x)
#if defined(SURFD_C2_ABLATE)
#define C2_CFG_ABLATE SURFD_C2_ABLATE
#else
#define C2_CFG_ABLATE 0
#endif
#if defined(SURFD_C2_DBG_POISON)
#define C2_CFG_POISON SURFD_C2_DBG_POISON
#else
#define C2_CFG_POISON 0
#endif
#if defined(SURFD_C2_PROBE)
#define C2_CFG_PROBE 1
#else
#define C2_CFG_PROBE 0
#endif
#if defined(SURFD_C2_STAMPS)
#define C2_CFG_STAMPS 1
#else
#define C2_CFG_STAMPS 0
#endif
#define C2_UNSAFE_COUNT ((SURFD_C2_GNW != 1) + (SURFD_C2_GNPAD < 1) + (SURFD_C2_LAT_D != 2) + \
                         (SURFD_C2_BPIPE == 2) + (C2_CFG_ABLATE != 0) + (C2_CFG_POISON != 0) + (C2_CFG_PROBE != 0))
#if !defined(SURFD_ALLOW_UNSAFE_VARIANTS)
#if SURFD_C2_GNW != 1
#error "SURFD_C2_GNW=0 (GroupNorm statistics through the LDS exchange) gives timing-dependent wrong 1/sigma values when workgroups share a CU (profiles/r06_conv2_instability.md); -DSURFD_ALLOW_UNSAFE_VARIANTS builds it anyway"
#endif
#if SURFD_C2_GNPAD < 1
#error "SURFD_C2_GNPAD=0: the in-wave GroupNorm reductions without wait states are NOT bit-stable run to run (profiles/r05_loop_experiments.md section 3, profiles/r06_conv2_instability.md); -DSURFD_ALLOW_UNSAFE_VARIANTS builds it anyway"
#endif
#if SURFD_C2_LAT_D != 2
#error "SURFD_C2_LAT_D != 2 is not a validated configuration of the latency form (profiles/r05_loop_experiments.md section 1); -DSURFD_ALLOW_UNSAFE_VARIANTS builds it anyway"
#endif
#if SURFD_C2_BPIPE == 2
#error "SURFD_C2_BPIPE=2 is the control of an experiment (not bit-stable with SURFD_C2_GNW, profiles/r06_conv2_instability.md); -DSURFD_ALLOW_UNSAFE_VARIANTS builds it anyway"
#endif
#if C2_CFG_ABLATE != 0 || C2_CFG_POISON != 0 || C2_CFG_PROBE != 0
#error "developer aids of conv_f16x2.hip (SURFD_C2_ABLATE / _DBG_POISON / _PROBE) change results or timing; -DSURFD_ALLOW_UNSAFE_VARIANTS builds them anyway"
#endif
#endif
const char *conv2_build_config() {
    return "C2_GNW=" C2_STR(SURFD_C2_GNW) " C2_BPIPE=" C2_STR(SURFD_C2_BPIPE) " C2_GNPAD=" C2_STR(SURFD_C2_GNPAD) " C2_PFN=" C2_STR(SURFD_C2_PFN) " C2_PFN_FORMS=" C2_STR(SURFD_C2_PFN_FORMS)
           " C2_PFN_N=" C2_STR(SURFD_C2_PFN_N) " C2_KAPF=" C2_STR(SURFD_C2_KAPF)
           " C2_FAST_RCP=" C2_STR(SURFD_C2_FAST_RCP) " C2_EPI_LATE=" C2_STR(SURFD_C2_EPI_LATE) " C2_LAT_D=" C2_STR(SURFD_C2_LAT_D) " C2_DEEP_D=" C2_STR(SURFD_C2_DEEP_D)
           " C2_ZB=" C2_STR(SURFD_C2_ZB) " C2_DIST=" C2_STR(SURFD_C2_DIST) " C2_LEAN_WAVES=" C2_STR(SURFD_C2_LEAN_WAVES) " C2_LEAN_U=" C2_STR(SURFD_C2_LEAN_U) " C2_PLANE_LEAN=" C2_STR(SURFD_C2_PLANE_LEAN)
           " C2_ABLATE=" C2_STR(C2_CFG_ABLATE) " C2_DBG_POISON=" C2_STR(C2_CFG_POISON) " C2_PROBE=" C2_STR(C2_CFG_PROBE) " C2_STAMPS=" C2_STR(C2_CFG_STAMPS);
}
int conv2_build_unsafe() { return C2_UNSAFE_COUNT; }

int conv2_set_attributes() {
    const int max_lds = 160 * 1024;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<8, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<8, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<8, false, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<8, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<16, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<16, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<16, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_kernel<8, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    return SURFD_OK;
}

}  // namespace surfd
