// Device helpers shared by the split-fp16 convolution kernels of the latent denoiser (conv_f16x2.hip: the four-wave latency /
// wide / lean forms; conv_lat16.hip: the sixteen-wave latency form): vector types, SiLU, kernel-argument prefetch, the in-wave
// GroupNorm reductions with their padded cross-lane reads, the LDS barrier.  Moved here unchanged in round 6.
#pragma once
#include "common.h"

namespace surfd {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16x8 __attribute__((address_space(1))) gf16x8;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

}  // namespace surfd
