// A cross-lane reduction right behind the VALU instruction that produced its input, next to waves that keep the SIMD's matrix
// pipe and register-file write ports busy: does every lane still get the right sum?
//
// Round 6 traced conv2_kernel's timing-dependent wrong results (profiles/r06_conv2_instability.md) to ONE quantity: the
// GroupNorm 1/sigma of one or two groups of one sample, short by about one "between rows" term — i.e. one lane of the 8-lane
// all-reduce of the statistics (conv_f16x2.hip: `sm += __shfl_xor(sm, 4); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 1);`,
// compiled to v_add_f32 -> ds_bpermute_b32 of the SAME register -> s_waitcnt -> v_add_f32 ...; in the in-wave form v_add_f32 ->
// s_nop 1 -> DPP) worked with a slightly different group mean than its neighbours.  Only workgroups that share a CU with
// another one in a different phase of its life are hit.  This test pairs "victim" workgroups running exactly that reduction on
// exact small integers with "aggressor" workgroups running the kernel's K loop (ds_read_b128 + three dependent matrix
// instructions) on the same CUs, and counts lanes whose sum is wrong.
//
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/xlane_raw_test tools/ubench/xlane_raw_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CTRL>
__device__ __forceinline__ float dpp(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true)); }

// MODE 0: ds_bpermute chain (HIP's __shfl_xor), MODE 1: DPP chain (quad_perm, quad_perm, row_half_mirror),
// PAD: s_nop wait states forced between every add and the cross-lane read of its result (the candidate fix)
template <int MODE, int PAD>
__global__ __launch_bounds__(256) void xlane_kernel(int n_iter, int victims_every, unsigned *bad, unsigned *bad_wg, unsigned *example) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 32768 / 4; e += 256) reinterpret_cast<unsigned *>(lds)[e] = e < 4096 ? 0x3c003c00u : 0x40004000u;
    float *ex = reinterpret_cast<float *>(lds + 32768);          // the victims' exchange array (as the kernel's ex_mean / ex_m2)
    __syncthreads();
    const bool victim = (blockIdx.x % victims_every) == 0;
    if (!victim) {
        // the kernel's K loop: B operand from LDS, three matrix instructions, first and third on the same accumulator
        f32x16 c0, c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        f16x8 a, b;
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = (_Float16)1.0f;
        const unsigned addr = (unsigned)(size_t)lds + (unsigned)lane * 16u + (unsigned)(tid >> 6) * 1024u;
        asm volatile(
            "s_mov_b32 s20, %[n]\n"
            "1:\n\t"
            "ds_read_b128 %[b], %[addr]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_f16 %[c0], %[a], %[b], %[c0]\n\t"
            "v_mfma_f32_32x32x16_f16 %[c1], %[a], %[b], %[c1]\n\t"
            "ds_read_b128 %[b], %[addr] offset:16384\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_f16 %[c0], %[a], %[b], %[c0]\n\t"
            "s_sub_u32 s20, s20, 1\n\t"
            "s_cmp_lg_u32 s20, 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
            : [c0] "+v"(c0), [c1] "+v"(c1), [b] "=&v"(b)
            : [a] "v"(a), [addr] "v"(addr), [n] "s"(n_iter * 4)
            : "s20", "scc", "memory");
        if (c1[0] != 16.f * (float)(n_iter * 4)) atomicAdd(bad + 1, 1u);          // keeps the loop alive; also a check of its own
        return;
    }
    unsigned nbad = 0;
    const int grp = lane & ~7;
    for (int it = 0; it < n_iter; ++it) {
        // leaf: a value that exists only since the previous VALU instruction (as the kernel's row terms: loaded from LDS, combined)
        ex[tid] = (float)(((tid * 7 + it * 13) & 1023));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float x = ex[tid ^ 64] * 0.f + ex[tid];          // one LDS round trip, then a VALU op in front of the reduction
        float sm = x + (float)(it & 3);
        if constexpr (MODE == 0) {
            if constexpr (PAD) asm volatile(".rept %1\n\ts_nop 0\n\t.endr" : "+v"(sm) : "i"(PAD));
            sm += __shfl_xor(sm, 4);
            if constexpr (PAD) asm volatile(".rept %1\n\ts_nop 0\n\t.endr" : "+v"(sm) : "i"(PAD));
            sm += __shfl_xor(sm, 2);
            if constexpr (PAD) asm volatile(".rept %1\n\ts_nop 0\n\t.endr" : "+v"(sm) : "i"(PAD));
            sm += __shfl_xor(sm, 1);
        } else {
            if constexpr (PAD) asm volatile(".rept %1\n\ts_nop 0\n\t.endr" : "+v"(sm) : "i"(PAD));
            sm += dpp<0xB1>(sm);
            if constexpr (PAD) asm volatile(".rept %1\n\ts_nop 0\n\t.endr" : "+v"(sm) : "i"(PAD));
            sm += dpp<0x4E>(sm);
            if constexpr (PAD) asm volatile(".rept %1\n\ts_nop 0\n\t.endr" : "+v"(sm) : "i"(PAD));
            sm += dpp<0x141>(sm);
        }
        float want = 0.f;
        for (int k = 0; k < 8; ++k) want += (float)((((tid & ~63) + grp + k) * 7 + it * 13) & 1023) + (float)(it & 3);
        if (sm != want) {
            ++nbad;
            if (atomicAdd(example, 1u) == 0u) { example[1] = blockIdx.x; example[2] = it; example[3] = tid; example[4] = __float_as_uint(want); example[5] = __float_as_uint(sm); }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (nbad) { atomicAdd(bad, nbad); atomicAdd(bad_wg + blockIdx.x, nbad); }
}

template <int MODE, int PAD>
static void run_case(int cus, int n_iter, int reps, unsigned *bad, unsigned *bad_wg, unsigned *example) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&xlane_kernel<MODE, PAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int per_cu = 1; per_cu <= 4; ++per_cu)
        for (int every : {1, 2, 3}) {              // 1: victims only (nobody keeps the matrix pipe busy); 2 / 3: every second / third workgroup is a victim
            if (per_cu == 1 && every > 1) continue;
            const size_t lds = (size_t)(160 * 1024 / per_cu) - 1024;
            const int wgs = cus * per_cu * 2;
            unsigned total = 0, aggr_bad = 0;
            (void)hipMemset(example, 0, 32);
            for (int r = 0; r < reps; ++r) {
                (void)hipMemset(bad, 0, 8); (void)hipMemset(bad_wg, 0, cus * 8 * 4);
                hipLaunchKernelGGL((xlane_kernel<MODE, PAD>), dim3(wgs), dim3(256), lds, 0, n_iter, every, bad, bad_wg, example);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
                unsigned h[2];
                (void)hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
                total += h[0]; aggr_bad += h[1];
            }
            printf("%s, %d wait state(s) of padding, %d workgroup(s) per CU, %s: wrong sums %u (aggressor self-check failures %u)\n",
                   MODE == 0 ? "ds_bpermute chain" : "DPP chain", PAD, per_cu,
                   every == 1 ? "victims only" : (every == 2 ? "every 2nd workgroup a victim" : "every 3rd workgroup a victim"), total, aggr_bad);
            if (total) {
                unsigned ex[8];
                (void)hipMemcpy(ex, example, 32, hipMemcpyDeviceToHost);
                float w, g;
                memcpy(&w, &ex[4], 4); memcpy(&g, &ex[5], 4);
                printf("      first: workgroup %u iteration %u thread %u: expected %.1f, got %.1f\n", ex[1], ex[2], ex[3], w, g);
            }
        }
}


int main(int argc, char **argv) {
    const int n_iter = argc > 1 ? atoi(argv[1]) : 20000;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned *bad, *bad_wg, *example;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&bad_wg, cus * 8 * 4); (void)hipMalloc(&example, 32);
    printf("xlane_raw_test: %d CUs, %d reductions per victim wave, %d launches per case\n", cus, n_iter, reps);
    run_case<0, 0>(cus, n_iter, reps, bad, bad_wg, example);
    run_case<1, 0>(cus, n_iter, reps, bad, bad_wg, example);
    run_case<0, 4>(cus, n_iter, reps, bad, bad_wg, example);
    run_case<1, 4>(cus, n_iter, reps, bad, bad_wg, example);
    return 0;
}
