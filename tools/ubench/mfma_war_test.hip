// Is a matrix instruction's SOURCE operand safe against a younger LDS load that writes the same registers?
//
// conv2_kernel's K loop (conv_f16x2.hip) compiles to
//     ds_read_b128 v[38:41], addr              ; high fp16 plane of the operand
//     s_waitcnt lgkmcnt(0)
//     v_mfma_f32_32x32x16_f16 acc_sm, wl, v[38:41], acc_sm
//     v_mfma_f32_32x32x16_f16 acc_hh, wh, v[38:41], acc_hh
//     ds_read_b128 v[38:41], addr offset:PLANE ; low plane INTO THE SAME REGISTERS, issued while the two MFMAs are in flight
//     s_waitcnt lgkmcnt(0)
//     v_mfma_f32_32x32x16_f16 acc_sm, wh, v[38:41], acc_sm
// — a write-after-read on the source of an in-flight matrix instruction that the compiler considers safe (the load's data is
// ~64+ cycles away, the MFMA reads its operands when it starts).  Round 5 saw timing-dependent wrong results ONLY in workgroups
// that share a CU with an older one (two or three waves per SIMD).  This test runs exactly that instruction sequence with
// known data — A = 1, high plane = 1, low plane = 2 — at 1 / 2 / 3 / 4 waves per SIMD: acc of the first two products must be
// 16 * N exactly.  mode 1 is the control: the second load goes to other registers.
//
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma_war_test tools/ubench/mfma_war_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PLANE_BYTES = 16384;

template <int MODE>
__global__ __launch_bounds__(256) void war_kernel(int n_iter, int stagger, unsigned *bad, unsigned *bad_wg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    // planes: [0, 16 KB) = fp16 1.0, [16 KB, 32 KB) = fp16 2.0
    for (int e = tid; e < PLANE_BYTES / 4; e += 256) {
        reinterpret_cast<unsigned *>(lds)[e] = 0x3c003c00u;
        reinterpret_cast<unsigned *>(lds + PLANE_BYTES)[e] = 0x40004000u;
    }
    __syncthreads();
    // desynchronise the workgroups of a CU (the failing launches were the ones whose workgroups start at different times)
    for (int k = 0; k < (int)((blockIdx.x * 7u) % (unsigned)(stagger + 1)); ++k) __builtin_amdgcn_s_sleep(20);
    f32x16 c0, c1, c2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; }
    f16x8 a;
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = (_Float16)1.0f;
    f16x8 b, b2;
    const unsigned addr = (unsigned)(size_t)lds + (unsigned)(tid & 63) * 16u + (unsigned)(tid >> 6) * 1024u;
    // MODE 0 / 1: three independent accumulators.  MODE 2 / 3: the kernel's dependency structure — the first and the third
    // product of a step go to the SAME accumulator (acc_sm), so the first product of a step waits for the third of the
    // previous one inside the matrix pipe while the wave goes on issuing (second product, then the load).
    // Odd modes are the controls: the second load goes to other registers.
#define WAR_LOOP(C_THIRD, B_SECOND)                                                  \
        asm volatile(                                                                 \
            "s_mov_b32 s20, %[n]\n"                                                   \
            "1:\n\t"                                                                  \
            "ds_read_b128 %[b], %[addr]\n\t"                                          \
            "s_waitcnt lgkmcnt(0)\n\t"                                                \
            "v_mfma_f32_32x32x16_f16 %[c0], %[a], %[b], %[c0]\n\t"                    \
            "v_mfma_f32_32x32x16_f16 %[c1], %[a], %[b], %[c1]\n\t"                    \
            "ds_read_b128 %[" B_SECOND "], %[addr] offset:16384\n\t"                  \
            "s_waitcnt lgkmcnt(0)\n\t"                                                \
            "v_mfma_f32_32x32x16_f16 %[" C_THIRD "], %[a], %[" B_SECOND "], %[" C_THIRD "]\n\t" \
            "s_sub_u32 s20, s20, 1\n\t"                                               \
            "s_cmp_lg_u32 s20, 0\n\t"                                                 \
            "s_cbranch_scc1 1b\n\t"                                                   \
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"                            \
            : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [b] "=&v"(b), [b2] "=&v"(b2) \
            : [a] "v"(a), [addr] "v"(addr), [n] "s"(n_iter)                           \
            : "s20", "scc", "memory")
    if constexpr (MODE == 0) WAR_LOOP("c2", "b");
    else if constexpr (MODE == 1) WAR_LOOP("c2", "b2");
    else if constexpr (MODE == 2) WAR_LOOP("c0", "b");
    else WAR_LOOP("c0", "b2");
    const bool chain = MODE >= 2;
    const float e0 = (chain ? 48.f : 16.f) * (float)n_iter, e1 = 16.f * (float)n_iter, e2 = chain ? 0.f : 32.f * (float)n_iter;
    unsigned nbad = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) nbad += (c0[r] != e0) + (c1[r] != e1) + (c2[r] != e2);
    if (nbad) {
        atomicAdd(bad, nbad);
        atomicAdd(bad_wg + blockIdx.x, nbad);
    }
}

int main(int argc, char **argv) {
    const int n_iter = argc > 1 ? atoi(argv[1]) : 4096;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned *bad, *bad_wg;
    const int max_wg = cus * 8;
    (void)hipMalloc(&bad, 4); (void)hipMalloc(&bad_wg, max_wg * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&war_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&war_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&war_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&war_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("mfma_war_test: %d CUs, %d iterations per wave, %d launches per case\n", cus, n_iter, reps);
    for (int mode = 0; mode < 4; ++mode)
        for (int per_cu = 1; per_cu <= 4; ++per_cu)
            for (int stagger : {0, 15}) {
                const size_t lds = (size_t)(160 * 1024 / per_cu) - 1024;       // exactly per_cu workgroups fit a CU
                const int wgs = cus * per_cu * 2;                               // two rounds: later workgroups join CUs that are busy
                unsigned total = 0, first_bad_wg = 0xffffffffu, n_bad_wg = 0, bad_low = 0;
                for (int r = 0; r < reps; ++r) {
                    (void)hipMemset(bad, 0, 4); (void)hipMemset(bad_wg, 0, max_wg * 4);
                    if (mode == 0) hipLaunchKernelGGL(war_kernel<0>, dim3(wgs), dim3(256), lds, 0, n_iter, stagger, bad, bad_wg);
                    else if (mode == 1) hipLaunchKernelGGL(war_kernel<1>, dim3(wgs), dim3(256), lds, 0, n_iter, stagger, bad, bad_wg);
                    else if (mode == 2) hipLaunchKernelGGL(war_kernel<2>, dim3(wgs), dim3(256), lds, 0, n_iter, stagger, bad, bad_wg);
                    else hipLaunchKernelGGL(war_kernel<3>, dim3(wgs), dim3(256), lds, 0, n_iter, stagger, bad, bad_wg);
                    hipError_t e = hipDeviceSynchronize();
                    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
                    unsigned h;
                    (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
                    total += h;
                    if (h) {
                        std::vector<unsigned> w(wgs);
                        (void)hipMemcpy(w.data(), bad_wg, wgs * 4, hipMemcpyDeviceToHost);
                        for (int i = 0; i < wgs; ++i)
                            if (w[i]) { ++n_bad_wg; if ((unsigned)i < first_bad_wg) first_bad_wg = i; if (i < cus) ++bad_low; }
                    }
                }
                printf("mode %d (%s) workgroups/CU %d stagger %2d: wrong accumulator values %u, workgroups with errors %u (first %d, among the first %d: %u)\n",
                       mode, (mode & 1) == 0 ? (mode ? "dependent accumulator chain, load into the in-flight MFMA's source" : "load into the in-flight MFMA's source") : (mode > 1 ? "dependent chain, control: other registers" : "control: other registers"), per_cu, stagger, total, n_bad_wg,
                       (int)first_bad_wg, cus, bad_low);
            }
    return 0;
}
