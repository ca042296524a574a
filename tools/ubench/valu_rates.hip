// Issue cost of the VALU instructions used by the decoder epilogue, one wave per SIMD (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/ubench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 64
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define BENCH(name, asm_indep, asm_dep)                                                                          \
    __global__ void k_##name(long long *out, float seed) {                                                       \
        float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = seed * 5.f;                            \
        float e = a + 1.f, f = a + 2.f, g = a + 3.f, h = a + 4.f;                                               \
        long long t0 = __builtin_readcyclecounter();                                                             \
        for (int i = 0; i < 16; ++i) { REP16(asm volatile(asm_indep : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) } \
        long long t1 = __builtin_readcyclecounter();                                                             \
        for (int i = 0; i < 16; ++i) { REP16(asm volatile(asm_dep : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }   \
        long long t2 = __builtin_readcyclecounter();                                                             \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }                                            \
        if (a + b + c + d + e + f + g + h == 12345.f) out[2] = 1;                                                \
    }
// each asm block = 4 instructions
BENCH(fma, "v_fma_f32 %0, %4, %5, %0\nv_fma_f32 %1, %4, %5, %1\nv_fma_f32 %2, %4, %5, %2\nv_fma_f32 %3, %4, %5, %3",
           "v_fma_f32 %0, %0, %5, %4\nv_fma_f32 %0, %0, %5, %4\nv_fma_f32 %0, %0, %5, %4\nv_fma_f32 %0, %0, %5, %4")
BENCH(med3, "v_med3_f32 %0, %0, 0, %5\nv_med3_f32 %1, %1, 0, %5\nv_med3_f32 %2, %2, 0, %5\nv_med3_f32 %3, %3, 0, %5",
            "v_med3_f32 %0, %0, 0, %5\nv_med3_f32 %0, %0, 0, %5\nv_med3_f32 %0, %0, 0, %5\nv_med3_f32 %0, %0, 0, %5")
BENCH(cvtpk, "v_cvt_pk_f16_f32 %0, %4, %5\nv_cvt_pk_f16_f32 %1, %4, %5\nv_cvt_pk_f16_f32 %2, %4, %5\nv_cvt_pk_f16_f32 %3, %4, %5",
             "v_cvt_pk_f16_f32 %0, %0, %5\nv_cvt_pk_f16_f32 %0, %0, %5\nv_cvt_pk_f16_f32 %0, %0, %5\nv_cvt_pk_f16_f32 %0, %0, %5")
BENCH(cvtrtz, "v_cvt_pkrtz_f16_f32 %0, %4, %5\nv_cvt_pkrtz_f16_f32 %1, %4, %5\nv_cvt_pkrtz_f16_f32 %2, %4, %5\nv_cvt_pkrtz_f16_f32 %3, %4, %5",
              "v_cvt_pkrtz_f16_f32 %0, %0, %5\nv_cvt_pkrtz_f16_f32 %0, %0, %5\nv_cvt_pkrtz_f16_f32 %0, %0, %5\nv_cvt_pkrtz_f16_f32 %0, %0, %5")
BENCH(cvtf32, "v_cvt_f32_f16 %0, %4\nv_cvt_f32_f16 %1, %5\nv_cvt_f32_f16 %2, %6\nv_cvt_f32_f16 %3, %7",
              "v_cvt_f32_f16 %0, %0\nv_cvt_f32_f16 %0, %0\nv_cvt_f32_f16 %0, %0\nv_cvt_f32_f16 %0, %0")
BENCH(cvtf32hi, "v_cvt_f32_f16_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",
                "v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
BENCH(sub, "v_sub_f32 %0, %4, %0\nv_sub_f32 %1, %4, %1\nv_sub_f32 %2, %4, %2\nv_sub_f32 %3, %4, %3",
           "v_sub_f32 %0, %4, %0\nv_sub_f32 %0, %4, %0\nv_sub_f32 %0, %4, %0\nv_sub_f32 %0, %4, %0")
BENCH(and_, "v_and_b32 %0, %4, %0\nv_and_b32 %1, %4, %1\nv_and_b32 %2, %4, %2\nv_and_b32 %3, %4, %3",
            "v_and_b32 %0, %4, %0\nv_and_b32 %0, %4, %0\nv_and_b32 %0, %4, %0\nv_and_b32 %0, %4, %0")
BENCH(perm, "v_perm_b32 %0, %4, %5, %6\nv_perm_b32 %1, %4, %5, %6\nv_perm_b32 %2, %4, %5, %6\nv_perm_b32 %3, %4, %5, %6",
            "v_perm_b32 %0, %0, %5, %6\nv_perm_b32 %0, %0, %5, %6\nv_perm_b32 %0, %0, %5, %6\nv_perm_b32 %0, %0, %5, %6")
BENCH(accrw, "v_accvgpr_write_b32 a0, %4\nv_accvgpr_read_b32 %0, a1\nv_accvgpr_write_b32 a2, %5\nv_accvgpr_read_b32 %1, a3",
             "v_accvgpr_write_b32 a0, %0\nv_accvgpr_read_b32 %0, a0\nv_accvgpr_write_b32 a0, %0\nv_accvgpr_read_b32 %0, a0")
#define RUN(name, per) { hipLaunchKernelGGL(k_##name, dim3(1), dim3(64), 0, 0, d, 1.5f); hipDeviceSynchronize(); long long h[3]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); \
    printf("%-10s independent %.2f cyc/instr   dependent %.2f cyc/instr\n", #name, (double)h[0] / (256.0 * per), (double)h[1] / (256.0 * per)); }
int main() {
    long long *d; hipMalloc(&d, 64);
    RUN(fma, 4) RUN(med3, 4) RUN(cvtpk, 4) RUN(cvtrtz, 4) RUN(cvtf32, 4) RUN(cvtf32hi, 4) RUN(sub, 4) RUN(and_, 4) RUN(perm, 4) RUN(accrw, 4)
    return 0;
}
