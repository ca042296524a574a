// Does a VALU write to a register that a just-issued matrix instruction reads as SrcA / SrcB reach the register file before
// the matrix instruction has fetched it?  (LLVM's gfx950 hazard recogniser pads this write-after-read case for SrcC only.)
//
// conv2_kernel's K loop (conv_f16x2.hip), as compiled in round 5, ends every k16 step with
//     ds_read_b128 v[38:41], addr offset:PLANE
//     s_waitcnt lgkmcnt(0)
//     v_mfma_f32_32x32x16_f16 acc_sm, wh, v[38:41], acc_sm        ; third product of the step, B = low plane
//     <6-8 scalar instructions: loop bookkeeping>
//     v_cndmask_b32 v38, 0, 1, s[26:27]                           ; address arithmetic of the NEXT step's load, in v38
// and the third product of a step cannot start before the first one (same accumulator, two matrix instructions earlier) has
// finished.  This test runs that chain — A = 1, high plane = 1, low plane = 2 — and writes 0 into one register of the third
// product's B operand NOPS wait states after issuing it: acc_sm must still be 48 * N.
//
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma_valu_war_test tools/ubench/mfma_valu_war_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define STR_(x) #x
#define STR(x) STR_(x)
#define CMP1(r) "v_cmp_neq_f32 vcc, s21, v" #r "\n\tv_addc_co_u32 v61, vcc, 0, v61, vcc\n\t"
#define CMP1B(r) "v_cmp_neq_f32 vcc, s22, v" #r "\n\tv_addc_co_u32 v61, vcc, 0, v61, vcc\n\t"

// REG: the VGPR that is overwritten (48..51 = A operand, 52..55 = B operand); NOPS: s_nop count between the third product and the write
// AFTER: 3 = behind the third product (what the kernel does), 2 = behind the second (B = high plane, read by products 1 and 2)
template <int REG, int NOPS, int AFTER>
__global__ __launch_bounds__(256) void war_kernel(int n_iter, float exp_sm, float exp_hh, unsigned *bad, unsigned *bad_wg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    for (int e = tid; e < 16384 / 4; e += 256) {
        reinterpret_cast<unsigned *>(lds)[e] = 0x3c003c00u;             // fp16 1.0
        reinterpret_cast<unsigned *>(lds + 16384)[e] = 0x40004000u;     // fp16 2.0
    }
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)lds + (unsigned)(tid & 63) * 16u + (unsigned)(tid >> 6) * 1024u;
    unsigned nbad;
    asm volatile(
        "v_mov_b32 v60, %[addr]\n\t"
        "s_mov_b32 s20, %[n]\n\t"
        "s_mov_b32 s21, %[esm]\n\t"
        "s_mov_b32 s22, %[ehh]\n\t"
        "v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\t"
        "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
        "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
        "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"
        "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"
        "s_nop 7\n"
        "1:\n\t"
        "v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\t"      // A restored (when REG is one of them)
        "ds_read_b128 v[52:55], v60\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mfma_f32_32x32x16_f16 v[0:15], v[48:51], v[52:55], v[0:15]\n\t"
        "v_mfma_f32_32x32x16_f16 v[16:31], v[48:51], v[52:55], v[16:31]\n\t"
        ".if %[after] == 2\n\t"
        ".rept %[nops]\n\ts_nop 0\n\t.endr\n\t"
        "v_mov_b32 v[%[reg]], 0\n\t"
        "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
        "v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\t"
        "s_nop 1\n\t"
        ".endif\n\t"
        "ds_read_b128 v[56:59], v60 offset:16384\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mfma_f32_32x32x16_f16 v[0:15], v[48:51], v[56:59], v[0:15]\n\t"
        ".if %[after] == 3\n\t"
        ".rept %[nops]\n\ts_nop 0\n\t.endr\n\t"
        ".if %[reg] >= 52\n\t"
        "v_mov_b32 v[%[reg]+4], 0\n\t"        // B of the third product lives in v[56:59]
        ".else\n\t"
        "v_mov_b32 v[%[reg]], 0\n\t"
        ".endif\n\t"
        ".endif\n\t"
        "s_sub_u32 s20, s20, 1\n\t"
        "s_cmp_lg_u32 s20, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        "v_mov_b32 v61, 0\n\t"
        CMP1(0) CMP1(1) CMP1(2) CMP1(3) CMP1(4) CMP1(5) CMP1(6) CMP1(7) CMP1(8) CMP1(9) CMP1(10) CMP1(11) CMP1(12) CMP1(13) CMP1(14) CMP1(15)
        CMP1B(16) CMP1B(17) CMP1B(18) CMP1B(19) CMP1B(20) CMP1B(21) CMP1B(22) CMP1B(23) CMP1B(24) CMP1B(25) CMP1B(26) CMP1B(27) CMP1B(28) CMP1B(29) CMP1B(30) CMP1B(31)
        "v_mov_b32 %[out], v61"
        : [out] "=v"(nbad)
        : [addr] "v"(addr), [n] "s"(n_iter), [esm] "s"(exp_sm), [ehh] "s"(exp_hh), [after] "i"(AFTER), [nops] "i"(NOPS), [reg] "i"(REG)
        : "s20", "s21", "s22", "scc", "vcc", "memory",
          "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23",
          "v24","v25","v26","v27","v28","v29","v30","v31","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61");
    if (nbad) { atomicAdd(bad, nbad); atomicAdd(bad_wg + blockIdx.x, nbad); }
}

template <int REG, int NOPS, int AFTER>
static void run_case(int cus, int n_iter, int reps, unsigned *bad, unsigned *bad_wg) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&war_kernel<REG, NOPS, AFTER>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const float esm = 48.f * (float)n_iter, ehh = 16.f * (float)n_iter;
    for (int per_cu = 1; per_cu <= 4; ++per_cu) {
        const size_t lds = (size_t)(160 * 1024 / per_cu) - 1024;
        const int wgs = cus * per_cu * 2;
        unsigned total = 0, n_bad_wg = 0;
        for (int r = 0; r < reps; ++r) {
            (void)hipMemset(bad, 0, 4); (void)hipMemset(bad_wg, 0, cus * 8 * 4);
            hipLaunchKernelGGL((war_kernel<REG, NOPS, AFTER>), dim3(wgs), dim3(256), lds, 0, n_iter, esm, ehh, bad, bad_wg);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
            unsigned h;
            (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
            total += h;
            if (h) {
                std::vector<unsigned> w(wgs);
                (void)hipMemcpy(w.data(), bad_wg, wgs * 4, hipMemcpyDeviceToHost);
                for (int i = 0; i < wgs; ++i) n_bad_wg += w[i] != 0;
            }
        }
        printf("write to v%d (%s operand) %d wait state(s) after product %d, %d workgroup(s) per CU: wrong accumulator values %u in %u workgroups of %d x %d\n",
               REG, REG >= 52 ? "B" : "A", NOPS, AFTER, per_cu, total, n_bad_wg, reps, wgs);
    }
}

int main(int argc, char **argv) {
    const int n_iter = argc > 1 ? atoi(argv[1]) : 2048;
    const int reps = argc > 2 ? atoi(argv[2]) : 4;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned *bad, *bad_wg;
    (void)hipMalloc(&bad, 4); (void)hipMalloc(&bad_wg, cus * 8 * 4);
    printf("mfma_valu_war_test: %d CUs, %d iterations per wave, %d launches per case\n", cus, n_iter, reps);
    // behind the third product (the kernel's case): every register of its B operand, then its A operand
    run_case<52, 0, 3>(cus, n_iter, reps, bad, bad_wg); run_case<53, 0, 3>(cus, n_iter, reps, bad, bad_wg);
    run_case<54, 0, 3>(cus, n_iter, reps, bad, bad_wg); run_case<55, 0, 3>(cus, n_iter, reps, bad, bad_wg);
    run_case<52, 1, 3>(cus, n_iter, reps, bad, bad_wg); run_case<52, 2, 3>(cus, n_iter, reps, bad, bad_wg);
    run_case<52, 4, 3>(cus, n_iter, reps, bad, bad_wg); run_case<52, 8, 3>(cus, n_iter, reps, bad, bad_wg);
    run_case<55, 2, 3>(cus, n_iter, reps, bad, bad_wg); run_case<55, 4, 3>(cus, n_iter, reps, bad, bad_wg); run_case<55, 8, 3>(cus, n_iter, reps, bad, bad_wg);
    run_case<48, 0, 3>(cus, n_iter, reps, bad, bad_wg); run_case<51, 0, 3>(cus, n_iter, reps, bad, bad_wg); run_case<51, 4, 3>(cus, n_iter, reps, bad, bad_wg);
    // behind the second product (B = high plane, read by the first two products)
    run_case<52, 0, 2>(cus, n_iter, reps, bad, bad_wg); run_case<55, 0, 2>(cus, n_iter, reps, bad, bad_wg);
    run_case<55, 4, 2>(cus, n_iter, reps, bad, bad_wg); run_case<55, 8, 2>(cus, n_iter, reps, bad, bad_wg);
    run_case<51, 0, 2>(cus, n_iter, reps, bad, bad_wg); run_case<51, 8, 2>(cus, n_iter, reps, bad, bad_wg);
    return 0;
}
