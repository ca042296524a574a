// Does the staging pattern of conv2_kernel (conv_f16x2.hip) hold at two / three workgroups per CU with > 64 KB of LDS each?
// Every iteration: thread = channel writes one fp16 per (position, plane) with ds_write_b16 (two planes PLANE halfs apart, the
// low one addressed through an immediate offset), `s_waitcnt lgkmcnt(0)` + s_barrier, every wave reads the slab back with
// ds_read_b128 (the matrix instructions' operand fetch) and compares with what must be there, barrier.  Values depend on
// (iteration, workgroup, position, channel): a stale, foreign or half-written word shows.  Round 6, hunting the
// "second workgroup on a CU" instability: this isolates LDS + barrier from the rest of the kernel.
//
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/lds_stage_test tools/ubench/lds_stage_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void lds_bar() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned short val16(int it, int wg, int p, int c, int plane) {
    unsigned h = (unsigned)it * 2654435761u ^ (unsigned)wg * 40503u ^ (unsigned)p * 9176u ^ (unsigned)c * 31u ^ (unsigned)plane * 0x5bd1u;
    h ^= h >> 13;
    return (unsigned short)(h & 0x3fffu);          // a small positive fp16 bit pattern (no NaN / inf)
}

template <int PLANE>      // halfs between the planes
__global__ __launch_bounds__(256) void stage_kernel(int n_iter, int npos, int cs, int stagger, unsigned *bad, unsigned *bad_wg) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    unsigned short *slab = reinterpret_cast<unsigned short *>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < (int)((blockIdx.x * 7u) % (unsigned)(stagger + 1)); ++k) __builtin_amdgcn_s_sleep(20);
    unsigned nbad = 0;
    for (int it = 0; it < n_iter; ++it) {
        // the exchange arrays of the aliased GroupNorm statistics: fp32 words over the start of the slab, read back, barrier
        float *ex = reinterpret_cast<float *>(lds_raw);
        ex[tid] = (float)(it + tid); ex[256 + tid] = (float)(it - tid);
        lds_bar();
        const float e0 = ex[(tid + 64) & 255], e1 = ex[256 + ((tid + 128) & 255)];
        nbad += (e0 != (float)(it + ((tid + 64) & 255))) + (e1 != (float)(it - ((tid + 128) & 255)));
        lds_bar();
        if (tid < cs - 8)
            for (int p = 0; p < npos; ++p) {
                unsigned short *d = slab + p * cs + tid;
                d[0] = val16(it, blockIdx.x, p, tid, 0);
                d[PLANE] = val16(it, blockIdx.x, p, tid, 1);
            }
        if (tid >= cs - 8 && tid < cs)          // padded channels: zeros
            for (int p = 0; p < npos; ++p) { slab[p * cs + tid] = 0; slab[p * cs + tid + PLANE] = 0; }
        lds_bar();
        // every wave reads every 16-channel group of the positions lane & 31 (as the k16 steps of the matrix loop do)
        for (int kk = 0; kk + 16 <= cs - 8; kk += 16) {
            const int p = (lane & 31) % npos, c0 = kk + 8 * (lane >> 5);
            const f16x8 bh = *reinterpret_cast<const f16x8 *>(slab + p * cs + c0);
            const f16x8 bl = *reinterpret_cast<const f16x8 *>(slab + p * cs + c0 + PLANE);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                nbad += __builtin_bit_cast(unsigned short, bh[e]) != val16(it, blockIdx.x, p, c0 + e, 0);
                nbad += __builtin_bit_cast(unsigned short, bl[e]) != val16(it, blockIdx.x, p, c0 + e, 1);
            }
        }
        (void)wave;
        lds_bar();
    }
    if (nbad) { atomicAdd(bad, nbad); atomicAdd(bad_wg + blockIdx.x, nbad); }
}

int main(int argc, char **argv) {
    const int n_iter = argc > 1 ? atoi(argv[1]) : 200;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned *bad, *bad_wg;
    const int max_wg = cus * 8;
    (void)hipMalloc(&bad, 4); (void)hipMalloc(&bad_wg, max_wg * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&stage_kernel<18432>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&stage_kernel<11264>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("lds_stage_test: %d CUs, %d iterations, %d launches per case\n", cus, n_iter, reps);
    struct Case { const char *name; int plane, npos, cs, lds, per_cu; } cases[] = {
        {"64-position rows, two workgroups per CU (74 240 B)", 18432, 64, 232, 74240, 2},
        {"lean form, three workgroups per CU (45 568 B)", 11264, 48, 232, 45568, 3},
        {"64-position rows, one workgroup per CU (94 240 B)", 18432, 64, 232, 94240, 1},
    };
    for (const Case &c : cases)
        for (int stagger : {0, 15}) {
            const int wgs = cus * c.per_cu * 2;
            unsigned total = 0, n_bad_wg = 0;
            for (int r = 0; r < reps; ++r) {
                (void)hipMemset(bad, 0, 4); (void)hipMemset(bad_wg, 0, max_wg * 4);
                if (c.plane == 18432) hipLaunchKernelGGL(stage_kernel<18432>, dim3(wgs), dim3(256), c.lds, 0, n_iter, c.npos, c.cs, stagger, bad, bad_wg);
                else hipLaunchKernelGGL(stage_kernel<11264>, dim3(wgs), dim3(256), c.lds, 0, n_iter, c.npos, c.cs, stagger, bad, bad_wg);
                hipError_t e = hipDeviceSynchronize();
                if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
                unsigned h;
                (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
                total += h;
                if (h) {
                    std::vector<unsigned> w(wgs);
                    (void)hipMemcpy(w.data(), bad_wg, wgs * 4, hipMemcpyDeviceToHost);
                    for (int i = 0; i < wgs; ++i) n_bad_wg += w[i] != 0;
                }
            }
            printf("%-55s stagger %2d: wrong values %u in %u workgroups\n", c.name, stagger, total, n_bad_wg);
        }
    return 0;
}
