// Standalone check of the slot-sum butterfly used by the GroupNorm staging of conv_f16x2.hip (c2_slot_sum): every lane of a
// slot of 2^log2P lanes must end with the bits of the same pairwise sum, run after run, under full occupancy.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/slot_sum_test tools/ubench/slot_sum_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <int CTRL>
__device__ __forceinline__ float c2_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
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
    if (log2P > 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += c2_dpp<0xB1>(x[i]);
    }
    if (log2P > 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += c2_dpp<0x4E>(x[i]);
    }
    if (log2P > 2) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += c2_dpp<0x141>(x[i]);
    }
    if (log2P > 3) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += c2_dpp<0x140>(x[i]);
    }
    if (log2P > 4) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = c2_swap_sum<16>(x[i]);
    }
    if (log2P > 5) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = c2_swap_sum<32>(x[i]);
    }
}

// the first form of the 32- and 64-lane steps (ds_bpermute through __shfl_xor), kept to test it next to LDS traffic
template <int N>
__device__ __forceinline__ void slot_sum_bperm(float (&x)[N], int log2P) {
    if (log2P > 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += c2_dpp<0xB1>(x[i]);
    }
    if (log2P > 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += c2_dpp<0x4E>(x[i]);
    }
    if (log2P > 2) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += c2_dpp<0x141>(x[i]);
    }
    if (log2P > 3) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += c2_dpp<0x140>(x[i]);
    }
    if (log2P > 4) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += __shfl_xor(x[i], 16);
    }
    if (log2P > 5) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] += __shfl_xor(x[i], 32);
    }
}

// BPERM: the ds_bpermute form; LDS_NOISE: waves 1-3 of every workgroup stream ds_write_b16 / ds_read_b128 through 40 KB of LDS
// while wave 0 runs the butterflies (what the staging of the other workgroups of a CU does to a wave of conv2_kernel)
template <int N, bool BPERM, bool LDS_NOISE>
__global__ __launch_bounds__(256, 3) void k2(const float *in, float *out, int log2P, int gs, int rounds, float *sink) {
    __shared__ _Float16 noise[20000];
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (LDS_NOISE && wave > 0) {
        float s = 0.f;
        for (int r = 0; r < rounds * 24; ++r) {
            for (int q = 0; q < 8; ++q) noise[((r * 131 + q * 2500) % 19000) + threadIdx.x] = (_Float16)(float)(r + q);
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const h8 v = *reinterpret_cast<const h8 *>(noise + ((r * 57 + threadIdx.x * 8) % 19000 & ~7));
            s += (float)v[0] + (float)v[7];
        }
        if (s == 12345.678f) sink[tid] = s;
        return;
    }
    const bool ok = (lane & ((1 << log2P) - 1)) < gs;
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float a[N];
#pragma unroll
        for (int i = 0; i < N; ++i) a[i] = ok ? in[((size_t)r * N + i) * gridDim.x * 256 + tid] : 0.f;
        if (BPERM) slot_sum_bperm<N>(a, log2P); else c2_slot_sum<N>(a, log2P);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] += a[i] * (1.f / 64.f);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[(size_t)i * gridDim.x * 256 + tid] = acc[i];
}

template <int N>
__global__ __launch_bounds__(256, 3) void k(const float *in, float *out, int log2P, int gs, int rounds) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool ok = (lane & ((1 << log2P) - 1)) < gs;
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float a[N];
#pragma unroll
        for (int i = 0; i < N; ++i) a[i] = ok ? in[((size_t)r * N + i) * gridDim.x * 256 + tid] : 0.f;
        c2_slot_sum<N>(a, log2P);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] += a[i] * (1.f / 64.f);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[(size_t)i * gridDim.x * 256 + tid] = acc[i];
}

int main() {
    const int blocks = 768 * 4, N = 8, rounds = 16;
    const size_t n = (size_t)blocks * 256;
    std::vector<float> h(n * N * rounds);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 4.f - 2.f;
    float *din, *dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, n * N * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ref(n * N), got(n * N);
    int bad_total = 0;
    for (int lp = 0; lp <= 6; ++lp) {
        const int P = 1 << lp, gs = P == 1 ? 1 : (P * 7) / 8;      // 1, 1(2->1?), ...: slots with unused lanes like 7 of 8, 14 of 16, 28 of 32, 56 of 64
        const int gsz = gs < 1 ? 1 : gs;
        // reference: pairwise butterfly order = the kernel's (xor 1, xor 2, mirror 8, mirror 16, xor 16, xor 32) on the host
        for (size_t t = 0; t < n; t += 64)
            for (int i = 0; i < N; ++i) {
                float accl[64] = {0};
                for (int r = 0; r < rounds; ++r) {
                    float a[64], b[64];
                    for (int l = 0; l < 64; ++l) a[l] = ((l & (P - 1)) < gsz) ? h[((size_t)r * N + i) * n + t + l] : 0.f;
                    auto step = [&](auto src) { for (int l = 0; l < 64; ++l) b[l] = a[l] + a[src(l)]; memcpy(a, b, sizeof(a)); };
                    if (lp > 0) step([](int l) { return l ^ 1; });
                    if (lp > 1) step([](int l) { return l ^ 2; });
                    if (lp > 2) step([](int l) { return (l & ~7) | (7 - (l & 7)); });
                    if (lp > 3) step([](int l) { return (l & ~15) | (15 - (l & 15)); });
                    if (lp > 4) step([](int l) { return l ^ 16; });
                    if (lp > 5) step([](int l) { return l ^ 32; });
                    for (int l = 0; l < 64; ++l) accl[l] += a[l] * (1.f / 64.f);
                }
                for (int l = 0; l < 64; ++l) ref[(size_t)i * n + t + l] = accl[l];
            }
        int bad = 0, differ_runs = 0;
        std::vector<float> first;
        for (int it = 0; it < 20; ++it) {
            hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, din, dout, lp, gsz, rounds);
            hipMemcpy(got.data(), dout, n * N * 4, hipMemcpyDeviceToHost);
            if (it == 0) {
                first = got;
                for (size_t e = 0; e < n * N; ++e) if (memcmp(&got[e], &ref[e], 4)) { if (bad < 5) printf("  lp %d: e %zu got %.9g ref %.9g\n", lp, e, got[e], ref[e]); ++bad; }
            } else if (memcmp(first.data(), got.data(), n * N * 4)) ++differ_runs;
        }
        printf("log2P %d (gs %d): %d of %zu values differ from the host butterfly; %d of 19 repeat runs differ from the first\n", lp, gsz, bad, n * N, differ_runs);
        bad_total += bad + differ_runs;
    }
    // the same butterflies on wave 0 of every workgroup next to LDS traffic of waves 1-3: compare run to run
    float *sink; hipMalloc(&sink, n * 4);
    for (int form = 0; form < 2; ++form)
        for (int lp = 5; lp <= 6; ++lp) {
            const int gsz = ((1 << lp) * 7) / 8;
            std::vector<float> first;
            int differ_runs = 0;
            for (int it = 0; it < 40; ++it) {
                hipMemset(dout, 0, n * N * 4);
                if (form == 0) hipLaunchKernelGGL((k2<8, true, true>), dim3(blocks), dim3(256), 0, 0, din, dout, lp, gsz, rounds, sink);
                else hipLaunchKernelGGL((k2<8, false, true>), dim3(blocks), dim3(256), 0, 0, din, dout, lp, gsz, rounds, sink);
                hipMemcpy(got.data(), dout, n * N * 4, hipMemcpyDeviceToHost);
                if (it == 0) first = got;
                else if (memcmp(first.data(), got.data(), n * N * 4)) ++differ_runs;
            }
            printf("%s next to LDS traffic, log2P %d: %d of 39 repeat runs differ from the first\n", form == 0 ? "ds_bpermute form" : "permlane-swap form", lp, differ_runs);
            if (form == 1) bad_total += differ_runs;
        }
    printf(bad_total ? "FAILED\n" : "OK\n");
    return bad_total ? 1 : 0;
}
