// Micro-benchmark: time per kernel of a chain of dependent launches (direct vs hipGraph replay), for an
// empty kernel at several grid sizes / dynamic LDS sizes.  Establishes the platform floor for the
// reverse loop's ~115 launches per iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel(float *p) { extern __shared__ float s[]; if (p == nullptr) s[threadIdx.x] = 1.f; }
__global__ void touch_kernel(float *p, int n) {
    extern __shared__ float s[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
    CK(hipFuncSetAttribute((const void *)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)touch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int NK = 115, REP = 300;
    struct Cfg { int blocks; int lds; int touch; } cfgs[] = {{1, 0, 0}, {256, 0, 0}, {256, 120 * 1024, 0}, {56, 120 * 1024, 0}, {256, 120 * 1024, 1}, {448, 120 * 1024, 1}};
    for (auto c : cfgs) {
        auto launch_chain = [&](hipStream_t s) {
            for (int k = 0; k < NK; ++k) {
                if (c.touch) hipLaunchKernelGGL(touch_kernel, dim3(c.blocks), dim3(256), c.lds, s, buf, 65536);
                else hipLaunchKernelGGL(empty_kernel, dim3(c.blocks), dim3(256), c.lds, s, buf);
            }
        };
        // direct
        launch_chain(st); CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < REP; ++r) launch_chain(st);
        CK(hipStreamSynchronize(st));
        double direct = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (REP * NK);
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        launch_chain(st);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double graph = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (REP * NK);
        printf("blocks=%4d lds=%6d touch=%d : direct %.2f us/kernel, graph %.2f us/kernel\n", c.blocks, c.lds, c.touch, direct, graph);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
