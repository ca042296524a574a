// Error reporting, version, device probe and the weight packer shared by all kernels.
#include "common.h"

namespace surfd {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ void pack_kernel(PackDesc d) {
    const long total = (long)(d.Npad / 32) * (d.Kpad / 8) * 256;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int q = e & 3;
        const int lane = (e >> 2) & 63;
        const long tk = e >> 8;
        const int kg = tk % (d.Kpad / 8);
        const int tile = tk / (d.Kpad / 8);
        const int n = tile * 32 + (lane & 31);
        const int k = kg * 8 + 4 * (lane >> 5) + q;
        float v = 0.f;
        if (n < d.N && k < d.K) {
            const int outer = k / d.inner, in = k % d.inner;
            if (in < d.inner_valid) v = d.src[n * d.rs + outer * d.os + in * d.is];
        }
        d.dst[(((long)tile * d.KGtot + d.kg_off + kg) * 64 + lane) * 4 + q] = v;
    }
}

int launch_pack(const PackDesc &d, hipStream_t s) {
    const long total = (long)(d.Npad / 32) * (d.Kpad / 8) * 256;
    int blocks = (int)std::min<long>(ceil_div<long>(total, 256), 4096);
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, s, d);
    LAUNCH_CHECK();
    return SURFD_OK;
}

}  // namespace surfd

extern "C" {
const char *surfd_last_error(void) { return surfd::g_err; }
int surfd_abi_version(void) { return 1; }
int surfd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
}
