// Error reporting, version, device probe and the weight packer shared by all kernels.
#include "common.h"
#include <mutex>
#include <string>

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

static bool g_prof = false;
struct ProfPair { hipEvent_t a, b; };
static std::vector<ProfPair> g_prof_pairs[PROF_KINDS];
static std::mutex g_prof_mu;          // loops of different chains and the grid stream are driven by different host threads

bool prof_enabled() { return g_prof; }
hipEvent_t prof_begin(int kind, hipStream_t st) {
    (void)kind;
    if (!g_prof) return nullptr;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, st);
    return e;
}
void prof_end(int kind, hipEvent_t begin, hipStream_t st) {
    if (!begin) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipEventDestroy(begin); return; }
    (void)hipEventRecord(e, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_pairs[kind].push_back({begin, e});
}

}  // namespace surfd

extern "C" {
int surfd_profile_enable(int on) {
    surfd::g_prof = on != 0;
    return SURFD_OK;
}
// host-sync: sums and clears the recorded event pairs of `kind`
int surfd_profile_read(int kind, int64_t *launches, double *total_ms) {
    using namespace surfd;
    if (kind < 0 || kind >= PROF_KINDS || !launches || !total_ms) SURFD_FAIL(SURFD_ERR_ARG, "surfd_profile_read: bad argument");
    double tot = 0.0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &p : g_prof_pairs[kind]) {
        HIP_TRY(hipEventSynchronize(p.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        tot += ms;
        (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
    }
    *launches = (int64_t)g_prof_pairs[kind].size();
    *total_ms = tot;
    g_prof_pairs[kind].clear();
    return SURFD_OK;
}
const char *surfd_last_error(void) { return surfd::g_err; }
// The compile-time configuration of this library: every experiment macro of the kernel sources with the value it was built with
// ("name=value" pairs) and `unsafe_variants=N`, the number of them that select a variant recorded as wrong, not bit-stable or a
// developer aid (0 for the product build; such variants only compile with -DSURFD_ALLOW_UNSAFE_VARIANTS).
const char *surfd_build_config(void) {
    static const std::string cfg = std::string("abi=1 ") + surfd::conv2_build_config() + " " + surfd::decoder_build_config() +
                                   " unsafe_variants=" + std::to_string(surfd::conv2_build_unsafe() + surfd::decoder_build_unsafe());
    return cfg.c_str();
}
int surfd_abi_version(void) { return 1; }
int surfd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
}
