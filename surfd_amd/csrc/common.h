// Shared host/device helpers for libsurfd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/surfd_hip.h"

namespace surfd {

void set_error(const char *fmt, ...);

#define SURFD_FAIL(code, ...)                \
    do {                                     \
        ::surfd::set_error(__VA_ARGS__);     \
        return (code);                       \
    } while (0)

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            SURFD_FAIL(SURFD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                                  \
    } while (0)

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

static inline hipStream_t as_stream(surfd_stream s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
static inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------
// MFMA f32 32x32x2 fragment convention used everywhere in this library (wave64):
//   D[32x32] += A[32x2] * B[2x32]
//   A operand: lane l holds A[i = l & 31][k = l >> 5]
//   B operand: lane l holds B[k = l >> 5][j = l & 31]
//   D: lane l holds column j = l & 31, rows (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r in [0,16)
// Weights are stored pre-packed ("fragment-major") so that ONE fully coalesced 16-byte load
// per lane yields the operand of FOUR consecutive MFMAs:
//   packed[((tile * KG + kg) * 64 + lane) * 4 + q] = W[tile*32 + (lane & 31)][kg*8 + 4*(lane >> 5) + q]
// i.e. MFMA q of k-group kg contracts k-slots {kg*8 + q, kg*8 + 4 + q}.  The other operand
// must be fetched with the same slot map: lane reads 4 consecutive k at kg*8 + 4*(lane >> 5).
// ------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// packs a logical [N x K] matrix given by an element functor into fragment-major layout
// (N padded to 32, K padded to 8; padding is zero-filled)
struct PackDesc {
    const float *src;
    float *dst;
    int N, K;            // logical sizes
    int Npad, Kpad;      // multiples of 32 / 8
    // logical W[n][k] = src[n*rs + (k / inner)*os + (k % inner)*is] with k split as (outer, inner)
    long rs;
    int inner;           // size of the inner (fast) logical k index
    long os, is;         // strides of outer / inner k index in src
    int inner_valid;     // inner indices >= inner_valid are zero padding
    int KGtot, kg_off;   // k-groups per tile in dst, and where this segment starts
};
int launch_pack(const PackDesc &d, hipStream_t s);

// Optional HIP-event instrumentation (surfd_profile_*): brackets a kernel class on the stream it
// is launched on.  kind: 0 decoder forward, 1 decoder forward+reverse, 2 whole reverse loop.
enum { PROF_DEC_FWD = 0, PROF_DEC_GRAD = 1, PROF_LOOP = 2, PROF_KINDS = 3 };
bool prof_enabled();
hipEvent_t prof_begin(int kind, hipStream_t st);                  // null when profiling is off
void prof_end(int kind, hipEvent_t begin, hipStream_t st);

// surfd_build_config: the experiment macros each kernel file was compiled with, and how many select a variant recorded as unsafe
const char *conv2_build_config();      // conv_f16x2.hip
int conv2_build_unsafe();
const char *decoder_build_config();    // decoder.hip
int decoder_build_unsafe();

}  // namespace surfd
