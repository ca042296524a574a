// Point sources / sinks shared by the decoder kernels and the grid-filler kernels.
//
// A "point" is either an explicit xyz triple (udf_func on arbitrary points) or a voxel of
// the N^3 grid addressed by its flat index idx = i*N*N + j*N + k whose coordinate is
//     x = float(i) * voxel + origin        (two separately rounded fp32 ops)
// exactly as GridFiller.__init__ builds its `samples` table (reference
// meshudf/meshudf.py:53-75: column 0 <- i, column 1 <- j, column 2 <- k).
#pragma once
#include "common.h"

namespace surfd {

enum PtMode {
    PT_XYZ = 0,       // xyz[n,3]
    PT_EMB = 1,       // pre-encoded emb[n,input_dim] (CbnDecoder.forward contract)
    PT_LIST = 2,      // voxel indices list[count]
    PT_CHILDREN = 3,  // 7 new lattice children (stride s) of every parent corner in list[count]
    PT_DENSE = 4,     // every voxel 0..N^3-1
    PT_LATTICE = 5,   // every point of the stride-s lattice ((N/s)^3 points)
    PT_CHILDREN8 = 6  // all 8 children incl. the parent itself (classification pass)
};

struct PtIO {
    int mode;
    const float *xyz;        // PT_XYZ / PT_EMB source
    const int *list;         // PT_LIST / PT_CHILDREN*
    const int *count_dev;    // device-side element count of `list` (nullable)
    long n;                  // host-side point count when count_dev == nullptr
    int N, s;                // grid resolution, lattice stride
    int log2N1;              // log2(N) + 1 when N is a power of two (index arithmetic by shifts), 0 otherwise / unknown
    float voxel, origin;
    int emb_dim;
    // sinks (all nullable)
    float *out_udf, *out_logit, *out_ngrad;   // dense, indexed by point number
    float *out_dlogit;                        // dense [n,3]: raw d logit / d xyz (autograd hook)
    float *grid_udf, *grid_grads;             // scattered by voxel index
    int *grad_list;                           // voxels with udf < grad_thr are appended here
    int *grad_count;
    float grad_thr;
    // grid-shard mode (one shape evaluated by several ranks): this launch walks only the 64-point tiles t = shard_i (mod shard_n)
    // of the source (shard_n <= 1: all of them); cap > 0 bounds the point count (the exchange buffers of the ranks have a fixed
    // capacity: the count itself lives on the device)
    int shard_n, shard_i;
    long cap;
};

__device__ __forceinline__ long pt_count(const PtIO &io) {
    long n = io.n;
    if (io.count_dev != nullptr) {
        const long c = *io.count_dev;
        n = io.mode == PT_CHILDREN ? 7 * c : (io.mode == PT_CHILDREN8 ? 8 * c : c);
    }
    return io.cap > 0 ? min(n, io.cap) : n;
}

// voxel index of point e (grid modes only).  Point numbers fit 32 bits (at most 8 children of at most N^3 / 8 cells, N <= 1024):
// the divisions by 7 / by the lattice width are 32-bit (the 64-bit forms were ~100 instructions each on the one wave that fetches a
// tile's points while the other seven wait)
__device__ __forceinline__ int pt_voxel(const PtIO &io, long e) {
    const unsigned ee = (unsigned)e;
    switch (io.mode) {
        case PT_LIST: return io.list[e];
        case PT_CHILDREN: {
            const unsigned pe = ee / 7u;
            const int parent = io.list[pe];
            const int c = (int)(ee - 7u * pe) + 1;
            return parent + (((c >> 2) & 1) * io.N * io.N + ((c >> 1) & 1) * io.N + (c & 1)) * io.s;
        }
        case PT_CHILDREN8: {
            const int parent = io.list[ee >> 3];
            const int c = (int)(ee & 7u);
            return parent + (((c >> 2) & 1) * io.N * io.N + ((c >> 1) & 1) * io.N + (c & 1)) * io.s;
        }
        case PT_LATTICE: {
            const unsigned n = (unsigned)(io.N / io.s);
            const unsigned q = ee / n, K = ee - q * n, I = q / n, J = q - I * n;
            return (int)((I * io.N * io.N + J * io.N + K) * io.s);
        }
        default: return (int)e;   // PT_DENSE
    }
}

__device__ __forceinline__ void voxel_xyz(const PtIO &io, int idx, float &x, float &y, float &z) {
    int i, j, k;
    if (io.log2N1) {          // uniform: every grid the drivers use is a power of two
        const int l = io.log2N1 - 1;
        k = idx & (io.N - 1); j = (idx >> l) & (io.N - 1); i = idx >> (2 * l);
    } else {
        k = idx % io.N; j = (idx / io.N) % io.N; i = idx / (io.N * io.N);
    }
    x = __fadd_rn(__fmul_rn((float)i, io.voxel), io.origin);
    y = __fadd_rn(__fmul_rn((float)j, io.voxel), io.origin);
    z = __fadd_rn(__fmul_rn((float)k, io.voxel), io.origin);
}

// Appends to a device list with ONE atomic per wave (a per-lane atomicAdd on one counter
// serialises at ~11 ns per lane: 0.37 ms for the 32^3 lattice).  Must be reached by every lane
// of the wave (pass pred = false for lanes with nothing to append).  Returns the slot or -1.
__device__ __forceinline__ int wave_append_slot(int *counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return -1;
    const int lane = (int)(threadIdx.x & 63);
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader);
    return pred ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// Several point sources served by ONE launch of the persistent decoder kernel: every workgroup walks its tiles of source 0,
// then of source 1, ... (each with the conditional-BN tables of its own latent `sample[i]`).  The grids of a batch of
// shapes are at the same refinement level at the same time, so a level of all of them is one launch: no launch boundary
// (workgroups that run out of tiles of one shape start on the next shape at once) and an eighth of the launches whose
// persistent workgroups have to find free CUs next to the reverse loops.
constexpr int PT_BATCH_MAX = 8;
struct PtBatch {
    int n;
    int sample[PT_BATCH_MAX];
    PtIO io[PT_BATCH_MAX];
};

// Enqueue the fused decoder over a point source / a batch of them (defined in decoder.hip).
int decoder_launch(surfd_decoder *d, int sample, PtIO io, bool grad, long ntiles_hint, hipStream_t st);
int decoder_launch_batch(surfd_decoder *d, const PtBatch &b, bool grad, long ntiles_hint, hipStream_t st);

}  // namespace surfd
