// Coarse-to-fine UDF grid filling on the device (HBM-bound index/byte work; the decoder
// evaluations it schedules are the MFMA-bound part, decoder.hip).
//
// Replaces GridFiller.__init__/fill_grid and get_udf_and_grads (reference
// meshudf/meshudf.py:36-206, 254-304).  The reference materialises, per sample, a
// samples[N^3,7] table, N^3 boolean masks per level and int64 [N^3/b, b] block-index tensors
// (about 1 GiB per level at 512^3).  Here every level is a compacted list of *active block
// corners* (int32 voxel indices); everything else is index arithmetic:
//
//   level l (lattice stride s_l = N / n_l), parents[l] = corners of the level l-1 blocks whose
//   corner value was "close" (|udf| < 1.5*1.7*2/n_{l-1}):
//     evaluate  : the 7 not-yet-evaluated children p + {0,1}^3 * s_l  (level 0: all 32^3)
//     classify  : all 8 children; close -> parents[l+1]; far -> block [c, c+s_l)^3 := udf(c)
//   gradients   : every evaluated voxel with udf < 2.5*2/N is appended to grad_list when its
//                 value is written (pruned voxels hold values >= the refine threshold, which
//                 always exceeds the gradient threshold, unless the field is negative — the
//                 fill kernel covers that case too).
// No host round trip is needed between levels: list lengths stay in device counters and the
// consuming kernels size their loops from them.
#include "common.h"
#include "points.h"
#include <string.h>
#include <algorithm>
#include <hipcub/hipcub.hpp>

namespace surfd {

constexpr int CTR_PARENT = 0;                       // [0..7]   parents[l] length
constexpr int CTR_FAR = SURFD_GRID_MAX_LEVELS;      // [8..15]  far blocks found at level l
constexpr int CTR_GRAD = 2 * SURFD_GRID_MAX_LEVELS; // [16]     gradient points
constexpr int CTR_TOTAL = CTR_GRAD + 1;

// Appends are aggregated per WORKGROUP and per four elements a thread: one pair of atomics reserves the slots of 1024
// elements (one atomic per wave serialised at ~11 ns each — 150 000 waves x 2 lists = 3.3 of the 4.0 ms this kernel took
// on the largest level; the decoder launches of the next level wait for it).
__global__ __launch_bounds__(256) void classify_kernel(PtIO io, const float *udf, float thr, int *close_list, int *close_count,
                                                       int *far_list, int *far_count) {
    constexpr int ITEMS = 4;
    __shared__ int wcnt[2][4], wbase[2][4];
    const long n = pt_count(io);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long e0 = blockIdx.x * (long)(blockDim.x * ITEMS); e0 < n; e0 += (long)gridDim.x * blockDim.x * ITEMS) {
        int idx[ITEMS];
        bool cl[ITEMS], fr[ITEMS];
        int ncl = 0, nfr = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const long e = e0 + (long)i * blockDim.x + threadIdx.x;
            idx[i] = 0; cl[i] = fr[i] = false;
            if (e < n) {
                idx[i] = pt_voxel(io, e);
                cl[i] = fabsf(udf[idx[i]]) < thr;
                fr[i] = !cl[i];
            }
            ncl += cl[i]; nfr += fr[i];
        }
        // exclusive prefix of the per-lane counts inside the wave, wave totals through LDS, one reservation per list
        int pcl = ncl, pfr = nfr;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int a = __shfl_up(pcl, off), b = __shfl_up(pfr, off);
            if (lane >= off) { pcl += a; pfr += b; }
        }
        if (lane == 63) { wcnt[0][wave] = pcl; wcnt[1][wave] = pfr; }
        __syncthreads();
        if (threadIdx.x < 2) {
            const int l = threadIdx.x;
            const int tot = wcnt[l][0] + wcnt[l][1] + wcnt[l][2] + wcnt[l][3];
            int base = tot ? atomicAdd(l ? far_count : close_count, tot) : 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { wbase[l][w] = base; base += wcnt[l][w]; }
        }
        __syncthreads();
        int sc = wbase[0][wave] + pcl - ncl, sf = wbase[1][wave] + pfr - nfr;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            if (cl[i]) close_list[sc++] = idx[i];
            if (fr[i]) far_list[sf++] = idx[i];
        }
        __syncthreads();          // wcnt / wbase are rewritten by the next iteration
    }
}

// every voxel of a pruned block takes the value of the block's corner
__global__ void fill_kernel(const int *far_list, const int *far_count, int s, int log2s, int N, float *udf,
                            float grad_thr, int *grad_list, int *grad_count) {
    const long total = (long)(*far_count) << (3 * log2s);
    const int mask = s - 1;
    for (long e0 = blockIdx.x * (long)blockDim.x; e0 < total; e0 += (long)gridDim.x * blockDim.x) {
        const long e = e0 + threadIdx.x;
        bool want_grad = false;
        int q = 0;
        if (e < total) {
            const int b = (int)(e >> (3 * log2s));
            const int w = (int)(e & ((1 << (3 * log2s)) - 1));
            const int dk = w & mask, dj = (w >> log2s) & mask, di = w >> (2 * log2s);
            const int p = far_list[b];
            const float v = udf[p];
            if (w != 0) {
                q = p + di * N * N + dj * N + dk;
                udf[q] = v;
                // only reachable for fields that go negative: the reference's gradient mask is on
                // the signed value (meshudf.py:200) while pruning is on |value| (:186)
                want_grad = grad_list != nullptr && v < grad_thr;
            }
        }
        if (grad_list) {
            const int slot = wave_append_slot(grad_count, want_grad);
            if (slot >= 0) grad_list[slot] = q;
        }
    }
}

__global__ void emit_points_kernel(PtIO io, float *xyz) {
    const long n = pt_count(io);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        float x, y, z;
        voxel_xyz(io, pt_voxel(io, e), x, y, z);
        xyz[e * 3 + 0] = x; xyz[e * 3 + 1] = y; xyz[e * 3 + 2] = z;
    }
}

// grid-shard mode, compact exchange ("all-gather of the level's values"): rank r's 64-point tiles r, r + world,
// ... travel as ONE contiguous segment of `seg` points per rank (tile t of the list = tile t / world of rank t % world's segment);
// the gathered buffer is [world][seg].  world <= 1: the buffer is indexed by point number.
__device__ __forceinline__ long shard_gathered_index(int world, long seg, long e) {
    if (world <= 1) return e;
    const long t = e >> 6;
    return (t % world) * seg + (t / world) * 64 + (e & 63);
}

__global__ void commit_kernel(PtIO io, const float *vals, int world, long seg) {
    const long n = pt_count(io);
    for (long e0 = blockIdx.x * (long)blockDim.x; e0 < n; e0 += (long)gridDim.x * blockDim.x) {
        const long e = e0 + threadIdx.x;
        bool want = false;
        int idx = 0;
        if (e < n) {
            idx = pt_voxel(io, e);
            const float v = vals[shard_gathered_index(world, seg, e)];
            io.grid_udf[idx] = v;
            want = io.grad_list != nullptr && v < io.grad_thr;
        }
        if (io.grad_list) {
            const int slot = wave_append_slot(io.grad_count, want);
            if (slot >= 0) io.grad_list[slot] = idx;
        }
    }
}

__global__ void grad_commit_kernel(const int *list, long n, const float *ng, float *grads) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long idx = list[e];
        grads[idx * 3 + 0] = ng[e * 3 + 0]; grads[idx * 3 + 1] = ng[e * 3 + 1]; grads[idx * 3 + 2] = ng[e * 3 + 2];
    }
}

// Ordered stream compaction of a voxel predicate over [0, n): out[] = the indices with pred(i), ascending; *count = how many.
// Wave w owns the contiguous range [w * SEL_SPAN, (w + 1) * SEL_SPAN): pass 1 counts, a single workgroup scans the wave
// counts, pass 2 writes with ballot prefixes (no workgroup barrier in either loop).  Two coalesced reads of the predicate's
// input instead of a general-purpose device-wide partition (1.4 ms per 512^3 volume -> the 0.3 ms two reads of 512 MB cost).
constexpr int SEL_SPAN = 16384;
template <class Pred>
__global__ __launch_bounds__(256) void select_count_kernel(Pred pred, long n, int *wave_counts) {
    const long w = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long lo = w * SEL_SPAN, hi = min(n, lo + SEL_SPAN);
    int c = 0;
    for (long e = lo + lane; e < hi; e += 64) c += pred((int)e) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0 && lo < n) wave_counts[w] = c;
}
// exclusive scan of `m` wave counts in place (m <= 16384), total -> *count
__global__ __launch_bounds__(1024) void select_scan_kernel(int *wave_counts, int m, int *count) {
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (m + 1023) / 1024;
    int s = 0;
    for (int i = 0; i < per; ++i) { const int k = t * per + i; if (k < m) s += wave_counts[k]; }
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = 0; i < per; ++i) { const int k = t * per + i; if (k < m) { const int c = wave_counts[k]; wave_counts[k] = run; run += c; } }
    if (t == 1023) *count = part[1023];
}
struct Identity { __device__ __forceinline__ int operator()(int i) const { return i; } };
template <class Pred, class Xf = Identity>
__global__ __launch_bounds__(256) void select_write_kernel(Pred pred, long n, const int *wave_offsets, int *out, Xf xf = Xf()) {
    const long w = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long lo = w * SEL_SPAN, hi = min(n, lo + SEL_SPAN);
    if (lo >= n) return;
    int off = wave_offsets[w];
    for (long e0 = lo; e0 < hi; e0 += 64) {
        const long e = e0 + lane;
        const bool p = e < hi && pred((int)e);
        const unsigned long long m = __ballot(p);
        if (p) out[off + __popcll(m & ((1ull << lane) - 1ull))] = xf((int)e);
        off += __popcll(m);
    }
}

// ---- grid-shard mode: the same classification with a VOXEL-ORDERED result --------------------------------------------
// Several ranks evaluate one shape's grid: rank r takes the 64-point tiles r, r + G, ... of every level's point list, so the
// list must be the same on every rank — and classify_kernel's appends are in scheduling order.  Here a level is classified
// over its whole lattice (n_l^3 <= 256^3 points: 1 byte of the parent level's flags + 4 bytes of udf each): flag_l[li] = the
// lattice point is active (level 0: all; else its parent block was close) and close; the next level's parents are then the
// ORDERED compaction of the flags (count on the device, no sort), the far blocks are appended in any order (the fill does not
// care).  Same sets as classify_kernel, hence the same grid as the fused fill, bit for bit.
struct LatticeGeom { int n, log2n, N, s; };      // lattice width (power of two), grid width, stride
__device__ __forceinline__ int lattice_voxel(const LatticeGeom &G, int li) {
    const int K = li & (G.n - 1), J = (li >> G.log2n) & (G.n - 1), I = li >> (2 * G.log2n);
    return ((I * G.N + J) * G.N + K) * G.s;
}
__global__ __launch_bounds__(256) void classify_flag_kernel(LatticeGeom G, const unsigned char *parent_flag, const float *udf, float thr,
                                                            unsigned char *flag, int *far_list, int *far_count) {
    const long n3 = 1L << (3 * G.log2n);
    for (long e0 = blockIdx.x * (long)blockDim.x; e0 < n3; e0 += (long)gridDim.x * blockDim.x) {
        const long e = e0 + threadIdx.x;
        bool far = false;
        int vox = 0;
        if (e < n3) {
            const int li = (int)e;
            bool active = true;
            if (parent_flag) {
                const int K = li & (G.n - 1), J = (li >> G.log2n) & (G.n - 1), I = li >> (2 * G.log2n);
                active = parent_flag[((((I >> 1) << (G.log2n - 1)) + (J >> 1)) << (G.log2n - 1)) + (K >> 1)] != 0;
            }
            vox = lattice_voxel(G, li);
            const bool close = active && fabsf(udf[vox]) < thr;
            flag[li] = close ? 1 : 0;
            far = active && !close;
        }
        const int slot = wave_append_slot(far_count, far);
        if (slot >= 0) far_list[slot] = vox;
    }
}
struct FlagSet {
    const unsigned char *flag;
    __device__ __forceinline__ bool operator()(const int &i) const { return flag[i] != 0; }
};
struct LatticeToVoxel {
    LatticeGeom G;
    __device__ __forceinline__ int operator()(int li) const { return lattice_voxel(G, li); }
};
__global__ void grad_commit_dev_kernel(const int *list, const int *count, long cap, const float *ng, float *grads, int world, long seg) {
    const long n = min((long)*count, cap);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long idx = list[e], ge = shard_gathered_index(world, seg, e);
        grads[idx * 3 + 0] = ng[ge * 3 + 0]; grads[idx * 3 + 1] = ng[ge * 3 + 1]; grads[idx * 3 + 2] = ng[ge * 3 + 2];
    }
}

// this rank's tiles of a point-indexed value buffer -> its contiguous segment (what the all-gather sends): seg[lt * 64 + i] =
// vals[(lt * world + rank) * 64 + i] for the tiles that exist (n = points of the list, device-side count applied by pt_count)
__global__ void shard_pack_kernel(PtIO io, int rank, int world, int width, const float *vals, float *seg) {
    const long n = pt_count(io);
    for (long le = blockIdx.x * (long)blockDim.x + threadIdx.x;; le += (long)gridDim.x * blockDim.x) {
        const long e = ((le >> 6) * world + rank) * 64 + (le & 63);
        if ((((le >> 6) * world + rank) << 6) >= n) break;          // this tile (and every later one of this thread) lies behind the list
        if (e < n)
            for (int w = 0; w < width; ++w) seg[le * width + w] = vals[e * width + w];
    }
}

// grid-shard mode: a list longer than the exchange buffer it travels in is cut (the buffers have a caller-chosen capacity, the
// count lives on the device) — never silently: every commit compares and counts (surfd_grid_shard_overflows)
__global__ void shard_overflow_kernel(const int *count, int per_entry, long cap, unsigned long long *overflow) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && (long)per_entry * (long)*count > cap) *overflow += 1ull;
}

// counters of the fill that just finished -> running totals of the handle (device side, no host sync)
__global__ void accumulate_totals_kernel(const int *counters, unsigned long long *totals, int n_levels, long level0, long dense_n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (dense_n > 0) totals[n_levels - 1] += (unsigned long long)dense_n;
    else
        for (int l = 0; l < n_levels; ++l) totals[l] += l == 0 ? (unsigned long long)level0 : 7ull * (unsigned long long)counters[CTR_PARENT + l];
    totals[SURFD_GRID_MAX_LEVELS] += (unsigned long long)counters[CTR_GRAD];
    totals[SURFD_GRID_MAX_LEVELS + 1] += 1ull;
}

}  // namespace surfd

using namespace surfd;

struct surfd_grid {
    int N = 0, n_levels = 0, levels[SURFD_GRID_MAX_LEVELS] = {};
    float refine[SURFD_GRID_MAX_LEVELS] = {};
    float grad_thr = 0.f, voxel = 0.f, origin = -1.f;
    bool thresholds_set = false, allocated = false;
    int *parents[SURFD_GRID_MAX_LEVELS] = {};
    int *far_list = nullptr, *grad_list = nullptr, *counters = nullptr;
    float *cur_udf = nullptr, *cur_grads = nullptr;   // callback path
    bool dense_last = false;
    long dense_n = 0;
    int *sort_out = nullptr; long sort_cap = 0; void *sort_tmp = nullptr; size_t sort_tmp_bytes = 0;   // callback path: deterministic list order (allocated on first use)
    unsigned long long *totals = nullptr;            // [MAX_LEVELS] forward queries per level, [MAX_LEVELS] gradient queries, [MAX_LEVELS + 1] fills — since the last reset
    void *sel_tmp = nullptr; size_t sel_tmp_bytes = 0;                                // fused path: ordered compaction of the gradient voxels
    unsigned char *flags[SURFD_GRID_MAX_LEVELS] = {};                                 // grid-shard mode: close flags per level lattice (allocated on first use)
    // grid-shard protocol state: the level the next eval / commit must name (-1: no sharded fill open; n_levels: the gradient
    // step), the capacity its eval calls used, whether the gradient list has been evaluated
    int shard_level = -1;
    int64_t shard_cap = 0;
    bool shard_evaluated = false;
    unsigned long long *overflow = nullptr;          // device: exchange buffers that were shorter than their list, since the last reset (grid-shard mode)
};

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// The compacted lists are appended with one atomicAdd per wave, so their ORDER depends on scheduling.  The native
// fill never looks at the order, but the callback path hands the points to the host in list order, and
// parallel.ShardedField splits that order over ranks: every rank must see the same one -> sort by voxel index
// (the host is synchronised at these points anyway).
static int sort_list(surfd_grid *g, int *list, long n, hipStream_t st) {
    if (n <= 1) return SURFD_OK;
    if (n > g->sort_cap) {                 // only the callback path sorts: sized to the lists it actually sees, not N^3 per handle
        if (g->sort_out) HIP_TRY(hipFree(g->sort_out));
        g->sort_out = nullptr;
        const long cap = std::max<long>(n + n / 4, 1 << 16);
        HIP_TRY(hipMalloc((void **)&g->sort_out, cap * sizeof(int)));
        g->sort_cap = cap;
    }
    size_t tmp_bytes = 0;
    const int end_bit = 3 * ilog2(g->N) + 1;
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, list, g->sort_out, (int)n, 0, end_bit, st));
    if (tmp_bytes > g->sort_tmp_bytes) {
        if (g->sort_tmp) HIP_TRY(hipFree(g->sort_tmp));
        g->sort_tmp = nullptr;
        HIP_TRY(hipMalloc(&g->sort_tmp, tmp_bytes));
        g->sort_tmp_bytes = tmp_bytes;
    }
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(g->sort_tmp, tmp_bytes, list, g->sort_out, (int)n, 0, end_bit, st));
    HIP_TRY(hipMemcpyAsync(list, g->sort_out, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, st));
    return SURFD_OK;
}

// Gradient voxels of the fused fill = all voxels of the finished grid below the threshold, in voxel order, by one ordered
// stream compaction (count stays on the device).  The forward kernels could collect the same SET with atomic appends,
// but in a different order every run, and the f16x2 gradient kernel's per-tile power-of-two scaling makes the last bits
// of a direction depend on which points share its 64-point tile: ordered list -> a fill is bitwise reproducible.
struct BelowThreshold {
    const float *udf; float thr;
    __device__ __forceinline__ bool operator()(const int &i) const { return udf[i] < thr; }
};

template <class Pred, class Xf = Identity>
static int ordered_select(Pred pred, long n, int *wave_tmp, int *out, int *count, hipStream_t st, Xf xf = Xf()) {
    const int m = (int)ceil_div<long>(n, SEL_SPAN);
    const unsigned blocks = (unsigned)ceil_div(m, 4);
    hipLaunchKernelGGL((select_count_kernel<Pred>), dim3(blocks), dim3(256), 0, st, pred, n, wave_tmp);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(1024), 0, st, wave_tmp, m, count);
    LAUNCH_CHECK();
    hipLaunchKernelGGL((select_write_kernel<Pred, Xf>), dim3(blocks), dim3(256), 0, st, pred, n, (const int *)wave_tmp, out, xf);
    LAUNCH_CHECK();
    return SURFD_OK;
}

static int compact_grad_list(surfd_grid *g, const float *udf, float thr, hipStream_t st) {
    const long N3 = (long)g->N * g->N * g->N;
    return ordered_select(BelowThreshold{udf, thr}, N3, (int *)g->sel_tmp, g->grad_list, g->counters + CTR_GRAD, st);
}

static int grid_alloc(surfd_grid *g) {
    if (g->allocated) return SURFD_OK;
    const long N3 = (long)g->N * g->N * g->N;
    for (int l = 1; l < g->n_levels; ++l) {
        const long cap = (long)g->levels[l - 1] * g->levels[l - 1] * g->levels[l - 1];
        HIP_TRY(hipMalloc((void **)&g->parents[l], cap * sizeof(int)));
    }
    const int nl = g->n_levels;
    const long far_cap = nl >= 2 ? (long)g->levels[nl - 2] * g->levels[nl - 2] * g->levels[nl - 2] : 1;
    HIP_TRY(hipMalloc((void **)&g->far_list, far_cap * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&g->grad_list, N3 * sizeof(int)));
    g->sel_tmp_bytes = (size_t)ceil_div<long>(N3, SEL_SPAN) * sizeof(int);      // wave counts / offsets of the ordered compaction
    HIP_TRY(hipMalloc(&g->sel_tmp, g->sel_tmp_bytes));
    HIP_TRY(hipMalloc((void **)&g->counters, CTR_TOTAL * sizeof(int)));
    HIP_TRY(hipMemset(g->counters, 0, CTR_TOTAL * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&g->totals, (SURFD_GRID_MAX_LEVELS + 2) * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(g->totals, 0, (SURFD_GRID_MAX_LEVELS + 2) * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void **)&g->overflow, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(g->overflow, 0, sizeof(unsigned long long)));
    g->allocated = true;
    return SURFD_OK;
}

static PtIO base_io(const surfd_grid *g) {
    PtIO io;
    memset(&io, 0, sizeof(io));
    io.N = g->N; io.voxel = g->voxel; io.origin = g->origin; io.s = 1;
    if (g->N > 0 && (g->N & (g->N - 1)) == 0) { int l = 0; while ((1 << l) < g->N) ++l; io.log2N1 = l + 1; }
    return io;
}

// enumeration of the points evaluated at `level`
static PtIO eval_io(const surfd_grid *g, int level) {
    PtIO io = base_io(g);
    io.s = g->N / g->levels[level];
    if (level == 0) {
        io.mode = PT_LATTICE;
        io.n = (long)g->levels[0] * g->levels[0] * g->levels[0];
    } else {
        io.mode = PT_CHILDREN;
        io.list = g->parents[level];
        io.count_dev = g->counters + CTR_PARENT + level;
    }
    return io;
}

// classification of all active lattice points of `level`, then pruned-block fill
static int refine_level(surfd_grid *g, int level, float *udf, bool want_grads, hipStream_t st) {
    if (g->levels[level] >= g->N) return SURFD_OK;
    PtIO io = eval_io(g, level);
    if (level > 0) io.mode = PT_CHILDREN8;
    const int blocks = 1024;
    hipLaunchKernelGGL(classify_kernel, dim3(blocks), dim3(256), 0, st, io, (const float *)udf, g->refine[level],
                       g->parents[level + 1], g->counters + CTR_PARENT + level + 1, g->far_list,
                       g->counters + CTR_FAR + level);
    LAUNCH_CHECK();
    const int s = g->N / g->levels[level];
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, st, (const int *)g->far_list,
                       (const int *)(g->counters + CTR_FAR + level), s, ilog2(s), g->N, udf, g->grad_thr,
                       want_grads ? g->grad_list : nullptr, g->counters + CTR_GRAD);
    LAUNCH_CHECK();
    return SURFD_OK;
}

static int add_to_totals(surfd_grid *g, long dense_n, hipStream_t st) {
    hipLaunchKernelGGL(accumulate_totals_kernel, dim3(1), dim3(64), 0, st, (const int *)g->counters, g->totals, g->n_levels,
                       (long)g->levels[0] * g->levels[0] * g->levels[0], dense_n);
    LAUNCH_CHECK();
    return SURFD_OK;
}

static int check_ready(const surfd_grid *g, const char *fn) {
    if (!g) SURFD_FAIL(SURFD_ERR_ARG, "%s: null handle", fn);
    if (!g->thresholds_set) SURFD_FAIL(SURFD_ERR_STATE, "%s: call surfd_grid_set_thresholds first", fn);
    return SURFD_OK;
}

extern "C" {

int surfd_grid_create(int N, surfd_grid **out) {
    if (!out) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_create: out is null");
    if (N < 64 || (N & (N - 1)) || N > 1024)
        SURFD_FAIL(SURFD_ERR_UNSUPPORTED, "surfd_grid_create: N must be a power of two in [64, 1024], got %d", N);
    auto *g = new surfd_grid();
    g->N = N;
    // levels 32, 64, ..., N   (int(log2(N) - 4) of them, meshudf.py:46)
    for (int n = 32; n <= N && g->n_levels < SURFD_GRID_MAX_LEVELS; n *= 2) g->levels[g->n_levels++] = n;
    *out = g;
    return SURFD_OK;
}

void surfd_grid_destroy(surfd_grid *g) {
    if (!g) return;
    for (int l = 0; l < SURFD_GRID_MAX_LEVELS; ++l)
        if (g->parents[l]) (void)hipFree(g->parents[l]);
    if (g->far_list) (void)hipFree(g->far_list);
    if (g->grad_list) (void)hipFree(g->grad_list);
    if (g->counters) (void)hipFree(g->counters);
    if (g->totals) (void)hipFree(g->totals);
    if (g->overflow) (void)hipFree(g->overflow);
    if (g->sort_out) (void)hipFree(g->sort_out);
    if (g->sort_tmp) (void)hipFree(g->sort_tmp);
    if (g->sel_tmp) (void)hipFree(g->sel_tmp);
    for (int l = 0; l < SURFD_GRID_MAX_LEVELS; ++l)
        if (g->flags[l]) (void)hipFree(g->flags[l]);
    delete g;
}

int surfd_grid_set_thresholds(surfd_grid *g, const float *refine, int n_levels, float grad_thr, float voxel, float origin) {
    if (!g || !refine) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_set_thresholds: null argument");
    if (n_levels != g->n_levels) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_set_thresholds: %d levels given, grid has %d", n_levels, g->n_levels);
    for (int l = 0; l < n_levels; ++l) g->refine[l] = refine[l];
    g->grad_thr = grad_thr; g->voxel = voxel; g->origin = origin;
    g->thresholds_set = true;
    return SURFD_OK;
}

int surfd_grid_fill(surfd_grid *g, surfd_decoder *d, int sample, float *udf, float *grads, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_fill");
    if (rc) return rc;
    if (!udf) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_fill: udf is null");
    if ((rc = grid_alloc(g))) return rc;
    hipStream_t st = as_stream(s);
    const long N3 = (long)g->N * g->N * g->N;
    HIP_TRY(hipMemsetAsync(g->counters, 0, CTR_TOTAL * sizeof(int), st));
    if (grads) HIP_TRY(hipMemsetAsync(grads, 0, N3 * 3 * sizeof(float), st));
    for (int l = 0; l < g->n_levels; ++l) {
        PtIO io = eval_io(g, l);
        io.grid_udf = udf;
        if ((rc = decoder_launch(d, sample, io, false, l == 0 ? ceil_div<long>(io.n, 64) : -1, st))) return rc;
        if ((rc = refine_level(g, l, udf, false, st))) return rc;
    }
    if (grads) {
        if ((rc = compact_grad_list(g, udf, g->grad_thr, st))) return rc;
        PtIO io = base_io(g);
        io.mode = PT_LIST; io.list = g->grad_list; io.count_dev = g->counters + CTR_GRAD; io.grid_grads = grads;
        if ((rc = decoder_launch(d, sample, io, true, -1, st))) return rc;
    }
    g->dense_last = false;
    return add_to_totals(g, 0, st);
}

// The grids of `n` shapes (one handle each, all of one resolution) level by level TOGETHER: the decoder evaluates a level
// of all shapes in one persistent launch (points.h: PtBatch), the bookkeeping kernels run per shape in between.  Same
// values as n calls of surfd_grid_fill, bit for bit (tiles never mix shapes).
int surfd_grid_fill_batch(surfd_grid *const *gs, int n, surfd_decoder *d, const int *samples, float *const *udfs,
                          float *const *grads, surfd_stream s) {
    if (!gs || !samples || !udfs || n < 1 || n > PT_BATCH_MAX)
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_fill_batch: need 1..%d grids", PT_BATCH_MAX);
    hipStream_t st = as_stream(s);
    int rc;
    for (int i = 0; i < n; ++i) {
        if ((rc = check_ready(gs[i], "surfd_grid_fill_batch"))) return rc;
        if (!udfs[i]) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_fill_batch: udf %d is null", i);
        if (gs[i]->N != gs[0]->N || gs[i]->n_levels != gs[0]->n_levels)
            SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_fill_batch: all grids must have one resolution");
        for (int j = 0; j < i; ++j)
            if (gs[j] == gs[i]) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_fill_batch: every shape needs its own grid handle");
        if ((rc = grid_alloc(gs[i]))) return rc;
        const long N3 = (long)gs[i]->N * gs[i]->N * gs[i]->N;
        HIP_TRY(hipMemsetAsync(gs[i]->counters, 0, CTR_TOTAL * sizeof(int), st));
        if (grads && grads[i]) HIP_TRY(hipMemsetAsync(grads[i], 0, N3 * 3 * sizeof(float), st));
    }
    for (int l = 0; l < gs[0]->n_levels; ++l) {
        PtBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n;
        for (int i = 0; i < n; ++i) {
            b.sample[i] = samples[i];
            b.io[i] = eval_io(gs[i], l);
            b.io[i].grid_udf = udfs[i];
        }
        if ((rc = decoder_launch_batch(d, b, false, l == 0 ? ceil_div<long>(b.io[0].n, 64) : -1, st))) return rc;
        for (int i = 0; i < n; ++i)
            if ((rc = refine_level(gs[i], l, udfs[i], false, st))) return rc;
    }
    PtBatch gb;
    memset(&gb, 0, sizeof(gb));
    for (int i = 0; i < n; ++i) {
        gs[i]->dense_last = false;
        if (!grads || !grads[i]) continue;
        if ((rc = compact_grad_list(gs[i], udfs[i], gs[i]->grad_thr, st))) return rc;
        PtIO io = base_io(gs[i]);
        io.mode = PT_LIST; io.list = gs[i]->grad_list; io.count_dev = gs[i]->counters + CTR_GRAD; io.grid_grads = grads[i];
        gb.sample[gb.n] = samples[i];
        gb.io[gb.n++] = io;
    }
    if (gb.n && (rc = decoder_launch_batch(d, gb, true, -1, st))) return rc;
    for (int i = 0; i < n; ++i)
        if ((rc = add_to_totals(gs[i], 0, st))) return rc;
    return SURFD_OK;
}

int surfd_grid_fill_dense(surfd_grid *g, surfd_decoder *d, int sample, float grad_below, float *udf, float *grads,
                          surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_fill_dense");
    if (rc) return rc;
    if (!udf) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_fill_dense: udf is null");
    if ((rc = grid_alloc(g))) return rc;
    hipStream_t st = as_stream(s);
    const long N3 = (long)g->N * g->N * g->N;
    HIP_TRY(hipMemsetAsync(g->counters, 0, CTR_TOTAL * sizeof(int), st));
    if (grads) HIP_TRY(hipMemsetAsync(grads, 0, N3 * 3 * sizeof(float), st));
    PtIO io = base_io(g);
    io.mode = PT_DENSE; io.n = N3; io.grid_udf = udf;
    if ((rc = decoder_launch(d, sample, io, false, ceil_div<long>(N3, 64), st))) return rc;
    if (grads) {
        if ((rc = compact_grad_list(g, udf, grad_below, st))) return rc;
        PtIO gi = base_io(g);
        gi.mode = PT_LIST; gi.list = g->grad_list; gi.count_dev = g->counters + CTR_GRAD; gi.grid_grads = grads;
        if ((rc = decoder_launch(d, sample, gi, true, -1, st))) return rc;
    }
    g->dense_last = true;
    g->dense_n = N3;
    return add_to_totals(g, N3, st);
}

int surfd_grid_get_stats(surfd_grid *g, surfd_grid_stats *out, surfd_stream s) {
    if (!g || !out) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_get_stats: null argument");
    if (!g->allocated) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_get_stats: nothing has been filled yet");
    int c[CTR_TOTAL];
    HIP_TRY(hipMemcpyAsync(c, g->counters, sizeof(c), hipMemcpyDeviceToHost, as_stream(s)));
    HIP_TRY(hipStreamSynchronize(as_stream(s)));
    memset(out, 0, sizeof(*out));
    out->n_levels = g->n_levels;
    for (int l = 0; l < g->n_levels; ++l) {
        out->levels[l] = g->levels[l];
        out->fwd_points[l] = l == 0 ? (long)g->levels[0] * g->levels[0] * g->levels[0] : 7L * c[CTR_PARENT + l];
    }
    if (g->dense_last) {
        for (int l = 0; l < g->n_levels; ++l) out->fwd_points[l] = 0;
        out->fwd_points[g->n_levels - 1] = g->dense_n;
    }
    out->grad_points = c[CTR_GRAD];
    return SURFD_OK;
}

int surfd_grid_get_totals(surfd_grid *g, surfd_grid_stats *out, int64_t *fills, int reset, surfd_stream s) {
    if (!g || !out) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_get_totals: null argument");
    if (!g->allocated) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_get_totals: nothing has been filled yet");
    unsigned long long t[SURFD_GRID_MAX_LEVELS + 2];
    HIP_TRY(hipMemcpyAsync(t, g->totals, sizeof(t), hipMemcpyDeviceToHost, as_stream(s)));
    if (reset) HIP_TRY(hipMemsetAsync(g->totals, 0, sizeof(t), as_stream(s)));
    HIP_TRY(hipStreamSynchronize(as_stream(s)));
    memset(out, 0, sizeof(*out));
    out->n_levels = g->n_levels;
    for (int l = 0; l < g->n_levels; ++l) { out->levels[l] = g->levels[l]; out->fwd_points[l] = (int64_t)t[l]; }
    out->grad_points = (int64_t)t[SURFD_GRID_MAX_LEVELS];
    if (fills) *fills = (int64_t)t[SURFD_GRID_MAX_LEVELS + 1];
    return SURFD_OK;
}

// ---- sparse hand-off to the host mesher -------------------------------------------------------------
// The UDF marching cubes only ever looks at voxels whose value is at most max_thr = 1.74 voxel (mcubes.cpp): instead of
// 16 N^3 bytes per shape (2.1 GB at 512^3) the device compacts that band — voxel index, value, gradient, in voxel order —
// and the host copies count + band over a copy stream into pinned memory.
}  // extern "C"

namespace surfd {
struct InBand {
    const float *udf; float thr;
    __device__ __forceinline__ bool operator()(const int &i) const { return !(udf[i] > thr); }
};
__global__ void band_gather_kernel(const int *idx, const int *count, long cap, const float *udf, const float *grads, float *packed) {
    const long n = min((long)*count, cap);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long p = idx[e];
        const float v = udf[p];
        f32x4 o;
        o[0] = v < 0.f ? 0.f : v;                        // get_mesh_from_udf clamps the grid at zero before meshing (meshudf.py:338)
        o[1] = grads[3 * p]; o[2] = grads[3 * p + 1]; o[3] = grads[3 * p + 2];
        *reinterpret_cast<f32x4 *>(packed + 4 * e) = o;
    }
}
}  // namespace surfd

struct surfd_band {
    int N = 0; long cap = 0;
    int *idx = nullptr, *count = nullptr; float *packed = nullptr;      // device
    void *tmp = nullptr; size_t tmp_bytes = 0;
    int *h_idx = nullptr, *h_count = nullptr; float *h_packed = nullptr;   // pinned host
    hipEvent_t ready = nullptr;
};

extern "C" {

int surfd_band_create(int N, int64_t capacity, surfd_band **out) {
    if (!out || N < 2 || N > 1024 || capacity < 1 || capacity > (int64_t)N * N * N) SURFD_FAIL(SURFD_ERR_ARG, "surfd_band_create: bad argument");
    auto *b = new surfd_band();
    b->N = N; b->cap = capacity;
    const int N3 = N * N * N;
    // the ordered select writes every selected index: it needs room for the worst case, the gather and the copy only `capacity`
    HIP_TRY(hipMalloc((void **)&b->idx, (size_t)N3 * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&b->count, sizeof(int)));
    HIP_TRY(hipMalloc((void **)&b->packed, (size_t)capacity * 4 * sizeof(float)));
    b->tmp_bytes = (size_t)ceil_div<long>(N3, SEL_SPAN) * sizeof(int);
    HIP_TRY(hipMalloc(&b->tmp, b->tmp_bytes));
    HIP_TRY(hipHostMalloc((void **)&b->h_idx, (size_t)capacity * sizeof(int), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&b->h_packed, (size_t)capacity * 4 * sizeof(float), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&b->h_count, sizeof(int), hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&b->ready, hipEventDisableTiming));
    *out = b;
    return SURFD_OK;
}

void surfd_band_destroy(surfd_band *b) {
    if (!b) return;
    if (b->idx) (void)hipFree(b->idx);
    if (b->count) (void)hipFree(b->count);
    if (b->packed) (void)hipFree(b->packed);
    if (b->tmp) (void)hipFree(b->tmp);
    if (b->h_idx) (void)hipHostFree(b->h_idx);
    if (b->h_packed) (void)hipHostFree(b->h_packed);
    if (b->h_count) (void)hipHostFree(b->h_count);
    if (b->ready) (void)hipEventDestroy(b->ready);
    delete b;
}

int surfd_band_compact(surfd_band *b, const float *udf, const float *grads, float max_thr, surfd_stream s) {
    if (!b || !udf || !grads) SURFD_FAIL(SURFD_ERR_ARG, "surfd_band_compact: null argument");
    hipStream_t st = as_stream(s);
    const long N3 = (long)b->N * b->N * b->N;
    int rc = ordered_select(InBand{udf, max_thr}, N3, (int *)b->tmp, b->idx, b->count, st);
    if (rc) return rc;
    hipLaunchKernelGGL(band_gather_kernel, dim3(1024), dim3(256), 0, st, (const int *)b->idx, (const int *)b->count, b->cap, udf, grads, b->packed);
    LAUNCH_CHECK();
    HIP_TRY(hipEventRecord(b->ready, st));
    return SURFD_OK;
}

int surfd_band_fetch(surfd_band *b, surfd_stream copy_stream, int64_t *count, const int32_t **index, const float **packed) {
    if (!b || !count || !index || !packed) SURFD_FAIL(SURFD_ERR_ARG, "surfd_band_fetch: null argument");
    hipStream_t cs = as_stream(copy_stream);
    HIP_TRY(hipStreamWaitEvent(cs, b->ready, 0));
    HIP_TRY(hipMemcpyAsync(b->h_count, b->count, sizeof(int), hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipStreamSynchronize(cs));
    const long n = *b->h_count;
    if (n > b->cap) SURFD_FAIL(SURFD_ERR_STATE, "surfd_band_fetch: the band holds %ld voxels, the handle was created for %ld", n, b->cap);
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(b->h_idx, b->idx, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipMemcpyAsync(b->h_packed, b->packed, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipStreamSynchronize(cs));
    }
    *count = n; *index = b->h_idx; *packed = b->h_packed;
    return SURFD_OK;
}

// ---- callback path -------------------------------------------------------------------------
int surfd_grid_begin(surfd_grid *g, float *udf, float *grads, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_begin");
    if (rc) return rc;
    if (!udf) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_begin: udf is null");
    if ((rc = grid_alloc(g))) return rc;
    hipStream_t st = as_stream(s);
    const long N3 = (long)g->N * g->N * g->N;
    HIP_TRY(hipMemsetAsync(g->counters, 0, CTR_TOTAL * sizeof(int), st));
    if (grads) HIP_TRY(hipMemsetAsync(grads, 0, N3 * 3 * sizeof(float), st));
    g->cur_udf = udf; g->cur_grads = grads; g->dense_last = false;
    g->shard_level = -1;
    return SURFD_OK;
}

static int level_count(surfd_grid *g, int level, long *n, hipStream_t st) {
    if (level == 0) { *n = (long)g->levels[0] * g->levels[0] * g->levels[0]; return SURFD_OK; }
    int c = 0;
    HIP_TRY(hipMemcpyAsync(&c, g->counters + CTR_PARENT + level, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *n = 7L * c;
    return SURFD_OK;
}

int surfd_grid_level_points(surfd_grid *g, int level, float *xyz, int64_t capacity, int64_t *n, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_level_points");
    if (rc) return rc;
    if (!g->cur_udf) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_level_points: call surfd_grid_begin first");
    if (level < 0 || level >= g->n_levels || !n) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_level_points: bad level %d", level);
    long cnt = 0;
    if ((rc = level_count(g, level, &cnt, as_stream(s)))) return rc;
    *n = cnt;
    if (!xyz || cnt == 0) return SURFD_OK;
    if (capacity < cnt) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_level_points: capacity %lld < %ld points", (long long)capacity, cnt);
    if (level > 0 && (rc = sort_list(g, g->parents[level], cnt / 7, as_stream(s)))) return rc;
    PtIO io = eval_io(g, level);
    hipLaunchKernelGGL(emit_points_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(cnt, 256), 4096)), dim3(256), 0,
                       as_stream(s), io, xyz);
    LAUNCH_CHECK();
    return SURFD_OK;
}

int surfd_grid_level_commit(surfd_grid *g, int level, const float *values, int64_t n, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_level_commit");
    if (rc) return rc;
    if (!g->cur_udf) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_level_commit: call surfd_grid_begin first");
    if (level < 0 || level >= g->n_levels) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_level_commit: bad level %d", level);
    hipStream_t st = as_stream(s);
    if (n > 0) {
        if (!values) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_level_commit: values is null");
        PtIO io = eval_io(g, level);
        io.grid_udf = g->cur_udf;
        if (g->cur_grads) { io.grad_list = g->grad_list; io.grad_count = g->counters + CTR_GRAD; io.grad_thr = g->grad_thr; }
        hipLaunchKernelGGL(commit_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(n, 256), 4096)), dim3(256), 0, st, io, values, 1, 0L);
        LAUNCH_CHECK();
    }
    return refine_level(g, level, g->cur_udf, g->cur_grads != nullptr, st);
}

int surfd_grid_grad_points(surfd_grid *g, float *xyz, int64_t capacity, int64_t *n, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_grad_points");
    if (rc) return rc;
    if (!g->cur_udf || !n) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_grad_points: call surfd_grid_begin first");
    int c = 0;
    HIP_TRY(hipMemcpyAsync(&c, g->counters + CTR_GRAD, sizeof(int), hipMemcpyDeviceToHost, as_stream(s)));
    HIP_TRY(hipStreamSynchronize(as_stream(s)));
    *n = c;
    if (!xyz || c == 0) return SURFD_OK;
    if (capacity < c) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_grad_points: capacity %lld < %d points", (long long)capacity, c);
    if ((rc = sort_list(g, g->grad_list, c, as_stream(s)))) return rc;
    PtIO io = base_io(g);
    io.mode = PT_LIST; io.list = g->grad_list; io.n = c;
    hipLaunchKernelGGL(emit_points_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(c, 256), 4096)), dim3(256), 0,
                       as_stream(s), io, xyz);
    LAUNCH_CHECK();
    return SURFD_OK;
}

int surfd_grid_grad_commit(surfd_grid *g, const float *ngrads, int64_t n, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_grad_commit");
    if (rc) return rc;
    if (!g->cur_grads) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_grad_commit: no gradient buffer was given to surfd_grid_begin");
    if (n <= 0) return SURFD_OK;
    if (!ngrads) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_grad_commit: ngrads is null");
    hipLaunchKernelGGL(grad_commit_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(n, 256), 4096)), dim3(256), 0,
                       as_stream(s), (const int *)g->grad_list, (long)n, ngrads, g->cur_grads);
    LAUNCH_CHECK();
    return SURFD_OK;
}


// ---- grid-shard mode, native (north star: "shard the per-sample 512^3 grid evaluation across the GPUs") ---------
// One shape, `world` ranks, every rank runs the same calls on its own device:
//   begin -> per level { level_eval (this rank's tiles -> vals[e]) -> [sum-reduce / gather vals over the ranks] -> level_commit }
//         -> grad_eval (this rank's tiles of the voxel-ordered gradient list -> ngrads[e][3]) -> [reduce] -> grad_commit
// Everything is stream-ordered; no call reads a count back: the lists are voxel-ordered on every rank (classify_flag_kernel +
// ordered compaction), their lengths stay in device counters, the exchange buffers have a caller-chosen fixed capacity (a level
// longer than that is cut — surfd_grid_get_stats shows the true counts afterwards, compare them with the capacity).
static int refine_level_ordered(surfd_grid *g, int level, float *udf, hipStream_t st) {
    if (g->levels[level] >= g->N) return SURFD_OK;
    const int n = g->levels[level];
    const long n3 = (long)n * n * n;
    if (!g->flags[level]) HIP_TRY(hipMalloc((void **)&g->flags[level], (size_t)n3));
    LatticeGeom G{n, ilog2(n), g->N, g->N / n};
    hipLaunchKernelGGL(classify_flag_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(n3, 256), 4096)), dim3(256), 0, st, G,
                       (const unsigned char *)(level > 0 ? g->flags[level - 1] : nullptr), (const float *)udf, g->refine[level],
                       g->flags[level], g->far_list, g->counters + CTR_FAR + level);
    LAUNCH_CHECK();
    int rc = ordered_select(FlagSet{g->flags[level]}, n3, (int *)g->sel_tmp, g->parents[level + 1], g->counters + CTR_PARENT + level + 1, st,
                            LatticeToVoxel{G});
    if (rc) return rc;
    const int s = g->N / n;
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, st, (const int *)g->far_list, (const int *)(g->counters + CTR_FAR + level), s,
                       ilog2(s), g->N, udf, g->grad_thr, (int *)nullptr, g->counters + CTR_GRAD);
    LAUNCH_CHECK();
    return SURFD_OK;
}

int surfd_grid_shard_begin(surfd_grid *g, float *udf, float *grads, surfd_stream s) {
    const int rc = surfd_grid_begin(g, udf, grads, s);
    if (rc) return rc;
    g->shard_level = 0; g->shard_cap = 0; g->shard_evaluated = false;
    return SURFD_OK;
}

// The calls of one sharded fill come in a fixed order (every rank the same): level 0 eval (once per rank this process plays) ->
// commit, level 1 ..., then the gradient pair.  Anything else would read stale parents, flags or counters: SURFD_ERR_STATE.
static int shard_expect(surfd_grid *g, const char *fn, int level, int64_t capacity, bool commit) {
    if (g->shard_level < 0) SURFD_FAIL(SURFD_ERR_STATE, "%s: call surfd_grid_shard_begin first", fn);
    if (level != g->shard_level)
        SURFD_FAIL(SURFD_ERR_STATE, "%s: out of order — the open fill is at %s %d, the call names %d", fn,
                   g->shard_level >= g->n_levels ? "the gradient step after level" : "level", std::min(g->shard_level, g->n_levels - 1), level);
    if (commit && !g->shard_evaluated) SURFD_FAIL(SURFD_ERR_STATE, "%s: nothing has been evaluated for this step yet", fn);
    if (g->shard_evaluated && capacity != g->shard_cap)
        SURFD_FAIL(SURFD_ERR_STATE, "%s: capacity %lld differs from the %lld the step's evaluation used", fn, (long long)capacity, (long long)g->shard_cap);
    return SURFD_OK;
}

int surfd_grid_shard_level_eval(surfd_grid *g, surfd_decoder *d, int sample, int level, int rank, int world, float *vals,
                                int64_t capacity, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_shard_level_eval");
    if (rc) return rc;
    if (level < 0 || level >= g->n_levels || !vals || capacity < 1 || world < 1 || rank < 0 || rank >= world)
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_shard_level_eval: bad argument (level %d, rank %d of %d)", level, rank, world);
    if ((rc = shard_expect(g, "surfd_grid_shard_level_eval", level, capacity, false))) return rc;
    PtIO io = eval_io(g, level);
    if (level == 0 && io.n > capacity)
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_shard_level_eval: level 0 has %ld lattice points, the exchange buffer's capacity is %lld", io.n, (long long)capacity);
    io.out_udf = vals; io.cap = capacity; io.shard_n = world; io.shard_i = rank;
    const long hint = level == 0 ? ceil_div<long>(ceil_div<long>(std::min<long>(io.n, capacity), 64), world) : -1;
    if ((rc = decoder_launch(d, sample, io, false, hint, as_stream(s)))) return rc;
    g->shard_cap = capacity; g->shard_evaluated = true;
    return SURFD_OK;
}

static int shard_check_layout(const char *fn, int64_t capacity, int world) {
    if (world < 1) SURFD_FAIL(SURFD_ERR_ARG, "%s: world must be >= 1", fn);
    if (world > 1 && capacity % (64 * (int64_t)world) != 0)
        SURFD_FAIL(SURFD_ERR_ARG, "%s: with %d ranks the capacity (%lld points) must be whole 64-point tiles of every rank", fn, world, (long long)capacity);
    return SURFD_OK;
}

// This rank's tiles of the step's point-indexed buffer (what surfd_grid_shard_level_eval / _grad_eval just wrote) -> its segment of
// capacity / world points: what the all-gather sends.  level = n_levels names the gradient step (3 floats per point).
int surfd_grid_shard_pack(surfd_grid *g, int level, int rank, int world, const float *vals, int64_t capacity, float *segment, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_shard_pack");
    if (rc) return rc;
    if (level < 0 || level > g->n_levels || !vals || !segment || capacity < 1 || rank < 0 || rank >= world)
        SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_shard_pack: bad argument (level %d, rank %d of %d)", level, rank, world);
    if ((rc = shard_check_layout("surfd_grid_shard_pack", capacity, world))) return rc;
    if ((rc = shard_expect(g, "surfd_grid_shard_pack", level, capacity, true))) return rc;
    hipStream_t st = as_stream(s);
    PtIO io;
    int width = 1;
    if (level == g->n_levels) {
        if (!g->cur_grads) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_shard_pack: the open fill has no gradient step");
        io = base_io(g);
        io.mode = PT_LIST; io.list = g->grad_list; io.count_dev = g->counters + CTR_GRAD;
        width = 3;
    } else {
        io = eval_io(g, level);
    }
    io.cap = capacity;
    const long seg = capacity / world;
    hipLaunchKernelGGL(shard_pack_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(seg, 256), 2048)), dim3(256), 0, st, io, rank, world, width, vals, segment);
    LAUNCH_CHECK();
    return SURFD_OK;
}

int surfd_grid_shard_level_commit(surfd_grid *g, int level, const float *vals, int64_t capacity, int world, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_shard_level_commit");
    if (rc) return rc;
    if (level < 0 || level >= g->n_levels || !vals || capacity < 1) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_shard_level_commit: bad argument");
    if ((rc = shard_check_layout("surfd_grid_shard_level_commit", capacity, world))) return rc;
    if ((rc = shard_expect(g, "surfd_grid_shard_level_commit", level, capacity, true))) return rc;
    hipStream_t st = as_stream(s);
    PtIO io = eval_io(g, level);
    io.grid_udf = g->cur_udf; io.cap = capacity;
    const long upper = level == 0 ? io.n : capacity;
    if (level > 0) {
        hipLaunchKernelGGL(shard_overflow_kernel, dim3(1), dim3(64), 0, st, (const int *)(g->counters + CTR_PARENT + level), 7, (long)capacity, g->overflow);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(commit_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(upper, 256), 4096)), dim3(256), 0, st, io, vals, world, (long)(capacity / world));
    LAUNCH_CHECK();
    if ((rc = refine_level_ordered(g, level, g->cur_udf, st))) return rc;
    g->shard_level = level + 1; g->shard_evaluated = false; g->shard_cap = 0;
    if (level == g->n_levels - 1 && !g->cur_grads) {      // a fill without gradients ends here (otherwise surfd_grid_shard_grad_commit does this)
        g->dense_last = false;
        g->shard_level = -1;
        return add_to_totals(g, 0, st);
    }
    return SURFD_OK;
}

int surfd_grid_shard_grad_eval(surfd_grid *g, surfd_decoder *d, int sample, int rank, int world, float *ngrads, int64_t capacity,
                               surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_shard_grad_eval");
    if (rc) return rc;
    if (!g->cur_udf || !g->cur_grads) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_shard_grad_eval: surfd_grid_shard_begin was not given a gradient buffer");
    if (!ngrads || capacity < 1 || world < 1 || rank < 0 || rank >= world) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_shard_grad_eval: bad argument");
    if ((rc = shard_expect(g, "surfd_grid_shard_grad_eval", g->n_levels, capacity, false))) return rc;
    hipStream_t st = as_stream(s);
    if (!g->shard_evaluated && (rc = compact_grad_list(g, g->cur_udf, g->grad_thr, st))) return rc;          // voxel order: the same list on every rank
    PtIO io = base_io(g);
    io.mode = PT_LIST; io.list = g->grad_list; io.count_dev = g->counters + CTR_GRAD;
    io.out_ngrad = ngrads; io.cap = capacity; io.shard_n = world; io.shard_i = rank;
    if ((rc = decoder_launch(d, sample, io, true, -1, st))) return rc;
    g->shard_cap = capacity; g->shard_evaluated = true;
    return SURFD_OK;
}

int surfd_grid_shard_grad_commit(surfd_grid *g, const float *ngrads, int64_t capacity, int world, surfd_stream s) {
    int rc = check_ready(g, "surfd_grid_shard_grad_commit");
    if (rc) return rc;
    if ((rc = shard_check_layout("surfd_grid_shard_grad_commit", capacity, world))) return rc;
    if (!g->cur_grads) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_shard_grad_commit: no gradient buffer was given to surfd_grid_shard_begin");
    if (!ngrads || capacity < 1) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_shard_grad_commit: bad argument");
    if ((rc = shard_expect(g, "surfd_grid_shard_grad_commit", g->n_levels, capacity, true))) return rc;
    hipStream_t st = as_stream(s);
    hipLaunchKernelGGL(shard_overflow_kernel, dim3(1), dim3(64), 0, st, (const int *)(g->counters + CTR_GRAD), 1, (long)capacity, g->overflow);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(grad_commit_dev_kernel, dim3((unsigned)std::min<long>(ceil_div<long>(capacity, 256), 4096)), dim3(256), 0, st,
                       (const int *)g->grad_list, (const int *)(g->counters + CTR_GRAD), (long)capacity, ngrads, g->cur_grads, world, (long)(capacity / world));
    LAUNCH_CHECK();
    g->dense_last = false;
    g->shard_level = -1;
    return add_to_totals(g, 0, st);
}

// Exchange buffers that were shorter than the list they carried (levels and gradient lists, summed over the sharded fills since
// the last reset): non-zero = at least one grid of that span is incomplete.  One 8-byte read; the counts behind it are in
// surfd_grid_get_stats of the fill in question.
int surfd_grid_shard_overflows(surfd_grid *g, int64_t *n, int reset, surfd_stream s) {
    if (!g || !n) SURFD_FAIL(SURFD_ERR_ARG, "surfd_grid_shard_overflows: null argument");
    if (!g->allocated) SURFD_FAIL(SURFD_ERR_STATE, "surfd_grid_shard_overflows: nothing has been filled yet");
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, g->overflow, sizeof(v), hipMemcpyDeviceToHost, as_stream(s)));
    if (reset) HIP_TRY(hipMemsetAsync(g->overflow, 0, sizeof(v), as_stream(s)));
    HIP_TRY(hipStreamSynchronize(as_stream(s)));
    *n = (int64_t)v;
    return SURFD_OK;
}

}  // extern "C"
