// Execution plan and handle of the denoiser, shared by unet.hip (plan builder, fp32 conv kernel, attention)
// and conv_f16x2.hip (split-fp16 conv kernel).
#pragma once
#include "common.h"
#include "unet_api.h"

namespace surfd {

struct ParamInfo {
    std::string key;
    std::vector<int64_t> shape;
    bool is_set = false;
};

struct View { int buf = -1; int choff = 0; };   // buf: index into buffers, -2 = external input, -3 = external output

struct BufInfo { int C; int ds; };               // [B][C][L/ds]

struct SegPlan {
    View src;
    int C = 0, taps = 1, stride = 1, ups = 0, gn = 0, act = 0;
    std::string wkey;      // weight tensor packed into this segment
    std::string gnkey;     // "<prefix>" of GroupNorm weight/bias
    int ds = 1;            // source ds (ds == 0: length-1 "linear" operand)
};

struct ConvPlan {
    SegPlan seg[2];
    int nseg = 1;
    int Cout = 0;
    int ds_out = 1;
    std::vector<std::string> bias_keys;   // summed
    int emb_off = -1;
    View res, dst;
    // resolved at finalize
    size_t w_off = 0; int KGtot = 0; int kg_off[2] = {0, 0}; size_t bias_off = 0; int gn_off[2] = {-1, -1};
    // split-fp16 ("f16x2") layout of the same weights (conv_f16x2.hip); f16_ok = 0: layer runs on the fp32 kernel
    int f16_ok = 0;
    int blk[2] = {0, 0}, blkp[2] = {0, 0}, nblk[2] = {0, 0}, k16_off[2] = {0, 0};   // per segment: channels per K block, padded to 16, blocks, first k16 step
    int KS16 = 0;              // k16 steps per 32-row tile (all segments)
    size_t whf_off = 0;        // halfs
    int sc_idx = -1;           // slot in surfd_unet::wsc ({SC, 1/SC, max|W| bits, pad})
    int id = -1;               // index among the conv ops of the denoiser body (debugging aid)
    // second f16x2 layout of the same weights: K blocks of <= 128 channels (whole GroupNorm groups) for the two-column-tile
    // form of the wide kernel (conv_f16x2.hip, NT2); f16_ok2 = 0: the layer has no such blocking
    int f16_ok2 = 0;
    int blk2[2] = {0, 0}, blkp2[2] = {0, 0}, nblk2[2] = {0, 0}, k16_off2[2] = {0, 0};
    int KS16_2 = 0;
    size_t whf2_off = 0;
    // what the last launch of this convolution on the f16x2 kernel looked like (launch_conv2): the launch BEFORE it uses the
    // record to request this layer's weight slices into the L2 of the XCD that will read them (conv_f16x2.hip, SURFD_C2_PFN)
    struct LaunchRec {
        long gen = -1; int B = 0, L = 0;
        const _Float16 *whf = nullptr;
        int ntiles = 0, KS16 = 0, tpg = 1, KS = 1, G = 1, nblk0 = 0, it0 = 0, nch = 0, it1 = 0;
    };
    mutable LaunchRec rec;
};

struct AttnPlan { View qkv, out; int C = 0, ds = 1; };

struct Op { int kind; ConvPlan conv; AttnPlan attn; };   // 0 conv, 1 attn

}  // namespace surfd

struct surfd_unet {
    using ParamInfo = surfd::ParamInfo; using BufInfo = surfd::BufInfo; using Op = surfd::Op; using ConvPlan = surfd::ConvPlan;
    surfd_unet_cfg cfg;
    int ted = 0;                                  // time-embed dim
    std::vector<ParamInfo> params;
    std::map<std::string, int> pindex;
    std::vector<BufInfo> bufs;
    std::vector<Op> ops;                          // denoiser body, in execution order
    ConvPlan lin1, lin2, lin3;                    // embedding path
    int emb_total = 0;                            // sum of ResBlock Cout (14112)
    std::vector<std::pair<std::string, int>> emb_layers;   // (prefix, Cout) in table order
    // device state
    bool allocated = false, finalized = false;
    int device = -1;
    float *wpack = nullptr; size_t wpack_floats = 0;
    float *vecs = nullptr; size_t vec_floats = 0;
    std::map<std::string, size_t> vec_off;        // raw vectors (GN gamma/beta, biases) by key
    float *label_table = nullptr;
    // workspace (grow-only)
    std::vector<float *> buf_ptr; int ws_B = 0, ws_L = 0;
    float *temb = nullptr, *h1 = nullptr, *emb = nullptr, *emb_table = nullptr; int emb_rows_cap = 0; int emb_rows = 0, emb_B = 0;
    int emb_shared = 0;                                // 1: one embedding row per loop step, shared by all samples (no context / labels)
    // conditioned loops (context / labels): nothing scales with steps x samples.  time_embed(t) for every step [T][ted] (u->emb),
    // the per-sample part sketch_emb(ctx) + label_emb[cls] [B][ted] (emb_ctx) once per loop, and INSIDE the captured graph one
    // Linear launch per iteration: emb_table[B][14112] = emb_layers(SiLU(time part[k] + sample part[b]))
    int emb_ingraph = 0;
    float *emb_ctx = nullptr; int emb_ctx_cap = 0;
    int64_t *t_dev = nullptr; int t_cap = 0;
    float *part = nullptr; size_t part_floats = 0;     // split-K partial tiles
    long long *dbg = nullptr; int dbg_launch = 0;      // SURFD_CONV_DEBUG=1: per-launch phase stamps
    surfd::LoopState loop;
    int *counters = nullptr;
    // f16x2 conv path (conv_f16x2.hip)
    _Float16 *whf = nullptr; size_t whf_halfs = 0;     // split-fp16 weight planes, fragment-major for the 32x32x16 MFMA
    _Float16 *whf2 = nullptr; size_t whf2_halfs = 0;   // the same weights in K blocks of <= 128 channels (two-column-tile form)
    float *wsc = nullptr; int n_sc = 0;                // per-layer power-of-two weight scale
    std::vector<float> wsc_host;                       // host copy of wsc (the conv kernel takes 1/SC by value)
    unsigned *sat = nullptr;                           // device counter: workgroups that clamped an operand to the fp16 range
    int wide_batch = 0;                                // > 0: wide form of the f16x2 conv kernel, K split designed for this batch (surfd_unet_set_wide)
    int cu_budget = 256;                               // CUs this context's launches can count on (split-K sizing)
    int precision = 1;                                 // denoiser conv arithmetic: 1 = f16x2 (default), 0 = exact fp32 MFMA
    int dbg_only = -1;                                 // >= 0: only this conv op runs on the f16x2 kernel (surfd_unet_debug_only_op)
    const surfd::ConvPlan *pf_next = nullptr;          // the convolution that runs after the one being launched (weight prefetch ahead)
    long ws_gen = 1;                                   // bumped whenever a buffer baked into the cached loop graph is reallocated
};


namespace surfd {
// conv_f16x2.hip
int conv2_plan_layout(surfd_unet *u);                       // host: K blocking of every layer (after the fp32 arena is laid out)
int conv2_finalize(surfd_unet *u, hipStream_t st);          // device: scales + fp16 planes from the fp32 packs
int conv2_set_attributes();                                 // dynamic-LDS limits of the kernel instantiations
// Launches planned convolution `c` on the f16x2 kernel; returns 1 if the layer/shape is not covered (caller falls
// back to the fp32 kernel), 0 when launched, < 0 on error.
struct ConvLaunchIO {
    const float *ext_in; long ext_in_bs;     // external operand (View.buf == -2)
    float *ext_out; long ext_out_bs;         // external result  (View.buf == -3)
    const float *emb; long emb_bs;           // embedding rows of this evaluation (nullable)
    const int *step_ptr;                     // device loop counter (nullable)
    const LoopFuse *lf = nullptr;            // DEVICE pointer: posterior update in the epilogue of the head convolution (nullable)
    bool *lf_done = nullptr;                 // set when the launch took it over
};
int launch_conv2(surfd_unet *u, const ConvPlan &c, int B, int L, const ConvLaunchIO &io, hipStream_t st);
}  // namespace surfd
