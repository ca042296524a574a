/*
 * surfd_hip.h — C ABI of libsurfd_hip.so: the MI355X (gfx950) implementation of Surf-D's
 * sampling hot path.  Plain pointers and sizes only; no torch types.
 *
 * The reference (Yzmblog/SurfD) has no FFI layer: its boundary is the set of Python call
 * signatures used by sample/generate_*.py.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference root); INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - every function returns 0 on success or a negative surfd_status; the message is
 *    available from surfd_last_error() (thread-local).  No exceptions, no abort().
 *  - all device buffers are owned by the caller (torch), contiguous, fp32 unless noted;
 *    the library owns only re-laid-out private weight copies and a workspace arena.
 *  - all work is enqueued on the caller's hipStream_t (passed as void*); functions do
 *    not synchronise unless documented ("host-sync").
 *  - handles are bound to the device current at create time; one handle per process/rank.
 */
#ifndef SURFD_HIP_H
#define SURFD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SURFD_OK = 0,
    SURFD_ERR_ARG = -1,        /* bad shape / null pointer / unknown key      */
    SURFD_ERR_STATE = -2,      /* missing parameter, latents not bound, ...   */
    SURFD_ERR_HIP = -3,        /* a HIP runtime call failed                   */
    SURFD_ERR_UNSUPPORTED = -4 /* configuration outside what the path covers  */
} surfd_status;

typedef void *surfd_stream;    /* hipStream_t */
typedef struct surfd_unet surfd_unet;
typedef struct surfd_decoder surfd_decoder;
typedef struct surfd_grid surfd_grid;

const char *surfd_last_error(void);
int surfd_abi_version(void);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int surfd_device_count(void);
/* Compile-time configuration of the library (no reference counterpart): "name=value" for every experiment macro of the kernel
 * sources, then "unsafe_variants=N" = how many of them select a variant recorded as wrong, not bit-stable, or a developer aid
 * (0 for the product build: such variants need -DSURFD_ALLOW_UNSAFE_VARIANTS to compile at all).  bench.py prints it as
 * config.build_flags; tests/test_abi_cpu.py asserts the shipped library was built with the defaults.  Static storage. */
const char *surfd_build_config(void);

/* Measurement aid (no reference counterpart): when enabled, the library brackets its dominant
 * kernels with HIP events on the stream they are launched on.  kind 0 = decoder forward kernel,
 * 1 = decoder forward+reverse kernel, 2 = whole surfd_sample_loop.  read is host-sync and clears. */
int surfd_profile_enable(int on);
int surfd_profile_read(int kind, int64_t *launches, double *total_ms);
/* developer aid: with SURFD_CONV_DEBUG=1 in the environment every conv launch records shader-clock
 * stamps of its phases (16 int64 per launch); host-sync read + reset, returns the launch count */
int surfd_unet_debug_read(surfd_unet *u, long long *out, int max_launches);

/* ------------------------------------------------------------------------------------ */
/* Denoiser: UNetModel (models/openaimodel.py:413-749) as configured by MDM             */
/* (models/mdm.py:34-57).                                                               */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int in_channels, model_channels, out_channels, num_res_blocks;
    int n_mult, channel_mult[8];
    int n_attn, attention_resolutions[8];
    int num_heads;
    int context_dim;   /* 0: no sketch_emb */
    int num_classes;   /* 0: no label_emb  */
} surfd_unet_cfg;

/* Builds the execution plan on the host (no device needed).  Replaces
 * UNetModel.__init__ (openaimodel.py:443-692). */
int surfd_unet_create(const surfd_unet_cfg *cfg, surfd_unet **out);
void surfd_unet_destroy(surfd_unet *u);
/* state_dict layout the handle expects (keys relative to "Unet.", reference order):
 * lets the host build its nn.Module / check a checkpoint without a device. */
int surfd_unet_num_params(const surfd_unet *u);
int surfd_unet_param_info(const surfd_unet *u, int i, const char **key, int64_t shape[4], int *ndim);
/* One call per state_dict tensor (load_model_wo_clip, utils/model_util.py:6-9).
 * dev_ptr: device fp32, contiguous; repacked into the private MFMA fragment layout. */
int surfd_unet_set_param(surfd_unet *u, const char *key, const void *dev_ptr,
                         const int64_t *shape, int ndim, surfd_stream s);
/* fails with SURFD_ERR_STATE naming the first tensor that was never set */
int surfd_unet_finalize(surfd_unet *u, surfd_stream s);
/* UNetModel.forward (openaimodel.py:710-749) / MDM.forward (mdm.py:91-110):
 * x[B,1,L], t[B] (int64, original-scale timesteps), ctx[B,context_dim] or NULL,
 * cls[B] (int64) or NULL -> out[B,1,L].  All device pointers. */
int surfd_unet_forward(surfd_unet *u, const float *x, const int64_t *t, const float *ctx,
                       const int64_t *cls, float *out, int B, int L, surfd_stream s);

/* Arithmetic of the denoiser's convolutions (every Conv1d of openaimodel.py ResBlock / AttentionBlock /
 * Downsample / Upsample / head; the embedding Linears always run in fp32):
 * 1 = "f16x2" (default): operands split into two fp16 terms, three products on the fp16 matrix pipe, fp32
 *     accumulation (same scheme as surfd_decoder_set_precision); operands outside +-65504 are clamped and
 *     counted (surfd_unet_saturation_count).  0 = "fp32": exact v_mfma_f32_32x32x2_f32, no range limit.
 * The initial mode can also be set with SURFD_UNET_PRECISION=fp32|f16x2. */
int surfd_unet_set_precision(surfd_unet *u, int mode);
/* How many CUs this handle's launches can count on (default 256 = the whole chip).  The conv kernel splits the
 * contraction of small layers over extra workgroups until about two per CU are in flight; a handle that shares the
 * chip (several loops next to the decoder, surfd_amd.parallel.BatchPipeline) is told its share so that it does not
 * pay the split's redundant operand staging for parallelism it cannot get.  Results for different budgets differ in
 * fp32 summation order only.  No reference counterpart. */
int surfd_unet_set_cu_budget(surfd_unet *u, int cus);
/* Work decomposition of the f16x2 conv kernel.  design_batch = 0 (default): latency form — a workgroup owns one
 * 32-row tile of a layer's output and the K split follows the batch at hand: fastest for ONE narrow loop alone on
 * the chip.  design_batch > 0: wide form for loops over tens of latents (reference: the `batch_size` of
 * sample/generate_*.py is the only batching the reference has) — a workgroup stages its GroupNorm/SiLU operand once
 * for FOUR row tiles, and the K split is fixed per layer for a batch of `design_batch`, so a latent's result is
 * bit-identical whatever batch width it rides in (tests/test_gpu_unet.py).  Results of the two forms differ in fp32
 * summation order only.  No reference counterpart. */
int surfd_unet_set_wide(surfd_unet *u, int design_batch);
/* host-sync: number of workgroups of the f16x2 conv kernel that had to clamp an operand to the fp16 range since the
 * last reset (0 = every evaluation so far was inside the range the mode is exact for) */
int surfd_unet_saturation_count(surfd_unet *u, int reset, int64_t *count, surfd_stream s);
/* Iterations the handle's fused reverse loop (surfd_sample_loop) has finished so far — the device-side loop counter, read over a
 * private stream beside the one the loop runs on, so a host thread can draw the reference's progress bar
 * (diffusion/gaussian_diffusion.py:677-681, `progress=True` in every sample/generate_*.py) while the loop is in flight.
 * *iteration = -1 before the first loop. */
int surfd_unet_loop_progress(surfd_unet *u, int *iteration);
/* developer aid: op >= 0 restricts the f16x2 kernel to that one conv op of the plan (the others run fp32); -1 lifts it */
int surfd_unet_debug_only_op(surfd_unet *u, int op);
/* test tap: runs only the ops of ONE module of UNetModel ("input_blocks.1.0" ResBlock, "input_blocks.1.1" AttentionBlock,
 * "input_blocks.3.0" Downsample, "out" head, ...) on in[B,Cin,Lin] -> out[B,Cout,Lout], with the embedding rows left by
 * the preceding surfd_unet_forward (same t, B, L): lets the tests compare single modules with the reference's hooks */
int surfd_unet_debug_run_module(surfd_unet *u, const char *module, const float *in, int Cin, int Lin,
                                float *out, int Cout, int Lout, int B, int L, surfd_stream s);

/* ------------------------------------------------------------------------------------ */
/* Reverse loop: p_sample_loop / ddim_sample_loop                                       */
/* (diffusion/gaussian_diffusion.py:570-708, 858-972; diffusion/respace.py:63-132).      */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int sampler;            /* 0 = DDPM ancestral (p_sample :471-520), 1 = DDIM (:711-761) */
    int num_steps;          /* T' (after respacing)                                         */
    int clip_denoised;      /* clamp x0 to [-1,1] (process_xstart :330-336)                 */
    float eta;              /* DDIM only                                                    */
    const int64_t *timestep_map;   /* host [T']: loop index -> original timestep (respace.py:123-128) */
    /* host float32 tables [T'], already cast float64->float32 as _extract_into_tensor does (:1339) */
    const float *coef1, *coef2, *log_variance;                /* DDPM */
    const float *sqrt_recip_ab, *sqrt_recipm1_ab, *ab, *ab_prev;   /* DDIM */
} surfd_sampler_cfg;

/* Whole reverse loop without returning to the host: noise[T'+1,B,1,L] (row 0 = x_T,
 * row 1+k = z of loop iteration k), ctx/cls as in surfd_unet_forward (constant over the
 * loop), x_out[B,1,L]; traj (nullable) [T',B,1,L] receives x after every iteration.
 * One iteration (~117 kernel nodes) is captured once into a hipGraph and replayed T' times on the
 * caller's stream.  Host-sync once at entry (schedule tables are uploaded and the stream is quiesced
 * before capture); the T' replays themselves are asynchronous. */
int surfd_sample_loop(surfd_unet *u, const surfd_sampler_cfg *cfg, const float *noise,
                      const float *ctx, const int64_t *cls, float *x_out, float *traj,
                      int B, int L, surfd_stream s);
/* The same loop in three calls, for a host thread that drives several loops (one surfd_unet handle and one stream each) in
 * turns: begin = the host-synchronous part (embedding rows, coefficient table, state <- noise row 0, the captured iteration),
 * run = up to `iterations` more graph replays on s (asynchronous; *remaining, nullable, = replays still to launch),
 * end = x_out <- state once every iteration has been launched (SURFD_ERR_STATE before that).  begin + run(T') + end IS
 * surfd_sample_loop; noise / ctx / cls / traj must stay alive until the stream has passed end.  One loop per handle at a time
 * (a second begin on the same handle abandons the first; several loops = several handles, all three calls of a handle from one
 * host thread or externally serialised); a begin that fails leaves no loop open.  The loop they spell is
 * p_sample_loop_progressive's for-loop (diffusion/gaussian_diffusion.py:682-708) cut at iteration boundaries. */
int surfd_sample_loop_begin(surfd_unet *u, const surfd_sampler_cfg *cfg, const float *noise,
                            const float *ctx, const int64_t *cls, float *traj, int B, int L,
                            surfd_stream s);
int surfd_sample_loop_run(surfd_unet *u, int iterations, int *remaining, surfd_stream s);
int surfd_sample_loop_end(surfd_unet *u, float *x_out, surfd_stream s);
/* single posterior updates on n elements (used by the generic Python loop) */
int surfd_ddpm_step(const float *x_t, const float *x0, const float *z, float coef1, float coef2,
                    float log_variance, int t_nonzero, int clip_denoised, float *out, int64_t n,
                    surfd_stream s);
int surfd_ddim_step(const float *x_t, const float *x0, const float *z, float sqrt_recip_ab,
                    float sqrt_recipm1_ab, float ab, float ab_prev, float eta, int t_nonzero,
                    int clip_denoised, float *out, int64_t n, surfd_stream s);

/* ------------------------------------------------------------------------------------ */
/* UDF field: CoordsEncoder.encode (AutoEncoder/models/coordsenc.py:25-51) +             */
/* CbnDecoder.forward (AutoEncoder/models/cbndec.py:35-47,127-134) + the udf_func        */
/* closure (sample/generate_uncond.py:96-101) + sample_grads (meshudf/meshudf.py:231-251) */
/* ------------------------------------------------------------------------------------ */
int surfd_decoder_create(int input_dim, int latent_dim, int hidden_dim, int num_blocks,
                         surfd_decoder **out);
void surfd_decoder_destroy(surfd_decoder *d);
int surfd_decoder_num_params(const surfd_decoder *d);
int surfd_decoder_param_info(const surfd_decoder *d, int i, const char **key, int64_t shape[4], int *ndim);
/* keys exactly as in ckpt["decoder"] ("decoder.fc_p.weight", ...); num_batches_tracked is
 * accepted and ignored (dev_ptr may be NULL for it). */
int surfd_decoder_set_param(surfd_decoder *d, const char *key, const void *dev_ptr,
                            const int64_t *shape, int ndim, surfd_stream s);
int surfd_decoder_finalize(surfd_decoder *d, surfd_stream s);
/* Arithmetic of the decoder kernels (udf / logits / grid fill, and the forward + reverse sweep of
 * surfd_decoder_udf_grad):
 * 1 = "f16x2" (default): every fp32 operand is split into two fp16 terms (weights pre-scaled by one power
 *     of two), the three significant products are accumulated in fp32 on the fp16 matrix pipe.  Error
 *     against an fp64 evaluation is the same size as the plain fp32 kernel's (tests/test_gpu_decoder_grid.py);
 *     activations saturate at 65504.  2.85x the throughput of mode 0 on MI355X.
 * 0 = "fp32": v_mfma_f32_32x32x2_f32, bitwise an fmaf chain, no range limit.
 * In mode 1 the reverse sweep scales the adjoint of each 64-point tile by one power of two before the fp16 split
 * (exact, divided out afterwards; no range limit): directions agree with mode 0 to the golden tolerance, but a point's
 * last bits can depend on the points it shares a tile with; mode 0 is bitwise independent of the tiling.
 * The initial mode can also be set with SURFD_DECODER_PRECISION=fp32|f16x2. */
int surfd_decoder_set_precision(surfd_decoder *d, int mode);
/* host-sync: waves of the f16x2 forward kernel that produced an activation beyond +-65504 (clamped) since the last
 * reset.  Non-zero = this checkpoint / latent leaves the range mode 1 is exact for: switch to mode 0. */
int surfd_decoder_saturation_count(surfd_decoder *d, int reset, int64_t *count, surfd_stream s);
/* Shader clock (GHz) the chip sustained under the forward decoder kernel since the last reset: workgroup 0 of every launch of the
 * 8-wave forward kernel adds its shader-cycle and 100 MHz real-time differences to a device-side record (the kernel is
 * power-bound: its rate follows this clock, bench.py `roofline.sustained_clock_ghz`).  0 when no launch ran.  Measurement aid,
 * no reference counterpart. */
int surfd_decoder_sustained_clock(surfd_decoder *d, int reset, double *ghz, surfd_stream s);
/* The decoder kernels are persistent: `blocks` workgroups (one per CU, LDS-limited) loop over the point tiles.
 * 0 (default) = every CU.  A smaller value leaves CUs free for work on another stream (bench.py overlaps the
 * reverse loop of the next batch with the grid evaluation of the current one this way). */
int surfd_decoder_set_grid_blocks(surfd_decoder *d, int blocks);
/* lat[S,D]: computes the per-sample conditional-BN scale/shift tables [S,11,2,H]
 * (the 22 per-point Conv1d(D->H) of cbndec.py:74-79 collapse to this when one latent is
 * broadcast to all points, cbndec.py:131-132). */
int surfd_decoder_bind_latents(surfd_decoder *d, const float *lat, int S, surfd_stream s);
/* CbnDecoder.forward on pre-encoded coordinates emb[n,input_dim] -> logits[n] */
int surfd_decoder_logits_emb(surfd_decoder *d, int sample, const float *emb, int64_t n,
                             float *logits, surfd_stream s);
/* udf_func: pts[n,3] -> udf[n] = (1 - sigmoid(decoder(encode(p), lat))) * 0.1;
 * logits (nullable) receives the raw decoder output */
int surfd_decoder_udf(surfd_decoder *d, int sample, const float *pts, int64_t n, float *udf,
                      float *logits, surfd_stream s);
/* sample_grads: ngrad[n,3] = -normalize(d udf/d p) (eps 1e-12; exact zero vector where the
 * fp32 sigmoid derivative vanishes); udf (nullable) as above; dlogit (nullable) receives the
 * raw d logit / d p [n,3] (what an autograd backward through CbnDecoder.forward needs).
 * At least one of ngrad / dlogit must be given. */
int surfd_decoder_udf_grad(surfd_decoder *d, int sample, const float *pts, int64_t n, float *udf,
                           float *ngrad, float *dlogit, surfd_stream s);

/* ------------------------------------------------------------------------------------ */
/* UDF grid: GridFiller / get_udf_and_grads (meshudf/meshudf.py:23-304)                  */
/* ------------------------------------------------------------------------------------ */
#define SURFD_GRID_MAX_LEVELS 8
typedef struct {
    int n_levels;
    int levels[SURFD_GRID_MAX_LEVELS];
    int64_t fwd_points[SURFD_GRID_MAX_LEVELS];   /* decoder forward queries per level */
    int64_t grad_points;                         /* forward+backward queries          */
} surfd_grid_stats;

/* N = final resolution (power of two >= 64); allocates the per-level work lists. */
int surfd_grid_create(int N, surfd_grid **out);
void surfd_grid_destroy(surfd_grid *g);
/* thresholds are computed by the host exactly as the reference's Python does and handed
 * over as float32: refine[l] = float32(1.5*1.7*(2.0/levels[l])) (meshudf.py:185-188),
 * grad = float32(2.5*2.0/N) (:200), voxel = float32(2.0/(N-1)) (:53), origin -1. */
int surfd_grid_set_thresholds(surfd_grid *g, const float *refine, int n_levels, float grad_thr,
                              float voxel, float origin);
/* GridFiller.fill_grid fused with the native decoder: no host round trip, no Python.
 * udf[N^3], grads[N^3*3] device outputs (grads may be NULL: watertight variant,
 * utils/utils.py:151-339). */
int surfd_grid_fill(surfd_grid *g, surfd_decoder *d, int sample, float *udf, float *grads,
                    surfd_stream s);
/* The same for the grids of n <= 8 shapes at once (what the sample scripts do shape after shape, generate_uncond.py:
 * 91-123): one grid handle, bound-latent index, udf and grads (entries or the array may be NULL) pointer per shape.
 * Every refinement level of all shapes is ONE launch of the persistent decoder kernel; values are bit-identical to n
 * calls of surfd_grid_fill.  No reference counterpart (throughput form). */
int surfd_grid_fill_batch(surfd_grid *const *grids, int n, surfd_decoder *d, const int *samples, float *const *udf,
                          float *const *grads, surfd_stream s);
/* get_udf_and_grads (use_fast_grid_filler=False): all N^3 points, gradients where
 * udf < grad_below (= max_dist - 1e-3 in the reference). */
int surfd_grid_fill_dense(surfd_grid *g, surfd_decoder *d, int sample, float grad_below,
                          float *udf, float *grads, surfd_stream s);
/* host-sync: counters of the last fill on this handle */
int surfd_grid_get_stats(surfd_grid *g, surfd_grid_stats *out, surfd_stream s);
/* host-sync: running totals over every fused fill (surfd_grid_fill / _fill_batch / _fill_dense) on this handle since
 * the last reset — decoder forward queries per level, forward+backward queries, number of fills.  The totals are kept
 * on the device by the fills themselves, so a throughput run can account for every query of hundreds of different
 * shapes without a host read-back per shape (the reference prints nothing of the kind; bench.py's roofline uses it). */
int surfd_grid_get_totals(surfd_grid *g, surfd_grid_stats *out, int64_t *fills, int reset, surfd_stream s);

/* Same algorithm with an arbitrary host callable (the reference's udf_func contract):
 * begin -> for each level { points -> [host evaluates] -> commit } -> grad_points ->
 * [host differentiates] -> grad_commit.  *_points are host-sync (they return a count). */
int surfd_grid_begin(surfd_grid *g, float *udf, float *grads, surfd_stream s);
int surfd_grid_level_points(surfd_grid *g, int level, float *xyz, int64_t capacity, int64_t *n,
                            surfd_stream s);
int surfd_grid_level_commit(surfd_grid *g, int level, const float *values, int64_t n, surfd_stream s);
int surfd_grid_grad_points(surfd_grid *g, float *xyz, int64_t capacity, int64_t *n, surfd_stream s);
int surfd_grid_grad_commit(surfd_grid *g, const float *ngrads, int64_t n, surfd_stream s);

/* Grid-shard mode (no reference counterpart: the reference fills one grid on one device, meshudf/meshudf.py:123-206; this is the
 * north star's "shard the per-sample 512^3 grid evaluation across the GPUs").  Every one of `world` ranks runs, on its own device,
 *   shard_begin -> per level { shard_level_eval -> shard_pack -> [ncclAllGather of the ranks' segments] -> shard_level_commit }
 *               -> shard_grad_eval -> shard_pack -> [all-gather] -> shard_grad_commit
 * with the native decoder: rank r evaluates the 64-point tiles r, r + world, ... of each level's VOXEL-ORDERED point list (the
 * same list on every rank) into vals[point number]; shard_pack compacts exactly those tiles into the rank's SEGMENT of
 * capacity / world points (tile t of the list = tile t / world of rank t % world's segment), the segments are all-gathered into
 * [world][capacity / world] (an all-gather of compact slices — half the bytes of summing zero-filled
 * point-indexed buffers, and nothing to zero), and shard_level_commit(world) reads point e from the gathered layout.  With
 * world = 1 there is no pack and no exchange: commit reads the point-indexed buffer.  All calls are stream-ordered and none reads a count back: list lengths stay on the device, `capacity` (points) bounds what a level may hold —
 * a list longer than its buffer is cut and COUNTED on the device (surfd_grid_shard_overflows; the counts themselves are in
 * surfd_grid_get_stats).  The handle tracks the protocol: a call that names another level than the open fill is at, a commit
 * without an evaluation, or a commit with another capacity than its evaluation returns SURFD_ERR_STATE.
 * With world = 1 the result equals surfd_grid_fill bit for bit. */
int surfd_grid_shard_begin(surfd_grid *g, float *udf, float *grads, surfd_stream s);
int surfd_grid_shard_level_eval(surfd_grid *g, surfd_decoder *d, int sample, int level, int rank, int world, float *vals,
                                int64_t capacity, surfd_stream s);
/* this rank's tiles of the step's point-indexed buffer -> its segment [capacity / world] (x 3 floats for the gradient step, named
 * by level = number of levels); capacity must be whole 64-point tiles of every rank */
int surfd_grid_shard_pack(surfd_grid *g, int level, int rank, int world, const float *vals, int64_t capacity, float *segment,
                          surfd_stream s);
int surfd_grid_shard_level_commit(surfd_grid *g, int level, const float *vals, int64_t capacity, int world, surfd_stream s);
int surfd_grid_shard_grad_eval(surfd_grid *g, surfd_decoder *d, int sample, int rank, int world, float *ngrads, int64_t capacity,
                               surfd_stream s);
int surfd_grid_shard_grad_commit(surfd_grid *g, const float *ngrads, int64_t capacity, int world, surfd_stream s);
/* Exchange buffers that were shorter than the list they carried, over the sharded fills since the last reset (levels and gradient
 * lists; 0 = every grid of that span is complete).  Synchronises the stream (one 8-byte read).  No reference counterpart. */
int surfd_grid_shard_overflows(surfd_grid *g, int64_t *n, int reset, surfd_stream s);

/* ------------------------------------------------------------------------------------ */
/* UDF marching cubes (host side, no device): udf_mc_lewiner / marching_cubes_udf        */
/* (meshudf/_marching_cubes_lewiner.py:87-154, meshudf/_marching_cubes_lewiner_cy.pyx:1115-1775) */
/* ------------------------------------------------------------------------------------ */
typedef struct surfd_mc surfd_mc;
/* udf[nz,ny,nx] and grads[nz,ny,nx,3] are HOST arrays (the grids of surfd_grid_fill copied back, udf clamped at 0).
 * Meshes the zero set: signs of the cube corners are voted from the gradient field, cubes are triangulated with
 * Lewiner et al.'s case tables.  Output bit-identical to the reference extension.  One shape per call, one thread. */
int surfd_mc_udf(const float *udf, const float *grads, int nz, int ny, int nx, int step, surfd_mc **out);
/* Level-set marching cubes over the whole volume (watertight path: sample/generate_text.py:139-141 extracts the 0.01
 * level with PyMCubes).  classic != 0: the original 256-case triangle table (PyMCubes' algorithm); 0: Lewiner's
 * disambiguated cases.  Same result accessors as surfd_mc_udf. */
int surfd_mc_iso(const float *volume, int nz, int ny, int nx, double level, int classic, int step, surfd_mc **out);
/* Sparse hand-off (reference meshudf/meshudf.py:344-349 copies the whole grid and gradient volume to the
 * host, _marching_cubes_lewiner_cy.pyx:1131,1157-1158 then looks only at cubes whose corners are all <= 1.74 voxel):
 *   device: surfd_band_compact  — voxel index, value (clamped at 0 as meshudf.py:338), gradient of every voxel with
 *           udf <= max_thr, in voxel order, into the handle's device buffers (stream-ordered, no host sync)
 *   host:   surfd_band_fetch    — host-sync on `copy_stream` only: count, then the band, into pinned memory
 *           surfd_mc_udf_band   — scatters the band into a host scratch volume that is "far" everywhere else, runs the
 *           same mesher, takes the band out again: the mesh of surfd_mc_udf on the dense volumes, bit for bit, at a cost
 *           proportional to the band. */
typedef struct surfd_band surfd_band;
typedef struct surfd_mc_scratch surfd_mc_scratch;
int surfd_band_create(int N, int64_t capacity, surfd_band **out);
void surfd_band_destroy(surfd_band *b);
int surfd_band_compact(surfd_band *b, const float *udf, const float *grads, float max_thr, surfd_stream s);
int surfd_band_fetch(surfd_band *b, surfd_stream copy_stream, int64_t *count, const int32_t **index, const float **packed);
int surfd_mc_band_threshold(int n, float *max_thr);          /* float(1.74 * 2 / (n - 1)), as the mesher compares */
int surfd_mc_scratch_create(int n, surfd_mc_scratch **out);  /* host: n^3 x (4 + 12 + 1) bytes, zero pages until touched */
void surfd_mc_scratch_destroy(surfd_mc_scratch *sc);
int surfd_mc_udf_band(surfd_mc_scratch *sc, const int32_t *index, const float *packed /* [count][4]: udf, gx, gy, gz */,
                      int64_t count, int step, surfd_mc **out);
int64_t surfd_mc_num_vertices(const surfd_mc *m);
int64_t surfd_mc_num_faces(const surfd_mc *m);
/* vertices[V,3] in (z,y,x) voxel units, faces[F,3] (winding of gradient_direction="descent"), normals[V,3] (unit),
 * values[V]; any pointer may be NULL */
int surfd_mc_copy(const surfd_mc *m, float *vertices, int32_t *faces, float *normals, float *values);
void surfd_mc_destroy(surfd_mc *m);
/* Wavefront OBJ writer for the meshes above ("v x y z" with 6 decimals, 1-based "f a b c"): the export step of the sample
 * scripts (sample/generate_uncond.py:113-122).  vertices[nv,3] float64, faces[nf,3] int64, host arrays. */
int surfd_write_obj(const char *path, const double *vertices, int64_t nv, const int64_t *faces, int64_t nf);
/* the case tables the library was built with (Lewiner et al. 2003), for tests */
int surfd_mc_lut_count(void);
int surfd_mc_lut(int i, const char **name, const signed char **values, int *ndim, int dims[3]);

/* ------------------------------------------------------------------------------------ */
/* CrossAttention over [b, n, c] tokens (modules/attention.py:152-193) — SURVEY §8 a19.  */
/* No Surf-D configuration instantiates the module (use_spatial_transformer is False);    */
/* a standalone op with its own handle, fp32 on the matrix pipe.                           */
/* ------------------------------------------------------------------------------------ */
typedef struct surfd_xattn surfd_xattn;
/* CrossAttention.__init__(query_dim, context_dim=None, heads=8, dim_head=64) (:153-169); context_dim <= 0 means
 * "same as query_dim".  dim_head <= 128. */
int surfd_xattn_create(int query_dim, int context_dim, int heads, int dim_head, surfd_xattn **out);
void surfd_xattn_destroy(surfd_xattn *a);
/* state_dict entries of the reference module, fp32 device pointers: "to_q.weight" [heads*dim_head, query_dim],
 * "to_k.weight" / "to_v.weight" [heads*dim_head, context_dim], "to_out.0.weight" [query_dim, heads*dim_head],
 * "to_out.0.bias" [query_dim] */
int surfd_xattn_set_param(surfd_xattn *a, const char *name, const float *src, const int64_t *shape, int ndim, surfd_stream s);
/* CrossAttention.forward(x, context=None, mask=None) (:171-193): x [B, N, query_dim]; context [B, M, context_dim] or
 * NULL (self-attention, M ignored); mask [B, M] bytes, non-zero = attend, or NULL (masked scores are set to
 * -FLT_MAX exactly as masked_fill_ does, so a fully masked row attends uniformly); out [B, N, query_dim].
 * Dropout (p = 0 in every constructor call of the reference) is the identity. */
int surfd_xattn_forward(surfd_xattn *a, const float *x, const float *context, const unsigned char *mask, float *out,
                        int B, int N, int M, surfd_stream s);

#ifdef __cplusplus
}
#endif
#endif /* SURFD_HIP_H */
