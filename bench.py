#!/usr/bin/env python3
"""Headline benchmark: shapes/sec end-to-end for unconditional Surf-D sampling.

One "step" = one pass of the hot path over one batch of synthetic shapes on every rank:
    noise -> 1000-step DDPM reverse loop over the latent denoiser (B shapes at once)
          -> per shape: coarse-to-fine UDF grid (N^3) + spatial gradient, resident in HBM
(end point E1 of SURVEY.md §8d).  Weights are the deterministic synthetic tensors of
surfd_amd.synth (no checkpoints exist offline); noise is seeded per global shape index, so a
shape's result does not depend on how shapes are sharded over ranks.

    python bench.py --gpus 1 --steps 6 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Batches are software-pipelined (--pipeline 1, default; surfd_amd.parallel.BatchPipeline): one reverse loop is a chain
of ~100 000 dependent, latency-bound launches that fills a fraction of the chip, so --loop-chains loops of DIFFERENT
batches run concurrently (own stream + execution context each) next to the grid evaluation (matrix-pipe bound) of an
older batch on --decoder-blocks of the CUs.  Every one of the K batches runs start to finish inside the timed region
(pipeline fill and drain included).  --pipeline 0 runs loop then grids on one stream.

The JSON line also carries: `roofline` (dominant kernel = forward decoder, ALGORITHMIC flops vs the
fp16 matrix peak; the issued figure beside it), `roofline_loop` (denoiser weight stream vs HBM), `w_trace` (the decoder
kernels timed on the thin-shell query trace of a trained-model-like field), `e2` (through marching cubes, host side)
and `cpu_baseline` (the oracle on this host's cores, bounded sample).

Multi-GPU: shapes are independent -> each rank owns its own B shapes (weak scaling), no
data-path collective; ranks meet only at the timing barriers.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FWD_FLOP = 5_308_416          # per forward decoder query
UNET_WEIGHT_BYTES = 553_294_340   # 138 323 585 fp32 parameters streamed once per denoiser evaluation;
                                  # the f16x2 planes (two fp16 per weight) are the same number of bytes
UNET_FLOP_PER_SAMPLE = {32: 2.057e9, 64: 4.104e9}
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3     # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TF = 2500.0     # MI355X dense fp16/bf16 matrix peak (same guide; not the 2:1-sparsity figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="shapes per GPU per step")
    ap.add_argument("--resolution", type=int, default=512)
    ap.add_argument("--diffusion-steps", type=int, default=1000, help="1000 = full DDPM chain (the metric)")
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--decoder-precision", choices=["f16x2", "fp32"], default="f16x2",
                    help="forward decoder kernel arithmetic (include/surfd_hip.h: surfd_decoder_set_precision)")
    ap.add_argument("--decoder-blocks", type=int, default=192,
                    help="with --pipeline 1: persistent decoder workgroups per launch while the next batch's reverse loop "
                         "runs on the remaining CUs (the last batch's grids, with nothing left to overlap, use every CU)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1: overlap the reverse loops of the next batches with the grid evaluation of the current one")
    ap.add_argument("--loop-chains", type=int, default=3,
                    help="with --pipeline 1: reverse loops (of different batches) in flight at once, each on its own stream "
                         "and execution context (MDM.replica)")
    ap.add_argument("--loop-cus", type=int, default=64,
                    help="pipeline mode: the CU budget each reverse loop sizes its split-K for (MDM.set_cu_budget); 256 = as if alone")
    ap.add_argument("--unet-precision", choices=["f16x2", "fp32"], default="f16x2",
                    help="denoiser conv arithmetic (include/surfd_hip.h: surfd_unet_set_precision)")
    ap.add_argument("--workload", choices=["real", "trace"], default="real",
                    help="real: coarse-to-fine grids of the synthetic decoder (the headline); trace: the decoder kernels over the "
                         "query lists a trained-model-like thin-shell field produces (SURVEY.md §8d W-trace)")
    ap.add_argument("--batch-grids", type=int, default=1,
                    help="1: the grids of a batch are refined together, one decoder launch per level for all shapes "
                         "(meshudf.fill_grids); 0: shape after shape")
    ap.add_argument("--timeline", action="store_true", help="print per-batch loop / grid completion times of the timed region to stderr")
    ap.add_argument("--no-trace", action="store_true", help="skip the untimed W-trace measurement")
    ap.add_argument("--no-e2", action="store_true", help="skip the E2 (through marching cubes) estimate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def setup_dist(n_gpus):
    if n_gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(n_gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")      # RCCL over xGMI
    assert world == n_gpus, f"--gpus {n_gpus} but WORLD_SIZE={world}"
    return world, rank, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def build_models(latent, precision, unet_precision="f16x2"):
    from surfd_amd import synth
    from surfd_amd.cbndec import CbnDecoder
    from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
    from surfd_amd.spec import DecoderConfig
    args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="deepfashion3d",
                                 noise_schedule="cosine", sigma_small=True, clip_value=1.0)
    model, diffusion = create_model_and_diffusion(args)
    load_model_wo_clip(model, synth.synth_unet_state_dict())
    model.to("cuda")
    model.eval()
    model.set_precision(unet_precision)
    dec = CbnDecoder(63, latent, 512, 5)
    dec.load_state_dict(synth.synth_decoder_state_dict(DecoderConfig(latent_dim=latent)), strict=True)
    dec = dec.cuda().eval()
    dec.set_precision(precision)
    return model, diffusion, dec


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(T, B, n_fwd, n_grad):
    """The oracle (CPU restatement of the reference's op graph) timed on this host on a bounded
    sample of the same workload, scaled to shapes/s: a few denoiser steps at batch B, forward decoder
    queries and forward+backward queries at the two chunk sizes the reference scripts use (the faster is quoted)."""
    from oracle import decoder as odec
    from oracle import unet as ounet
    from surfd_amd import synth
    # torch's CPU kernels stop scaling (and oversubscribe) far below the core count of a GPU host:
    # 32 threads was the fastest setting for this op mix; "cores" reports what was actually used
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sd = synth.synth_unet_state_dict()
    x = torch.randn(B, 1, 32)
    t = torch.full((B,), 500)
    with torch.no_grad():
        ounet.unet_forward(sd, x, t)
        t0 = time.time()
        n_it = 3
        for _ in range(n_it):
            ounet.unet_forward(sd, x, t)
        step_s = (time.time() - t0) / n_it
    dsd = synth.synth_decoder_state_dict()
    f = odec.make_udf_func(dsd, torch.randn(1, 32) * 0.8)
    pts = torch.rand(16384, 3) * 2 - 1
    odec.sample_udf(f, pts[:4096], 4096)
    rates = {}
    for chunk in (4096, 16384):
        t0 = time.time()
        odec.sample_udf(f, pts, chunk)
        rates[chunk] = 16384 / (time.time() - t0)
    fwd_chunk = max(rates, key=rates.get)
    fwd_rate = rates[fwd_chunk]
    odec.sample_grads(f, pts[:1024], 1024)
    t0 = time.time()
    odec.sample_grads(f, pts[:4096], 4096)
    grad_rate = 4096 / (time.time() - t0)
    per_shape = T * step_s / B + n_fwd / fwd_rate + n_grad / grad_rate
    return {"value": 1.0 / per_shape, "unit": "shapes/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(),
            "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{n_it} denoiser steps at batch {B} ({step_s * 1e3:.0f} ms/step), 16384 decoder forward queries "
                      f"(chunk {fwd_chunk}: {fwd_rate:.0f} pts/s), 4096 forward+backward queries (chunk 4096: {grad_rate:.0f} pts/s); "
                      f"extrapolated to {T} steps + {n_fwd:.0f} fwd + {n_grad:.0f} grad queries per shape (grid bookkeeping excluded)"}


def analytic_field_gpu(c):
    """SURVEY.md §8c G10's thin-shell field u(p) = min(0.1, d(p)) on the device (sphere shell above z=0, torus-like
    rim below): the occupancy pattern of a trained model, used to make the W-trace query lists."""
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    up = (torch.sqrt(x * x + y * y + z * z) - 0.6).abs()
    rho = torch.sqrt(x * x + y * y) - 0.6
    down = torch.sqrt(rho * rho + z * z)
    return torch.clamp(torch.where(z >= 0, up, down), max=0.1)


def make_trace(N):
    """Query lists of GridFiller(N) on the thin-shell field: forward points per level + gradient points."""
    from surfd_amd.meshudf import GridFiller
    lists = []

    class Field:
        def __call__(self, c):
            lists.append(("fwd", c.clone()))
            return analytic_field_gpu(c)

        def grads(self, c, max_batch):
            lists.append(("grad", c.clone()))
            return torch.zeros(c.shape[0], 3, device=c.device)
    u, g = GridFiller(N).fill_grid(Field(), 2 ** 30)
    del u, g
    torch.cuda.empty_cache()
    return lists


def time_trace(dec, lat, lists, reps=2):
    """Decoder kernels on the trace (values discarded): ms per shape and algorithmic TFLOP/s, forward and fwd+bwd."""
    n_f = sum(c.shape[0] for k, c in lists if k == "fwd")
    n_g = sum(c.shape[0] for k, c in lists if k == "grad")
    dec.bind_latents(lat.reshape(lat.shape[0], -1))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for kind, c in lists:                     # warm-up
        (dec.udf if kind == "fwd" else dec.udf_and_ngrad)(c, 0)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        for kind, c in lists:
            if kind == "fwd":
                dec.udf(c, 0)
    ev[1].record()
    for _ in range(reps):
        for kind, c in lists:
            if kind == "grad":
                dec.udf_and_ngrad(c, 0)
    ev[2].record()
    torch.cuda.synchronize()
    f_ms, g_ms = ev[0].elapsed_time(ev[1]) / reps, ev[1].elapsed_time(ev[2]) / reps
    return {"fwd_queries_per_shape": n_f, "grad_queries_per_shape": n_g, "decoder_fwd_ms_per_shape": f_ms,
            "decoder_fwd_bwd_ms_per_shape": g_ms,
            "fwd_algorithmic_tflops": n_f * FWD_FLOP / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0,
            "fwd_bwd_algorithmic_tflops": n_g * 2 * FWD_FLOP / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0}


def committed_traffic():
    """HBM traffic of the dominant kernel from the committed PMC pass of this command (profiles/, collected in its
    own rocprofv3 --pmc run as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled per the gfx950 note)."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except (OSError, ValueError):
            return None
    return None


def main():
    a = parse()
    world, rank, local = setup_dist(a.gpus)
    from surfd_amd import _native as Nn
    from surfd_amd import synth
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
    from surfd_amd.meshudf import fill_grids as fill_grids_batch
    L = Nn.lib()
    model, diffusion, dec = build_models(a.latent, a.decoder_precision, a.unet_precision)
    if a.diffusion_steps != 1000:
        from surfd_amd.diffusion import create_gaussian_diffusion
        diffusion = create_gaussian_diffusion(types.SimpleNamespace(noise_schedule="cosine", sigma_small=True),
                                              f"ddim{a.diffusion_steps}")
    T, B, N = diffusion.num_timesteps, a.batch, a.resolution
    first = rank * B                                     # global index of this rank's first shape
    noise = synth.synth_noise_batch(T, first, B, a.latent).cuda()
    filler = GridFiller(N)
    fillers = [filler] + [GridFiller(N) for _ in range(B - 1)] if a.batch_grids and B <= 8 else None
    udf = [torch.empty(N, N, N, device="cuda") for _ in range(B)]
    grads = [torch.empty(N, N, N, 3, device="cuda") for _ in range(B)]
    stats = []
    trace = make_trace(N) if a.workload == "trace" else None

    chains = [model] + [model.replica() for _ in range(max(1, a.loop_chains) - 1)] if a.pipeline else [model]
    for m in chains[1:]:
        m.set_precision(a.unet_precision)
    if a.pipeline:
        for m in chains:
            m.set_cu_budget(a.loop_cus)

    def sample_latents(chain=0):
        return diffusion.p_sample_loop(chains[chain], (B, 1, a.latent), clip_denoised=False, model_kwargs={"y": {}},
                                       noise_stream=noise, fused=True)

    def fill_grids(lat, collect=False):
        dec.bind_latents(lat.reshape(B, a.latent))
        if fillers is not None and trace is None:
            # all shapes of the batch level by level together: one persistent decoder launch per level (meshudf.fill_grids)
            fill_grids_batch(fillers, dec, list(range(B)), [(udf[k], grads[k]) for k in range(B)])
            if collect:
                stats.extend(f._stats() for f in fillers)
            return
        for k in range(B):
            if trace is not None:                        # W-trace: the decoder kernels over the trained-model-like query lists
                for kind, c in trace:
                    (dec.udf if kind == "fwd" else dec.udf_and_ngrad)(c, k)
                continue
            f = make_udf_func(dec, lat[k], sample=k)
            filler.fill_grid(f, 2 ** 16, out=(udf[k], grads[k]), stats=collect)
            if collect:
                stats.append(filler.last_stats)

    def one_step(collect=False):
        lat = sample_latents()
        fill_grids(lat, collect)
        return lat

    from surfd_amd.parallel import BatchPipeline
    pipe = BatchPipeline(dec, lambda s, q: sample_latents(q), lambda s, lat: fill_grids(lat), a.decoder_blocks,
                         loop_chains=len(chains))

    def run_steps(k_steps):
        """k_steps full passes (every batch: reverse loop + B grids), start to finish."""
        if not a.pipeline:
            for _ in range(k_steps):
                one_step()
        elif k_steps:
            pipe.run(k_steps)

    run_steps(a.warmup)
    pipe.record_timeline = a.timeline
    barrier(world)
    L.surfd_profile_enable(1)
    t0 = time.perf_counter()
    run_steps(a.steps)
    barrier(world)
    elapsed = time.perf_counter() - t0
    if a.timeline and a.pipeline:
        for m in pipe.timeline:
            print("[timeline] batch %2d: loop done %8.1f ms, grids %8.1f -> %8.1f ms" % (m["batch"], m["loop_done_ms"], m["grids_start_ms"], m["grids_done_ms"]), file=sys.stderr)
        pipe.record_timeline = False
    L.surfd_profile_enable(0)
    prof = {}
    for kind, name in [(0, "dec_fwd"), (1, "dec_grad"), (2, "loop")]:
        n, ms = C.c_int64(), C.c_double()
        Nn.check(L.surfd_profile_read(kind, C.byref(n), C.byref(ms)))
        prof[name] = (n.value, ms.value)
    # ---- untimed extras: workload counters (deterministic), one loop alone, the trace workload -------------
    dec.set_grid_blocks(0)
    if a.pipeline and a.loop_cus != 256:
        model.set_cu_budget(256)                         # one loop alone owns the chip
        sample_latents()                                 # re-capture outside the timed call
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    lat = sample_latents()
    torch.cuda.synchronize()
    loop_alone_ms = (time.perf_counter() - t1) * 1e3
    if trace is None:
        fill_grids(lat, collect=True)
        torch.cuda.synchronize()
        n_fwd = sum(sum(s["fwd_per_level"]) for s in stats) / B
        n_grad = sum(s["grad"] for s in stats) / B
    else:
        n_fwd = sum(c.shape[0] for k, c in trace if k == "fwd")
        n_grad = sum(c.shape[0] for k, c in trace if k == "grad")
    sat = sum(m.saturation_count() for m in chains) + dec.saturation_count()
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        from surfd_amd.parallel import gather_latents
        alllat = gather_latents(lat.contiguous(), [B] * world)            # ncclAllGather over xGMI: every rank's latents
        rccl_ranks = int(alllat.shape[0] // B)
    if rank != 0:
        return
    w_trace = None
    if not a.no_trace and rank == 0:
        del udf, grads
        torch.cuda.empty_cache()
        w_trace = time_trace(dec, lat, trace if trace is not None else make_trace(N))
    shapes = world * B * a.steps
    fwd_launches, fwd_ms = prof["dec_fwd"]
    grad_launches, grad_ms = prof["dec_grad"]
    loops, loop_ms = prof["loop"]
    fwd_flop_total = n_fwd * B * a.steps * FWD_FLOP            # this rank, timed region
    algorithmic = fwd_flop_total / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
    f16 = a.decoder_precision == "f16x2"
    peak = F16_MFMA_PEAK_TF if f16 else FP32_MFMA_PEAK_TF
    kname = ("decoder_kernel<false, f16x2> (fused encode + 11-layer CBN MLP + sigmoid; split-fp16 operands, fp32 accumulate)"
             if f16 else "decoder_kernel<false> (fused encode + 11-layer CBN MLP + sigmoid)")
    dtype = ("f32 (every matrix product of decoder and denoiser as three split-fp16 products on the fp16 MFMA pipe with fp32 "
             "accumulation — fp32-class error, same golden tolerances as the exact-fp32 kernels; samplers, attention, "
             "embedding MLP and grid bookkeeping in exact fp32)") if f16 and a.unet_precision == "f16x2" else "f32"
    pmc = committed_traffic()
    # reverse loop against its roofline: per evaluation max(weight bytes / HBM, B * flops / matrix peak)
    flops_eval = B * UNET_FLOP_PER_SAMPLE.get(a.latent, 2.057e9 * a.latent / 32)
    unet_peak = F16_MFMA_PEAK_TF / 3.0 if a.unet_precision == "f16x2" else FP32_MFMA_PEAK_TF      # algorithmic flops / s
    roof_eval_us = max(UNET_WEIGHT_BYTES / (HBM_PEAK_GBS * 1e9), flops_eval / (unet_peak * 1e12)) * 1e6
    streamed_gbs = loops * T * UNET_WEIGHT_BYTES / elapsed / 1e9                  # all chains, whole timed region
    out = {
        "metric": "shapes/sec end-to-end (1000-step uncond, 512^3 UDF) at 1/2/4/8 GPU",
        "value": shapes / elapsed, "unit": "shapes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic (seeded random-init weights of the reference architectures, seeded noise)",
        "rccl_ranks": rccl_ranks,
        "config": {"workload": f"unconditional, {T}-step {'DDPM' if a.diffusion_steps == 1000 else 'DDIM'}, L={a.latent}, "
                               + (f"{N}^3 coarse-to-fine UDF grid + gradients (end point E1: grids resident in HBM), W-real: the "
                                  "synthetic decoder's own occupancy" if trace is None else
                                  f"W-trace: decoder over the {N}^3 thin-shell query lists (2.05 M fwd + 0.75 M grad per shape)")
                               + f", {B} shapes/GPU (BASELINE configs[2] per-GPU shard)",
                   "shapes_per_gpu": B, "resolution": N, "diffusion_steps": T,
                   "decoder_fwd_queries_per_shape": n_fwd, "decoder_grad_queries_per_shape": n_grad,
                   "decoder_precision": a.decoder_precision, "unet_precision": a.unet_precision,
                   "fp16_range_saturations": sat,
                   "pipeline": (f"{len(chains)} reverse loops of different batches in flight (own stream + context each) next to the "
                                f"grids of an older batch (each loop sizes its split-K for {a.loop_cus} CUs); the decoder kernels run on {a.decoder_blocks} of the 256 CUs while loops "
                                "are in flight and on all of them for the last round of batches; every batch runs start to finish inside the "
                                "timed region") if a.pipeline else "none (loop then grids, one stream)",
                   "parallelism": f"shape-parallel x{world}, no data-path collective (latents all_gathered after the timed region)"},
        "roofline": {"kernel": kname, "bound": "mfma",
                     "achieved": algorithmic, "peak": peak, "unit": "TFLOP/s", "frac": algorithmic / peak,
                     "traffic": (pmc or {}).get("decoder_fwd_hbm_bytes_per_launch"),
                     "traffic_source": (pmc or {}).get("source"),
                     "algorithmic_bytes_per_launch": 16.0 * n_fwd * B * a.steps / max(fwd_launches, 1),
                     "launches": fwd_launches, "avg_launch_ms": fwd_ms / max(fwd_launches, 1),
                     "flop_per_point": FWD_FLOP,
                     "issued_tflops": (3.0 if f16 else 1.0) * algorithmic, "issued_frac": (3.0 if f16 else 1.0) * algorithmic / peak,
                     "cus": (f"{a.decoder_blocks} of 256 while loops are in flight, 256 for the last round of batches; peak is the whole chip's")
                            if a.pipeline else "256"},
        "roofline_loop": {"kernel": "conv2_kernel<8,false> x84 + attn_kernel x16 per denoiser evaluation (hipGraph replay)",
                          "bound": "hbm", "achieved": streamed_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": streamed_gbs / HBM_PEAK_GBS,
                          "traffic": (pmc or {}).get("unet_eval_hbm_bytes"),
                          "algorithmic_bytes_per_evaluation": UNET_WEIGHT_BYTES,
                          "roof_us_per_evaluation": roof_eval_us,
                          "one_loop_alone_ms_per_evaluation": loop_alone_ms / T,
                          "one_loop_alone_frac": roof_eval_us * 1e-3 / (loop_alone_ms / T),
                          "in_pipeline_ms_per_evaluation": loop_ms / max(loops, 1) / T,
                          "loops_in_flight": len(chains)},
        "breakdown_ms_per_step": {"reverse_loop_latency": loop_ms / max(loops, 1), "decoder_fwd": fwd_ms / a.steps,
                                  "decoder_fwd_bwd": grad_ms / a.steps,
                                  "decoder_fwd_bwd_tflops": (n_grad * B * a.steps * 2 * FWD_FLOP) / (grad_ms * 1e-3) / 1e12 if grad_ms > 0 else 0.0},
    }
    if w_trace is not None:
        # shapes/s on the trace workload if the grids were the only stage (the loop overlaps it in the pipeline)
        per_shape_ms = w_trace["decoder_fwd_ms_per_shape"] + w_trace["decoder_fwd_bwd_ms_per_shape"]
        w_trace["note"] = ("decoder kernels alone on all CUs, values discarded; with this occupancy a batch's grids take "
                           f"{per_shape_ms * B:.0f} ms, so the step is bound by the reverse loops ({loop_alone_ms:.0f} ms alone)")
        out["w_trace"] = w_trace
    if not a.no_e2:
        out["e2"] = e2_estimate(a, elapsed, shapes, world)
    if not a.no_cpu_baseline and world == 1:          # a stated baseline of the N=1 line only
        out["cpu_baseline"] = cpu_baseline(T, B, n_fwd, n_grad)
    print(json.dumps(out))


def e2_estimate(a, elapsed, shapes, world=1):
    """End point E2 = E1 + marching cubes on the host.  The native mesher is timed on one analytic
    thin-shell field of the benchmark's resolution and folded in as a host-side stage that runs on other cores
    while the GPU works on the next batches."""
    try:
        from surfd_amd import mcubes
    except Exception as e:       # the mesher is a later §8f row
        return {"status": f"native marching cubes unavailable ({type(e).__name__})"}
    # one meshing thread per shape of a batch and per GPU (the host of an 8-GPU node has the cores: 8 x 8 of 256)
    return mcubes.bench_e2(a.resolution, shapes / elapsed, threads=8 * world)


if __name__ == "__main__":
    main()
