#!/usr/bin/env python3
"""Headline benchmark: shapes/sec end-to-end for Surf-D sampling.

One "step" = one pass of the hot path over one batch of B (8) synthetic shapes on every rank:
    noise -> 1000-step DDPM reverse loop over the latent denoiser
          -> per shape: coarse-to-fine UDF grid (N^3) + spatial gradient, resident in HBM          (end point E1)
          -> [--endpoint e2] near-surface band compacted on the device, copied to pinned host memory on a copy stream
             and meshed by the native UDF marching cubes on host threads, every mesh finished inside the timed region
Weights are the deterministic synthetic tensors of surfd_amd.synth (no checkpoints exist offline).  Every step of every
rank works on DIFFERENT shapes: noise (and conditioning) is seeded per global shape index
((step * world + rank) * B + k), so a shape's result depends neither on the sharding nor on the schedule.

    python bench.py --gpus 1 --steps 6 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Schedules (--schedule):
  phased (default)  time-sliced: the reverse loops of a ROUND of steps first, as --loop-chains wide loops at once (each
                    over up to --loop-batches steps' latents: the denoiser's 553 MB of weights are streamed once per
                    evaluation for all of them; conv kernel in its wide form), the whole chip theirs; then the grids of
                    the round, the decoder's persistent workgroups on every CU (surfd_amd.parallel.PhasedPipeline)
  overlap           round 2's software pipeline: --loop-chains narrow loops of different steps on a quarter of the chip
                    next to the grids of an older step (surfd_amd.parallel.BatchPipeline)
  sequential        loop, then grids, step after step on one stream
Every step runs start to finish inside the timed region in every schedule.

Configurations (--config; BASELINE.json configs[1..4], one GPU's share): c3 (default, the metric's: unconditional, L=32,
512^3), c2 (same at 256^3), c4 (text-conditioned: per-shape context, classifier-free wrapper with scale 3.0, L=64, D=64,
512^3), c5 (image-conditioned: per-shape context, L=64, D=64, 512^3).  --workload trace replaces the synthetic decoder's
own coarse-to-fine queries (8.3 % of 512^3) by the query lists of a trained-model-like thin shell (1.5 %), loops included.

The JSON line carries: `roofline` (the kernel with the most GPU time = forward decoder: ALGORITHMIC flops
vs the fp16 matrix peak, the issued figure beside it), `roofline_loop` (denoiser weight stream vs HBM), `time_share`,
`w_trace`, `e2`, `cpu_baseline` (the oracle on this host's cores, bounded sample), and for N > 1 `per_rank`.
Multi-GPU: shapes are independent -> each rank owns its own B shapes per step (weak scaling), no data-path collective;
ranks meet only at the timing barriers.  --mode grid-shard instead splits every level of ONE shape's grid over the ranks
(device-side counts, ncclAllGather of the values over xGMI).  Prints ONE JSON line on rank 0.
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
MAX_CLOCK_GHZ = 2.4          # MI355X_MICROARCH.md: max shader clock; the matrix peaks are quoted at it
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3     # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TF = 2500.0     # MI355X dense fp16/bf16 matrix peak (same guide; not the 2:1-sparsity figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c3",
                    help="BASELINE.json configs[1..4], per-GPU share (c3 = the metric's configuration)")
    ap.add_argument("--batch", type=int, default=8, help="shapes per GPU per step")
    ap.add_argument("--resolution", type=int, default=None, help="grid resolution (default: the config's, 512 / 256 for c2)")
    ap.add_argument("--diffusion-steps", type=int, default=1000, help="1000 = full DDPM chain (the metric)")
    ap.add_argument("--latent", type=int, default=None, help="latent length (default: the config's, 32 / 64 for c4, c5)")
    ap.add_argument("--decoder-precision", choices=["f16x2", "fp32"], default="f16x2",
                    help="decoder kernel arithmetic (include/surfd_hip.h: surfd_decoder_set_precision)")
    ap.add_argument("--unet-precision", choices=["f16x2", "fp32"], default="f16x2",
                    help="denoiser conv arithmetic (include/surfd_hip.h: surfd_unet_set_precision)")
    ap.add_argument("--schedule", choices=["phased", "overlap", "sequential"], default="phased")
    ap.add_argument("--pipeline", type=int, default=None, help="(round-2 flag) 1 = --schedule overlap, 0 = --schedule sequential")
    ap.add_argument("--loop-chains", type=int, default=None,
                    help="reverse loops in flight at once, each on its own stream and execution context (MDM.replica); "
                         "default 2 (phased) / 3 (overlap)")
    ap.add_argument("--loop-driver", choices=["interleaved", "threads"], default="interleaved",
                    help="phased: the loops of a round driven by one host thread handing graph replays to them in turn (fixed "
                         "submission order; profiles/r05_loop_experiments.md §9) or by one host thread per loop (through round 4)")
    ap.add_argument("--loop-chunk", type=int, default=1, help="interleaved driver: graph replays handed to a loop per turn")
    ap.add_argument("--loop-batches", type=int, default=10,
                    help="phased: steps whose latents ride in ONE wide reverse loop (10 x 8 = 80 latents per loop)")
    ap.add_argument("--wide-design-batch", type=int, default=80,
                    help="phased: design batch of the conv kernel's wide form (MDM.set_wide); the K split, hence the bits, depend on it, not on the loop width")
    ap.add_argument("--overlap-blocks", type=int, default=0,
                    help="phased: 0 = time-sliced rounds; D > 0 = the loops of round r + 1 run next to the grids of round r, which then "
                         "use D persistent decoder workgroups (the chip is power-bound under the decoder: 192 CUs deliver 94 %% of 256)")
    ap.add_argument("--first-round", type=int, default=0, help="phased with --overlap-blocks: steps in the first round (its loops run alone)")
    ap.add_argument("--decoder-blocks", type=int, default=192,
                    help="overlap: persistent decoder workgroups per launch while loops run on the remaining CUs")
    ap.add_argument("--loop-cus", type=int, default=64,
                    help="overlap: the CU budget each reverse loop sizes its split-K for (MDM.set_cu_budget)")
    ap.add_argument("--workload", choices=["real", "trace"], default="real",
                    help="real: coarse-to-fine grids of the synthetic decoder (the headline); trace: reverse loops + the decoder "
                         "kernels over the query lists a trained-model-like thin-shell field produces (SURVEY.md §8d W-trace), end to end")
    ap.add_argument("--endpoint", choices=["e1", "e2"], default="e1",
                    help="e1: grids + gradients resident in HBM (the metric's end point); e2: through marching cubes, every mesh "
                         "finished inside the timed region (band compaction on the device, pinned D2H, host mesher threads)")
    ap.add_argument("--mesh-threads", type=int, default=0, help="e2: host meshing threads per rank (0 = min(8, cores / ranks - chains))")
    ap.add_argument("--mode", choices=["shape-parallel", "grid-shard"], default="shape-parallel",
                    help="grid-shard: every level of one shape's grid split over the ranks (needs --gpus >= 2 to mean anything)")
    ap.add_argument("--shard-path", choices=["native", "callback"], default="native",
                    help="grid-shard: native = surfd_grid_shard_* (voxel-ordered lists, tiles r, r + G, ... of every level, fixed-capacity buffers summed "
                         "over the ranks, no host read between levels); callback = round 3's ShardedField over the host-callback grid API")
    ap.add_argument("--shard-capacity", type=int, default=0,
                    help="grid-shard native: points EVERY level's exchange buffer holds (0 = per level, from the untimed shapes of the first step: 2 x the largest count seen at that level — 4 x for gradient points — in whole tiles of every rank, at least 2^16)")
    ap.add_argument("--shard-grad-capacity", type=int, default=0, help="grid-shard native: gradient points the exchange buffer holds (0 = sized the same way)")
    ap.add_argument("--batch-grids", type=int, default=1,
                    help="1: the grids of a step are refined together, one decoder launch per level for all shapes "
                         "(meshudf.fill_grids); 0: shape after shape")
    ap.add_argument("--timeline", action="store_true", help="print per-round loop / grid completion times of the timed region to stderr")
    ap.add_argument("--no-trace", action="store_true", help="skip the untimed W-trace measurement")
    ap.add_argument("--no-e2", action="store_true", help="skip the untimed E2 stage figures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict", action="store_true", help="skip the strict-C3 pass (one batch-8 request in flight, sequential)")
    ap.add_argument("--strict-steps", type=int, default=0, help="requests timed by the strict-C3 pass (0 = one per step of the headline: the same shapes)")
    ap.add_argument("--no-trace-e2", action="store_true", help="skip the timed W-trace / E2 pass (trained-model-like field through marching cubes)")
    ap.add_argument("--trace-steps", type=int, default=0, help="steps timed by the W-trace / E2 pass (0 = --steps)")
    a = ap.parse_args()
    if a.loop_chunk < 1:
        ap.error("--loop-chunk must be >= 1 graph replay per loop and turn")
    cfg = CONFIGS[a.config]
    if a.resolution is None:
        a.resolution = cfg["resolution"]
    if a.latent is None:
        a.latent = cfg["latent"]
    if a.pipeline is not None:
        a.schedule = "overlap" if a.pipeline else "sequential"
    if a.loop_chains is None:
        a.loop_chains = 3 if a.schedule == "overlap" else 2
    return a


# BASELINE.json configs[1..4] as one GPU's share (C2..C5)
CONFIGS = {
    "c2": {"cond_mode": "no_cond", "latent": 32, "resolution": 256, "cfg": False,
           "name": "C2 = BASELINE configs[1]: unconditional, 256^3, batch 8 on one GPU"},
    "c3": {"cond_mode": "no_cond", "latent": 32, "resolution": 512, "cfg": False,
           "name": "C3 = BASELINE configs[2]: unconditional, 512^3, 8 shapes per GPU (the per-GPU shard of batch 64 over 8 GPUs)"},
    "c4": {"cond_mode": "text", "latent": 64, "resolution": 512, "cfg": True,
           "name": "C4 = BASELINE configs[3]: text-conditioned (per-shape 512-d context in place of the CLIP text tower, "
                   "classifier-free wrapper, scale 3.0), L=64, 512^3, 8 shapes per GPU"},
    "c5": {"cond_mode": "img", "latent": 64, "resolution": 512, "cfg": False,
           "name": "C5 = BASELINE configs[4]: image-conditioned (per-shape 512-d context in place of the CLIP image tower), L=64, 512^3, "
                   "8 shapes per GPU"},
}


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


T_PROCESS_START = time.perf_counter()
PINNED_CPUS = None          # this rank's CPU set once pin_rank_cpus has run (None: not pinned)


def _cpu_ranges(cpus):
    """[0, 1, 2, 3, 8, 9] -> '0-3,8-9'"""
    cpus, out, i = sorted(cpus), [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def rank_cpu_slice(cpus, world, local):
    """The CPUs rank `local` of `world` keeps: equal contiguous slices of the CPUs this process may run on (host logic, unit
    tested).  Contiguous ids are neighbours in the topology on the hosts this runs on (cores of one socket / NUMA node are
    numbered together), so a rank's loop threads, meshing threads and its share of torch's pool stay near each other and the
    ranks never contend for a core.  Fewer than two CPUs per rank: no pinning."""
    cpus = sorted(cpus)
    per = len(cpus) // max(world, 1)
    if world <= 1 or per < 2:
        return None
    return cpus[local * per:(local + 1) * per]


def pin_rank_cpus(world, local):
    """8 ranks x (2 loop threads + meshing threads + torch's intra-op pool) on one host: without placement the scheduler moves
    them across sockets and the first thing that bends in a weak-scaling line is the host side.  SURFD_BENCH_NO_PIN=1
    leaves the placement to the launcher."""
    global PINNED_CPUS
    if os.environ.get("SURFD_BENCH_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    mine = rank_cpu_slice(os.sched_getaffinity(0), world, local)
    if mine:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 16)))
        PINNED_CPUS = mine
    return mine


def setup_dist(n_gpus):
    if n_gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(n_gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    pin_rank_cpus(world, local)
    # SURFD_BENCH_BACKEND=gloo: development aid — runs the N > 1 code path (launcher, barriers, per-rank gather) with
    # several ranks SHARING the GPUs that exist (RCCL refuses two ranks on one device); the number it prints is not a
    # scaling measurement and says so
    backend = os.environ.get("SURFD_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local % torch.cuda.device_count() if backend == "gloo" else local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)      # nccl = RCCL over xGMI
    assert world == n_gpus, f"--gpus {n_gpus} but WORLD_SIZE={world}"
    return world, rank, local


def _host_if_gloo(t):
    import torch.distributed as dist
    return t.cpu() if dist.get_backend() == "gloo" else t


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(T, B, n_fwd, n_grad, latent=32, N=512):
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
    x = torch.randn(B, 1, latent)
    t = torch.full((B,), 500)
    with torch.no_grad():
        ounet.unet_forward(sd, x, t)
        t0 = time.time()
        n_it = 3
        for _ in range(n_it):
            ounet.unet_forward(sd, x, t)
        step_s = (time.time() - t0) / n_it
    from surfd_amd.spec import DecoderConfig
    dsd = synth.synth_decoder_state_dict(DecoderConfig(latent_dim=latent))
    f = odec.make_udf_func(dsd, torch.randn(1, latent) * 0.8)
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
    # grid bookkeeping (coordinates, masks, scatter of five refinement levels — everything of GridFiller except the field itself):
    # the oracle's fill of the analytic thin-shell field at 256^3, scaled by 8 to the N^3 arrays of 512^3 (the field is a few
    # elementwise ops; < 1 % of a shape either way)
    from oracle import gridfiller as ogrid
    ogrid.fill_grid(ogrid.analytic_field, 64, 2 ** 30)
    t0 = time.time()
    ogrid.fill_grid(ogrid.analytic_field, 256, 2 ** 30)
    book_s = (time.time() - t0) * (N / 256.0) ** 3
    per_shape = T * step_s / B + n_fwd / fwd_rate + n_grad / grad_rate + book_s
    out = {"value": 1.0 / per_shape, "unit": "shapes/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(),
           "host_cores": os.cpu_count(), "kind": "port",
           "sample": f"{n_it} denoiser steps at batch {B} ({step_s * 1e3:.0f} ms/step), 16384 decoder forward queries "
                     f"(chunk {fwd_chunk}: {fwd_rate:.0f} pts/s), 4096 forward+backward queries (chunk 4096: {grad_rate:.0f} pts/s), one 256^3 "
                     f"grid fill of an analytic field for the bookkeeping ({book_s:.2f} s per {N}^3 shape); "
                     f"extrapolated to {T} steps + {n_fwd:.0f} fwd + {n_grad:.0f} grad queries per shape"}
    # what ties this number to the reference itself: the oracle and the imported reference timed on the same cores of the build
    # container (tools/cpu_ratio.py; the reference does not exist on the GPU box)
    try:
        r = json.load(open(os.path.join(ROOT, "profiles", "r04_cpu_oracle_vs_reference.json")))
        out["oracle_over_reference_time"] = r["headline_mix"]["oracle_over_reference"]
        out["reference_equivalent_value"] = out["value"] * r["headline_mix"]["oracle_over_reference"]
        out["oracle_vs_reference"] = (f"profiles/r04_cpu_oracle_vs_reference.json: on {r['threads']} threads of the build container the oracle takes "
                                      f"{r['headline_mix']['oracle_over_reference']:.3f} x the reference's time on this op mix (per case: "
                                      + ", ".join(f"{x['case'].split(',')[0]} {x['oracle_over_reference']:.2f}" for x in r["rows"]) + ")")
    except (OSError, ValueError, KeyError):
        out["oracle_vs_reference"] = "profiles/r04_cpu_oracle_vs_reference.json missing: ratio not recorded"
    return out


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


def thin_shell_grid(N):
    """(udf [N,N,N], grads [N,N,N,3]) of the thin-shell field on the device, as GridFiller(N) fills them (analytic
    -normalize(grad) where the reference would differentiate): the mesher's input in the W-trace end-to-end mode."""
    from surfd_amd.meshudf import GridFiller

    class Field:
        def __call__(self, c):
            return analytic_field_gpu(c)

        def grads(self, c, max_batch):
            out = torch.empty_like(c)
            for lo in range(0, c.shape[0], 1 << 20):
                p = c[lo:lo + (1 << 20)].detach().clone().requires_grad_(True)
                with torch.enable_grad():
                    (g,) = torch.autograd.grad(analytic_field_gpu(p).sum(), p)
                out[lo:lo + (1 << 20)] = -torch.nn.functional.normalize(g, dim=1)
            return out
    u, g = GridFiller(N).fill_grid(Field(), 2 ** 30)
    u = u.clamp_min(0).contiguous()
    torch.cuda.synchronize()
    return u, g.contiguous()


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


def latency_form_roofline(loop_s, T, B, latent, unet_precision):
    """The reverse loop of ONE narrow request (the conv kernel's latency form, what sample/generate_* runs) against ITS roof
    (the weakest number of the repository belongs in the driver line): B latents per evaluation are far below the
    ridge, so the weight stream — 553 MB per evaluation at the HBM peak — bounds an evaluation; `frac` = roof time / measured time."""
    flops = B * UNET_FLOP_PER_SAMPLE.get(latent, 2.057e9 * latent / 32)
    unet_peak = F16_MFMA_PEAK_TF / 3.0 if unet_precision == "f16x2" else FP32_MFMA_PEAK_TF
    hbm_us, mfma_us = UNET_WEIGHT_BYTES / (HBM_PEAK_GBS * 1e9) * 1e6, flops / (unet_peak * 1e12) * 1e6
    eval_us = loop_s / T * 1e6
    return {"kernel": "conv2_kernel (latency form: one 32-row tile per workgroup, K split follows the batch) x84 + attn_kernel x16 per evaluation",
            "bound": "hbm" if hbm_us >= mfma_us else "mfma", "us_per_evaluation": eval_us, "roof_us_hbm": hbm_us, "roof_us_mfma": mfma_us,
            "achieved": UNET_WEIGHT_BYTES / (eval_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s (weight stream: 553 MB per evaluation)",
            "frac": max(hbm_us, mfma_us) / eval_us, "latents_per_evaluation": B}


def build_config():
    """The compile-time configuration of the loaded libsurfd_hip.so (surfd_build_config: every experiment macro of the kernel
    sources with the value it was built with, and how many of them select a variant recorded as unsafe)."""
    from surfd_amd import _native as Nn
    try:
        return Nn.lib().surfd_build_config().decode()
    except Exception as e:      # an older library without the export
        return f"unknown ({type(e).__name__})"


def committed_traffic():
    """(profile, note): HBM traffic of the dominant kernels from the committed PMC passes of this command (profiles/,
    collected in their own rocprofv3 --pmc runs as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled per the gfx950 note).
    A profile names the kernel sources it was collected with (sha256); if they have changed since, nothing is quoted."""
    import hashlib
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        want = prof.get("source_sha256")
        if not want:
            return None, f"profiles/{name} does not name the kernel sources it was collected with: not quoted"
        for rel, digest in want.items():
            try:
                have = hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()
            except OSError:
                have = None
            if have != digest:
                return None, f"profiles/{name} was collected with a different {rel}: stale, not quoted"
        return prof, f"committed profile profiles/{name} (not a counter of this run; kernel sources unchanged since): " + str(prof.get("source"))
    return None, "no committed PMC profile"


def build_models(cfg, latent, precision, unet_precision="f16x2"):
    """(denoiser, diffusion, decoder) for a configuration: synthetic weights in the reference's checkpoint layouts."""
    from surfd_amd import synth
    from surfd_amd.cbndec import CbnDecoder
    from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
    from surfd_amd.spec import DecoderConfig
    # the conditioned denoisers are built as the image model (sketch/img/text share the architecture: a 512-d context
    # enters through sketch_emb, models/mdm.py:91-110); text mode is then selected on the module, as
    # sample/generate_text.py gets it from its checkpoint's args
    build_mode = "no_cond" if cfg["cond_mode"] == "no_cond" else "img"
    args = types.SimpleNamespace(cond_mode=build_mode, arch="OpenUNet", num_actions=9, dataset="deepfashion3d",
                                 noise_schedule="cosine", sigma_small=True, clip_value=1.0)
    model, diffusion = create_model_and_diffusion(args)
    load_model_wo_clip(model, synth.synth_unet_state_dict())
    model.to("cuda")
    model.eval()
    model.cond_mode = cfg["cond_mode"]
    model.set_precision(unet_precision)
    dec = CbnDecoder(63, latent, 512, 5)
    dec.load_state_dict(synth.synth_decoder_state_dict(DecoderConfig(latent_dim=latent)), strict=True)
    dec = dec.cuda().eval()
    dec.set_precision(precision)
    return model, diffusion, dec


class Job:
    """One schedule over one workload: everything `run_steps` needs, built once.  The headline job, the strict-C3 job (one
    batch-8 request in flight, sequential) and the timed W-trace / E2 job share the models, the noise and the decoder."""

    def __init__(self, a, world, rank, cfg, model, chains_pool, diffusion, dec, noise_bank, ctx_bank, *, workload, endpoint,
                 schedule, loop_batches, n_chains, wide, trace=None, trace_grid=None):
        from surfd_amd.cbndec import make_udf_func
        from surfd_amd.meshudf import GridFiller
        from surfd_amd.meshudf import fill_grids as fill_grids_batch
        from surfd_amd.parallel import BatchPipeline, PhasedPipeline
        self.a, self.world, self.rank, self.cfg = a, world, rank, cfg
        self.step_ids = None          # sequential schedule: the steps (= shape sets) the requests sample, in order; None = 0, 1, 2, ...
        self.diffusion, self.dec = diffusion, dec
        self.workload, self.endpoint, self.schedule = workload, endpoint, schedule
        T, B, N = diffusion.num_timesteps, a.batch, a.resolution
        self.T, self.B, self.N = T, B, N
        conditioned = cfg["cond_mode"] != "no_cond"
        self.n_chains = n_chains = max(1, n_chains) if schedule != "sequential" else 1
        self.loop_batches = loop_batches
        self.chains = chains = chains_pool[:n_chains]
        for m in chains:
            m.set_wide(wide if schedule == "phased" else 0)
            m.set_cu_budget(a.loop_cus if schedule == "overlap" else 256)
        if cfg["cfg"]:
            from surfd_amd.mdm import ClassifierFreeSampleModel
            wrapped = [ClassifierFreeSampleModel(m) for m in chains]
        else:
            wrapped = chains

        def loop_kwargs(first, n):
            if not conditioned:
                return {"y": {}}
            y = {"context": ctx_bank[first:first + n].reshape(n * B, -1).contiguous()}
            if cfg["cfg"]:
                y["scale"] = torch.full((n * B,), 3.0, device="cuda")
            return {"y": y}

        def sample_latents(first, n, chain=0):
            """ONE reverse loop over the latents of steps [first, first + n) -> [n * B, 1, L]"""
            noise = noise_bank[first:first + n].permute(1, 0, 2, 3, 4).reshape(T + 1, n * B, 1, a.latent).contiguous()
            return diffusion.p_sample_loop(wrapped[chain], (n * B, 1, a.latent), clip_denoised=False, model_kwargs=loop_kwargs(first, n),
                                           noise_stream=noise, fused=True)
        self.sample_latents = sample_latents

        def sample_round(parts, streams):
            """all loops of a round from ONE host thread, graph replays handed to the loops in turn (fixed submission order)"""
            jobs = []
            for (c, first, n), st in zip(parts, streams):
                with torch.cuda.stream(st):                   # the loop's inputs are gathered on the loop's own stream
                    jobs.append({"model": wrapped[c], "shape": (n * B, 1, a.latent), "model_kwargs": loop_kwargs(first, n), "stream": st,
                                 "noise_stream": noise_bank[first:first + n].permute(1, 0, 2, 3, 4).reshape(T + 1, n * B, 1, a.latent).contiguous()})
            return diffusion.fused_loops_interleaved(jobs, sampler="ddpm", clip_denoised=False, chunk=a.loop_chunk, wait_current=False)

        self.trace = trace if workload == "trace" else None
        trace = self.trace
        self.filler = filler = GridFiller(N)
        self.fillers = fillers = ([filler] + [GridFiller(N) for _ in range(B - 1)] if a.batch_grids and B <= 8 else [filler]) if trace is None else [filler]
        self.udf = udf = [torch.empty(N, N, N, device="cuda") for _ in range(B)] if trace is None else None
        self.grads = grads = [torch.empty(N, N, N, 3, device="cuda") for _ in range(B)] if trace is None else None
        self.mesher = None
        if endpoint == "e2":
            from surfd_amd.mcubes import BandMesher
            threads = a.mesh_threads or max(1, min(8, (len(PINNED_CPUS) if PINNED_CPUS else (os.cpu_count() or 8) // world) - n_chains))
            self.mesher = BandMesher(N, threads=threads, slots=2 * B)
        mesher = self.mesher

        def fill_grids(step, lat):
            dec.bind_latents(lat.reshape(B, a.latent))
            if trace is not None:                            # W-trace: the decoder kernels over the trained-model-like query lists
                for k in range(B):
                    for kind, c in trace:
                        (dec.udf if kind == "fwd" else dec.udf_and_ngrad)(c, k)
                    if mesher is not None:                   # ... and the mesh of the same field (222 793 vertices): band compaction, D2H, host mesher
                        mesher.submit(trace_grid[0], trace_grid[1], tag=(step, k))
                return
            if len(fillers) == B:
                # all shapes of the step level by level together: one persistent decoder launch per level (meshudf.fill_grids)
                fill_grids_batch(fillers, dec, list(range(B)), [(udf[k], grads[k]) for k in range(B)])
            else:
                for k in range(B):
                    filler.fill_grid(make_udf_func(dec, lat[k], sample=k), 2 ** 16, out=(udf[k], grads[k]), stats=False)
            if mesher is not None:
                for k in range(B):
                    mesher.submit(udf[k], grads[k], tag=(step, k))     # device-side band compaction + async D2H; meshing on host threads
        self.fill_grids = fill_grids

        if schedule == "phased":
            self.pipe = PhasedPipeline(sample_latents, fill_grids, chains=n_chains, max_loop_batches=loop_batches, overlap_blocks=a.overlap_blocks,
                                       decoder=dec, first_round_batches=a.first_round if a.overlap_blocks else 0,
                                       loops_fn=sample_round if a.loop_driver == "interleaved" else None)
        elif schedule == "overlap":
            self.pipe = BatchPipeline(dec, lambda s, q: sample_latents(s, 1, q), fill_grids, a.decoder_blocks, loop_chains=n_chains)
        else:
            self.pipe = None

    def run_steps(self, k_steps):
        """k_steps full passes (every step: reverse loop + B grids [+ B meshes]), start to finish."""
        if k_steps <= 0:
            return
        if self.pipe is None:
            for s in range(k_steps):
                sid = self.step_ids[s % len(self.step_ids)] if self.step_ids else s
                self.fill_grids(sid, self.sample_latents(sid, 1))
        else:
            self.pipe.run(k_steps)
        if self.mesher is not None:
            self.mesher.drain()                               # every mesh of the job is finished

    def reset_totals(self):
        if self.trace is None:
            for f in self.fillers:
                if f._handle is not None:
                    f.totals(reset=True)
        if self.mesher is not None:
            self.mesher.reset_stats()

    def measure(self, steps, warmup):
        """warmup untimed steps, then exactly `steps` timed ones between barrier + synchronize on both sides."""
        from surfd_amd import _native as Nn
        L = Nn.lib()
        B = self.B
        self.run_steps(warmup)
        if self.pipe is not None:
            self.pipe.record_timeline = True                  # a few HIP events per round: time_share / per_rank come from them
        torch.cuda.synchronize()
        self.reset_totals()
        self.dec.sustained_clock_ghz(reset=True)             # warm-up launches are not part of the record
        startup_s = time.perf_counter() - T_PROCESS_START     # imports, weight synthesis, packing, warm-up: everything before the clock
        barrier(self.world)
        L.surfd_profile_enable(1)
        t0 = time.perf_counter()
        self.run_steps(steps)
        torch.cuda.synchronize()
        local_elapsed = time.perf_counter() - t0
        barrier(self.world)
        elapsed = time.perf_counter() - t0
        timeline = list(self.pipe.timeline) if self.pipe is not None and self.pipe.record_timeline else []
        if self.pipe is not None:
            self.pipe.record_timeline = False
        L.surfd_profile_enable(0)
        prof = {}
        for kind, name in [(0, "dec_fwd"), (1, "dec_grad"), (2, "loop")]:
            n, ms = C.c_int64(), C.c_double()
            Nn.check(L.surfd_profile_read(kind, C.byref(n), C.byref(ms)))
            prof[name] = (n.value, ms.value)
        # ---- exact query counts of the timed region (kept on the device by the fills themselves) ------------------
        if self.trace is None:
            tot = [f.totals(reset=True) for f in self.fillers]
            fwd_total = float(sum(sum(t["fwd_per_level"]) for t in tot))
            grad_total = float(sum(t["grad"] for t in tot))
            assert sum(t["fills"] for t in tot) == B * steps, (sum(t["fills"] for t in tot), B * steps)
        else:
            fwd_total = float(sum(c.shape[0] for k, c in self.trace if k == "fwd")) * B * steps
            grad_total = float(sum(c.shape[0] for k, c in self.trace if k == "grad")) * B * steps
        mesh_stats = self.mesher.stats() if self.mesher is not None else None
        # time the loops had the chip: phased = up to the last loop of each round; otherwise they overlap the grids
        a = self.a
        if self.schedule == "phased" and timeline and a.overlap_blocks:
            loop_phase_ms = timeline[0]["loops_done_ms"]         # only the first round's loops have the chip to themselves
        elif self.schedule == "phased" and timeline:
            prev, loop_phase_ms = 0.0, 0.0
            for m in timeline:
                loop_phase_ms += m["loops_done_ms"] - prev
                prev = m["grids_done_ms"]
        else:
            loop_phase_ms = None
        return {"elapsed": elapsed, "local_elapsed": local_elapsed, "timeline": timeline, "prof": prof, "fwd_total": fwd_total,
                "grad_total": grad_total, "mesh_stats": mesh_stats, "loop_phase_ms": loop_phase_ms, "steps": steps,
                "clock_ghz": self.dec.sustained_clock_ghz(reset=True), "startup_s": startup_s, "rounds": len(timeline) if timeline else None}

    def close(self):
        if self.mesher is not None:
            self.mesher.close()
            self.mesher = None
        self.udf = self.grads = None


def decoder_rooflines(m, f16):
    """forward and forward+reverse decoder kernels of a measured pass against the matrix peak (algorithmic flops)."""
    peak = F16_MFMA_PEAK_TF if f16 else FP32_MFMA_PEAK_TF
    fl, fms = m["prof"]["dec_fwd"]
    gl, gms = m["prof"]["dec_grad"]
    fwd = m["fwd_total"] * FWD_FLOP / (fms * 1e-3) / 1e12 if fms > 0 else 0.0
    grd = m["grad_total"] * 2 * FWD_FLOP / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
    return {"bound": "mfma", "peak": peak, "unit": "TFLOP/s",
            "decoder_fwd": {"achieved": fwd, "frac": fwd / peak, "launches": fl, "avg_launch_ms": fms / max(fl, 1), "ms": fms},
            "decoder_fwd_bwd": {"achieved": grd, "frac": grd / peak, "launches": gl, "avg_launch_ms": gms / max(gl, 1), "ms": gms,
                                "accounting": "2 x forward flops per gradient query"}}


def main():
    a = parse()
    world, rank, local = setup_dist(a.gpus)
    if a.mode == "grid-shard":
        return grid_shard_main(a, world, rank)
    from surfd_amd import synth
    cfg = CONFIGS[a.config]
    model, diffusion, dec = build_models(cfg, a.latent, a.decoder_precision, a.unet_precision)
    if a.diffusion_steps != 1000:
        from surfd_amd.diffusion import create_gaussian_diffusion
        diffusion = create_gaussian_diffusion(types.SimpleNamespace(noise_schedule="cosine", sigma_small=True),
                                              f"ddim{a.diffusion_steps}")
    T, B, N = diffusion.num_timesteps, a.batch, a.resolution
    conditioned = cfg["cond_mode"] != "no_cond"
    n_steps_max = max(a.steps, a.warmup, 1)
    # global shape index of (step s, rank, k): every step of every rank samples different shapes
    gidx = lambda s: (s * world + rank) * B
    noise_bank = torch.stack([synth.synth_noise_batch(T, gidx(s), B, a.latent) for s in range(n_steps_max)], 0).cuda()   # [S, T+1, B, 1, L]
    ctx_bank = torch.stack([synth.synth_context(gidx(s), B) for s in range(n_steps_max)], 0).cuda() if conditioned else None
    n_chains = max(1, a.loop_chains) if a.schedule != "sequential" else 1
    chains = [model] + [model.replica() for _ in range(n_chains - 1)]
    for m in chains[1:]:
        m.set_precision(a.unet_precision)
    # (a conditioned loop keeps one embedding row per (iteration, latent) in HBM: 56 KB x T x latents = 4.5 GB for an
    # 80-wide loop — sized for 288 GB, not for a 16 GB card)
    loop_batches = a.loop_batches
    trace = make_trace(N) if a.workload == "trace" else None
    trace_grid = thin_shell_grid(N) if (a.endpoint == "e2" and trace is not None) else None   # the field whose query lists the trace is
    job = Job(a, world, rank, cfg, model, chains, diffusion, dec, noise_bank, ctx_bank, workload=a.workload, endpoint=a.endpoint,
              schedule=a.schedule, loop_batches=loop_batches, n_chains=n_chains, wide=a.wide_design_batch, trace=trace, trace_grid=trace_grid)
    sample_latents = job.sample_latents
    meas = job.measure(a.steps, a.warmup)
    elapsed, local_elapsed, timeline, prof = meas["elapsed"], meas["local_elapsed"], meas["timeline"], meas["prof"]
    fwd_total, grad_total, mesh_stats, loop_phase_ms = meas["fwd_total"], meas["grad_total"], meas["mesh_stats"], meas["loop_phase_ms"]
    if a.timeline:
        for m in timeline:
            print("[timeline] " + json.dumps(m), file=sys.stderr)
    n_fwd, n_grad = fwd_total / (B * a.steps), grad_total / (B * a.steps)
    job.close()
    # ---- untimed extras: one loop alone at the widest batch the schedule used --------------------------------------
    dec.set_grid_blocks(0)
    if a.schedule == "overlap" and a.loop_cus != 256:
        model.set_cu_budget(256)                         # one loop alone owns the chip
    alone_n = min(loop_batches, max(1, -(-a.steps // n_chains))) if a.schedule == "phased" else 1
    alone_n = min(alone_n, n_steps_max)
    lat = sample_latents(0, alone_n)                     # (re-)capture outside the timed call
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    lat = sample_latents(0, alone_n)
    torch.cuda.synchronize()
    loop_alone_ms = (time.perf_counter() - t1) * 1e3
    lat = lat[:B]
    sat = sum(m.saturation_count() for m in chains) + dec.saturation_count()
    rccl_ranks = 1
    per_rank = None
    if world > 1:
        import torch.distributed as dist
        fwd_l, fwd_m = prof["dec_fwd"]; grd_l, grd_m = prof["dec_grad"]; lp_n, lp_m = prof["loop"]
        loops_done = max((m.get("loops_done_ms", 0.0) for m in timeline), default=0.0)
        mine = _host_if_gloo(torch.tensor([local_elapsed * 1e3, lp_m, fwd_m, grd_m, loops_done, fwd_total, float(os.cpu_count() or 0),
                                           float(gidx(0)), meas["startup_s"], float(len(PINNED_CPUS) if PINNED_CPUS else 0),
                                           float(PINNED_CPUS[0] if PINNED_CPUS else -1), meas["clock_ghz"]],
                                          device="cuda", dtype=torch.float64))
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "ms_per_step": float(v[0]) / a.steps, "loop_latency_ms_sum": float(v[1]), "decoder_fwd_ms": float(v[2]),
                     "decoder_fwd_bwd_ms": float(v[3]), "last_round_loops_done_ms": float(v[4]), "decoder_fwd_queries": float(v[5]),
                     "first_shape_index": int(v[7]), "startup_s": float(v[8]),
                     "cpus": (f"{int(v[10])}-{int(v[10]) + int(v[9]) - 1}" if v[9] > 0 else None), "decoder_clock_ghz": float(v[11])}
                    for r, v in enumerate(allr)]
        for pr in per_rank:          # one line per rank on stderr: a driver timeout or a bent scaling line can be read from the log
            print(f"[bench] rank {pr['rank']}: ready after {pr['startup_s']:.1f} s, {pr['ms_per_step']:.1f} ms per step, shapes from {pr['first_shape_index']}, "
                  f"cpus {pr['cpus']}, decoder clock {pr['decoder_clock_ghz']:.2f} GHz", file=sys.stderr)
        tt = _host_if_gloo(torch.tensor([elapsed], device="cuda", dtype=torch.float64))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        from surfd_amd.parallel import gather_latents
        alllat = gather_latents(lat.contiguous(), [B] * world)            # ncclAllGather over xGMI: every rank's latents
        rccl_ranks = int(alllat.shape[0] // B)
    if rank != 0:
        return
    f16 = a.decoder_precision == "f16x2"
    # ---- the two other numbers of the line (N = 1 only, each its own timed region after the headline's) --------------
    strict = None
    if not a.no_strict and world == 1 and a.schedule == "phased" and a.workload == "real" and a.endpoint == "e1":
        # BASELINE configs[2] read strictly: ONE batch-8 request in flight per GPU — its reverse loop (8 latents, the conv
        # kernel's latency form), then its 8 grids, request after request on one stream
        sj = Job(a, world, rank, cfg, model, chains, diffusion, dec, noise_bank, ctx_bank, workload="real", endpoint="e1",
                 schedule="sequential", loop_batches=1, n_chains=1, wide=0)
        # the requests run the shapes of the headline's own steps — all of them by default (the synthetic shapes differ by +-25 % in
        # decoder queries: rounds 4-5 timed the first three, 13.1 M queries per shape against the 10.4 M the headline averages over);
        # with fewer requests the steps are spread evenly over the range; step ids and queries per shape are in the record
        k = max(1, min(a.strict_steps if a.strict_steps > 0 else a.steps, n_steps_max, a.steps))
        sj.step_ids = [min(a.steps - 1, (i * a.steps) // k) for i in range(k)]
        sm = sj.measure(k, 1)
        sj.close()
        loop_s = sm["prof"]["loop"][1] / max(sm["prof"]["loop"][0], 1) * 1e-3
        strict = {"value": B * k / sm["elapsed"], "unit": "shapes/s", "latency_s_per_request": sm["elapsed"] / k, "requests": k, "warmup": 1,
                  "shapes_per_request": B, "requests_in_flight": 1, "steps_sampled": sj.step_ids,
                  "decoder_fwd_queries_per_shape": sm["fwd_total"] / (B * k), "headline_decoder_fwd_queries_per_shape": n_fwd,
                  "reverse_loop_s_per_request": loop_s,
                  "decoder_fwd_s_per_request": sm["prof"]["dec_fwd"][1] / k * 1e-3, "decoder_fwd_bwd_s_per_request": sm["prof"]["dec_grad"][1] / k * 1e-3,
                  "other_s_per_request": sm["elapsed"] / k - loop_s - (sm["prof"]["dec_fwd"][1] + sm["prof"]["dec_grad"][1]) / k * 1e-3,
                  "other_means": "everything of a request that is neither the loop's graph replays nor a decoder kernel: embedding rows and noise of the loop, grid bookkeeping kernels, host submission",
                  "roofline": latency_form_roofline(loop_s, T, B, a.latent, a.unet_precision),
                  "what": "one batch-8 request in flight per GPU (BASELINE configs[2] = batch 64 over 8 GPUs, read strictly): 1000-step loop over 8 "
                          "latents (latency form of the conv kernel), then the request's 8 grids, sequentially on one stream"}
    trace_e2 = None
    if not a.no_trace_e2 and world == 1 and a.workload == "real" and a.endpoint == "e1" and a.schedule == "phased":
        # the trained-model-like workload END TO END, timed: wide reverse loops + decoder over the thin-shell query lists + band
        # compaction, pinned D2H and the host mesher, every mesh finished before the clock stops
        torch.cuda.empty_cache()
        tr = make_trace(N)
        tg = thin_shell_grid(N)
        tj = Job(a, world, rank, cfg, model, chains, diffusion, dec, noise_bank, ctx_bank, workload="trace", endpoint="e2",
                 schedule="phased", loop_batches=loop_batches, n_chains=n_chains, wide=a.wide_design_batch, trace=tr, trace_grid=tg)
        k = max(1, min(a.trace_steps or a.steps, n_steps_max))
        tm = tj.measure(k, 1)
        tstats = tm["mesh_stats"]
        tj.close()
        trace_e2 = {"status": "timed", "value": B * k / tm["elapsed"], "unit": "shapes/s", "steps": k, "warmup": 1, "ms_per_step": tm["elapsed"] / k * 1e3,
                    "workload": f"W-trace ({N}^3 thin shell: {tm['fwd_total'] / (B * k):.0f} forward + {tm['grad_total'] / (B * k):.0f} gradient queries per shape), "
                                "end point E2 (every mesh finished inside the timed region)",
                    "roofline": {**decoder_rooflines(tm, f16), "sustained_clock_ghz": tm["clock_ghz"] or None,
                                 "sustained_clock_means": "shader clock under the forward kernel's launches of this pass (the gradient kernel does not record one)"},
                    "time_share": ({"loops_alone_on_chip_ms": tm["loop_phase_ms"], "loops_frac": tm["loop_phase_ms"] / (tm["elapsed"] * 1e3)}
                                   if tm["loop_phase_ms"] is not None else None),
                    "mesher": tstats}
        del tr, tg
        torch.cuda.empty_cache()
    w_trace = None
    if not a.no_trace:
        torch.cuda.empty_cache()
        w_trace = time_trace(dec, lat, trace if trace is not None else make_trace(N))
    shapes = world * B * a.steps
    fwd_launches, fwd_ms = prof["dec_fwd"]
    grad_launches, grad_ms = prof["dec_grad"]
    loops, loop_ms = prof["loop"]
    algorithmic = fwd_total * FWD_FLOP / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
    peak = F16_MFMA_PEAK_TF if f16 else FP32_MFMA_PEAK_TF
    kname = ("decoder_fwd8_kernel (fused encode + 11-layer CBN MLP + sigmoid; 8 waves per 64-point tile, split-fp16 operands, fp32 accumulate)"
             if f16 else "decoder_kernel<false> (fused encode + 11-layer CBN MLP + sigmoid)")
    dtype = ("f32 (every matrix product of decoder and denoiser as three split-fp16 products on the fp16 MFMA pipe with fp32 "
             "accumulation — fp32-class error, same golden tolerances as the exact-fp32 kernels; samplers, attention, "
             "embedding MLP and grid bookkeeping in exact fp32)") if f16 and a.unet_precision == "f16x2" else "f32"
    pmc, pmc_note = committed_traffic()
    # reverse loop against its roofline: per evaluation max(weight bytes / HBM, latents * flops / matrix peak)
    lat_per_loop = alone_n * B
    flops_eval = lat_per_loop * UNET_FLOP_PER_SAMPLE.get(a.latent, 2.057e9 * a.latent / 32)
    unet_peak = F16_MFMA_PEAK_TF / 3.0 if a.unet_precision == "f16x2" else FP32_MFMA_PEAK_TF      # algorithmic flops / s
    roof_eval_us = max(UNET_WEIGHT_BYTES / (HBM_PEAK_GBS * 1e9), flops_eval / (unet_peak * 1e12)) * 1e6
    streamed_gbs = loops * T * UNET_WEIGHT_BYTES / ((loop_phase_ms * 1e-3) if loop_phase_ms else elapsed) / 1e9
    if a.schedule == "phased" and a.overlap_blocks:
        phased_txt = (f"overlapped rounds: the reverse loops of round r + 1 ({n_chains} wide loops at once, up to {loop_batches} steps each, conv kernel in its wide "
                      f"form) run next to the grids of round r on {a.overlap_blocks} persistent decoder workgroups (first round of {a.first_round or 'equal'} steps: its loops "
                      "run alone; last round's grids: all CUs); every step runs start to finish inside the timed region")
    else:
        phased_txt = None
    sched = {"phased": phased_txt or (f"time-sliced rounds: the reverse loops of up to {n_chains} x {loop_batches} steps as {n_chains} wide loops at once (conv kernel in "
                        f"its wide form, K split designed for {a.wide_design_batch} latents), the whole chip theirs, then the round's grids with the decoder's "
                        "persistent workgroups on every CU; every step runs start to finish inside the timed region"),
             "overlap": (f"{n_chains} reverse loops of different steps in flight (own stream + context each, split-K sized for {a.loop_cus} CUs) next to the grids "
                         f"of an older step on {a.decoder_blocks} of the 256 CUs"),
             "sequential": "none (loop then grids, one stream)"}[a.schedule]
    # what is in flight on one GPU in this schedule (say it next to shapes_per_gpu)
    if a.schedule == "phased":
        round_steps = min(a.steps, n_chains * loop_batches)
        latents_per_loop_cfg = B * min(loop_batches, max(1, -(-round_steps // n_chains)))
        shapes_in_flight = B * round_steps
    else:
        latents_per_loop_cfg, shapes_in_flight = B, B * (n_chains + 1 if a.schedule == "overlap" else 1)
    # the loop against the roof that BINDS at the width it ran at: per evaluation the weight stream needs
    # bytes / HBM peak, the matrix work latents x flops / matrix peak — at 80 latents per loop the matrix work is the larger one.
    # `frac` = roof time / measured time in the schedule (the loops of a round share the chip: roof time x loops in flight per
    # wall-clock evaluation); the HBM figure the earlier rounds quoted stays beside it.
    hbm_us = UNET_WEIGHT_BYTES / (HBM_PEAK_GBS * 1e9) * 1e6
    mfma_us = flops_eval / (unet_peak * 1e12) * 1e6
    sched_ms_eval = (loop_phase_ms / (T * max(1, -(-loops // n_chains)))) if (loop_phase_ms and loops) else None     # wall time of one evaluation of the loops running side by side
    binding = "mfma" if mfma_us >= hbm_us else "hbm"
    roofline_loop = {"kernel": "conv2_kernel x84 (the head's epilogue carries the posterior update and the loop counter) + attn_kernel x16 per denoiser evaluation (hipGraph replay)",
                     "bound": binding,
                     "roof_us_per_evaluation": roof_eval_us, "roof_us_hbm": hbm_us, "roof_us_mfma": mfma_us,
                     "frac": (n_chains * roof_eval_us * 1e-3 / sched_ms_eval) if sched_ms_eval else (roof_eval_us * 1e-3 / (loop_alone_ms / T)),
                     "frac_means": "roof time of the loops in flight / wall time per evaluation in the schedule, against the binding roof at this width",
                     "hbm": {"achieved": streamed_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": streamed_gbs / HBM_PEAK_GBS},
                     "achieved": (n_chains * flops_eval / (sched_ms_eval * 1e-3) / 1e12) if (sched_ms_eval and binding == "mfma") else streamed_gbs,
                     "peak": unet_peak if binding == "mfma" else HBM_PEAK_GBS, "unit": "TFLOP/s (algorithmic; x3 issued on the fp16 pipe)" if binding == "mfma" else "GB/s",
                     "traffic": (pmc or {}).get("unet_eval_hbm_bytes"),
                     "traffic_over_algorithmic": ((pmc or {}).get("unet_eval_fetch_bytes") / UNET_WEIGHT_BYTES) if (pmc or {}).get("unet_eval_fetch_bytes") else None,
                     "traffic_over_algorithmic_means": "fabric-side FETCH bytes per evaluation (committed PMC pass) / the 553 MB of weights an evaluation needs once",
                     "traffic_from": pmc_note,
                     "algorithmic_bytes_per_evaluation": UNET_WEIGHT_BYTES, "algorithmic_flop_per_evaluation": flops_eval,
                     "latents_per_loop": lat_per_loop, "loops_in_flight": n_chains, "loops": loops,
                     "one_loop_alone_ms_per_evaluation": loop_alone_ms / T,
                     "one_loop_alone_frac": roof_eval_us * 1e-3 / (loop_alone_ms / T),
                     "one_loop_alone_us_per_evaluation_and_latent": loop_alone_ms / T * 1e3 / lat_per_loop,
                     "in_schedule_ms_per_evaluation": loop_ms / max(loops, 1) / T,
                     "in_schedule_us_per_evaluation_and_latent": (loop_phase_ms * 1e3 / (T * B * a.steps)) if loop_phase_ms else None}
    out = {
        "metric": "shapes/sec end-to-end (1000-step uncond, 512^3 UDF) at 1/2/4/8 GPU",
        "value": shapes / elapsed, "unit": "shapes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic (seeded random-init weights of the reference architectures, seeded noise" + (", seeded context vectors" if conditioned else "") + "; different shapes in every step)",
        "rccl_ranks": rccl_ranks,
        **({"not_a_scaling_measurement": "SURFD_BENCH_BACKEND=gloo: the ranks share the GPUs of this box"} if os.environ.get("SURFD_BENCH_BACKEND") == "gloo" and world > 1 else {}),
        "config": {"workload": f"{cfg['name']}; {T}-step {'DDPM' if a.diffusion_steps == 1000 else 'DDIM'}, L={a.latent}, "
                               + (f"{N}^3 coarse-to-fine UDF grid + gradients, W-real: the synthetic decoder's own occupancy" if trace is None else
                                  f"W-trace: reverse loops + decoder over the {N}^3 thin-shell query lists of a trained-model-like field")
                               + (", end point E1 (grids resident in HBM)" if a.endpoint == "e1" else ", end point E2 (through marching cubes: every mesh finished inside the timed region)"),
                   "baseline_config": a.config, "endpoint": a.endpoint,
                   "shapes_per_gpu": B, "shapes_per_gpu_means": "shapes per step and GPU; the schedule keeps `shapes_in_flight` shapes of consecutive steps in flight on one GPU",
                   "latents_per_loop": latents_per_loop_cfg, "loops_in_flight": n_chains, "shapes_in_flight": shapes_in_flight,
                   "resolution": N, "diffusion_steps": T, "latent": a.latent,
                   "decoder_fwd_queries_per_shape": n_fwd, "decoder_grad_queries_per_shape": n_grad,
                   "decoder_precision": a.decoder_precision, "unet_precision": a.unet_precision,
                   "fp16_range_saturations": sat,
                   "schedule": a.schedule, "pipeline": sched,
                   "loop_driver": (a.loop_driver + (f" (one host thread, {a.loop_chunk} graph replay(s) per loop and turn)" if a.loop_driver == "interleaved" else " (one host thread per loop)")) if a.schedule == "phased" else "n/a",
                   "rounds": meas["rounds"], "rounds_means": "time-sliced rounds inside the timed region (loops of a round, then its grids); 1 = the whole region is one round",
                   "startup_s": meas["startup_s"],
                   "build_flags": build_config(),
                   "host_threads_per_rank": {"loop_chains": n_chains, "loop_driver_threads": 1 if (a.schedule == "phased" and a.loop_driver == "interleaved") else n_chains, "meshing": (mesh_stats or {}).get("threads", 0), "host_cores": os.cpu_count(), "ranks": world,
                                             "cpus_of_rank_0": _cpu_ranges(PINNED_CPUS) if PINNED_CPUS else "not pinned (one rank)"},
                   "parallelism": f"shape-parallel x{world}, no data-path collective (latents all_gathered after the timed region)"},
        "roofline": {"kernel": kname, "bound": "mfma",
                     "achieved": algorithmic, "peak": peak, "unit": "TFLOP/s", "frac": algorithmic / peak,
                     "traffic": (pmc or {}).get("decoder_fwd_hbm_bytes_per_launch"),
                     "traffic_from": pmc_note,
                     "algorithmic_bytes_per_launch": 16.0 * fwd_total / max(fwd_launches, 1),
                     "launches": fwd_launches, "avg_launch_ms": fwd_ms / max(fwd_launches, 1),
                     "flop_per_point": FWD_FLOP,
                     "issued_tflops": (3.0 if f16 else 1.0) * algorithmic, "issued_frac": (3.0 if f16 else 1.0) * algorithmic / peak,
                     # the kernel is power-bound: workgroup 0 of every launch measures the shader clock the chip held under it
                     # (shader cycles / 100 MHz ticks, surfd_decoder_sustained_clock); the matrix peak scales with it
                     "sustained_clock_ghz": meas["clock_ghz"] or None, "max_clock_ghz": MAX_CLOCK_GHZ,
                     "clock_limited_peak": (peak * meas["clock_ghz"] / MAX_CLOCK_GHZ) if meas["clock_ghz"] else None,
                     "frac_of_clock_limited_peak": (algorithmic / (peak * meas["clock_ghz"] / MAX_CLOCK_GHZ)) if meas["clock_ghz"] else None,
                     "issued_frac_of_clock_limited_peak": ((3.0 if f16 else 1.0) * algorithmic / (peak * meas["clock_ghz"] / MAX_CLOCK_GHZ)) if meas["clock_ghz"] else None,
                     "cus": (f"{a.decoder_blocks} of 256 while loops are in flight, 256 for the last round; peak is the whole chip's")
                            if a.schedule == "overlap" else "256"},
        "roofline_loop": roofline_loop,
        "time_share": ({"loops_alone_on_chip_ms": loop_phase_ms, "grids_ms": elapsed * 1e3 - loop_phase_ms,
                        "loops_frac": loop_phase_ms / (elapsed * 1e3)} if loop_phase_ms is not None else None),
        "breakdown_ms_per_step": {"reverse_loop_latency": loop_ms / max(loops, 1), "decoder_fwd": fwd_ms / a.steps,
                                  "decoder_fwd_bwd": grad_ms / a.steps,
                                  "decoder_fwd_bwd_tflops": (grad_total * 2 * FWD_FLOP) / (grad_ms * 1e-3) / 1e12 if grad_ms > 0 else 0.0},
    }
    if per_rank is not None:
        out["per_rank"] = per_rank
    if strict is not None:
        out["strict_c3"] = strict
    if trace_e2 is not None:
        out["trace_e2"] = trace_e2
    if w_trace is not None:
        per_shape_ms = w_trace["decoder_fwd_ms_per_shape"] + w_trace["decoder_fwd_bwd_ms_per_shape"]
        w_trace["note"] = ("decoder kernels alone on all CUs, values discarded; with this occupancy a step's grids take "
                           f"{per_shape_ms * B:.0f} ms against {loop_alone_ms / max(alone_n, 1):.0f} ms of reverse loop per step (one wide loop alone)")
        out["w_trace"] = w_trace
    if mesh_stats is not None:
        out["e2"] = {"status": "timed", "value": shapes / elapsed, "unit": "shapes/s", **mesh_stats}
    elif trace_e2 is not None:
        out["e2"] = {"status": "see trace_e2 (timed, trained-model-like field); the synthetic decoder's own field has almost no surface to mesh"}
    elif not a.no_e2:
        out["e2"] = e2_estimate(a, elapsed, shapes, world)
    if not a.no_cpu_baseline and world == 1:          # a stated baseline of the N=1 line only
        out["cpu_baseline"] = cpu_baseline(T, B, n_fwd, n_grad, a.latent, N)
    print(json.dumps(out))


def grid_shard_main(a, world, rank):
    """--mode grid-shard (north star: "shard the per-sample 512^3 grid evaluation across the GPUs"): the reverse loop of a
    step's B shapes is replicated (deterministic: every rank computes the same latents), then every refinement level of
    every shape's grid is split over the ranks.  Native path (default, GridFiller.fill_grid_sharded over surfd_grid_shard_*):
    the level's point list is voxel-ordered on every rank (ordered compaction on the device), rank r runs the decoder on the
    64-point tiles r, r + G, ... into a fixed-capacity value buffer, the buffers are summed over the ranks (ncclAllReduce
    over xGMI; every entry is non-zero on one rank), every rank commits the whole level — no host read anywhere between the
    latent and the finished grid.  Callback path (--shard-path callback, round 3): surfd_amd.parallel.ShardedField over
    GridFiller's callback API, one host read of the level's length per level.  Strong scaling of one shape's latency."""
    from surfd_amd import synth
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
    from surfd_amd.parallel import ShardedField
    cfg = CONFIGS[a.config]
    if cfg["cond_mode"] != "no_cond":
        raise SystemExit("--mode grid-shard is wired for the unconditional configurations (c2, c3)")
    model, diffusion, dec = build_models(cfg, a.latent, a.decoder_precision, a.unet_precision)
    if a.diffusion_steps != 1000:
        from surfd_amd.diffusion import create_gaussian_diffusion
        diffusion = create_gaussian_diffusion(types.SimpleNamespace(noise_schedule="cosine", sigma_small=True), f"ddim{a.diffusion_steps}")
    T, B, N = diffusion.num_timesteps, a.batch, a.resolution
    n_steps_max = max(a.steps, a.warmup, 1)
    noise_bank = [synth.synth_noise_batch(T, s * B, B, a.latent).cuda() for s in range(n_steps_max)]     # the same shapes on every rank
    filler = GridFiller(N)
    udf = torch.empty(N, N, N, device="cuda")
    grads = torch.empty(N, N, N, 3, device="cuda")
    fwd_pts = [0.0]

    def one_step(s):
        lat = diffusion.p_sample_loop(model, (B, 1, a.latent), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise_bank[s], fused=True)
        dec.bind_latents(lat.reshape(B, a.latent))
        for k in range(B):
            if a.shard_path == "native":
                # adaptive: the exchange buffers follow the fields (largest counts seen so far x 1.5 / 2): one read of the counts per
                # shape; a shape that does not fit (fields differ by more than that: round 5's first run with buffers sized from the
                # first step's shapes cut one) is repeated with larger ones INSIDE the timed region, on every rank alike
                filler.fill_grid_sharded(make_udf_func(dec, lat[k], sample=k), rank=rank, world=world, out=(udf, grads), stats=False,
                                         adaptive=not fixed_caps, capacity=a.shard_capacity if fixed_caps else None,
                                         grad_capacity=a.shard_grad_capacity if fixed_caps else None)
            else:
                filler.fill_grid(ShardedField(make_udf_func(dec, lat[k], sample=k)), 2 ** 22, out=(udf, grads), stats=True)
                fwd_pts[0] += sum(filler.last_stats["fwd_per_level"])
        return lat

    fixed_caps = a.shard_capacity > 0 and a.shard_grad_capacity > 0          # both given on the command line: fixed buffers, no read per shape
    if a.shard_path == "native" and not fixed_caps:
        # exchange-buffer capacities from the field itself, PER LEVEL: the B untimed shapes of the first step with generous
        # buffers, their device-side counts read once each; every rank sees the same grids, hence the same counts and the same
        # capacities (the collectives' sizes must agree).  Fields differ from shape to shape: twice the largest count seen per
        # level (four times for the near-surface gradient points, which vary most), in whole tiles of every rank, at least 2^16
        # — a thin-shell level travels as a few MB, not as the 64 MB of one global capacity.
        lat0 = diffusion.p_sample_loop(model, (B, 1, a.latent), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise_bank[0], fused=True)
        dec.bind_latents(lat0.reshape(B, a.latent))
        seen = []
        for k in range(B):
            filler.fill_grid_sharded(make_udf_func(dec, lat0[k], sample=k), rank=rank, world=world, out=(udf, grads), stats=True,
                                     capacity=min(1 << 26, 7 * (N // 2) ** 3), grad_capacity=min(1 << 25, N ** 3))
            seen.append(filler.last_stats)
        filler._shard_seen = {"fwd_per_level": [max(st["fwd_per_level"][l] for st in seen) for l in range(len(filler.N_levels))],
                              "grad": max(st["grad"] for st in seen)}
        filler.plan_shard_capacities([filler._shard_seen], world, margin=1.5, grad_margin=2.0)      # the adaptive fills continue from here
        filler._shard_buf = None
        torch.cuda.empty_cache()
    for s in range(a.warmup):
        one_step(s)
    if filler._handle is not None:
        filler.totals(reset=True)
        if a.shard_path == "native":
            filler.shard_overflows(reset=True)
    filler.shard_bytes_exchanged = 0
    fwd_pts[0] = 0.0
    barrier(world)
    t0 = time.perf_counter()
    for s in range(a.steps):
        lat = one_step(s)
    torch.cuda.synchronize()
    local_ms = (time.perf_counter() - t0) * 1e3
    barrier(world)
    elapsed = time.perf_counter() - t0               # the timed region ends HERE; what follows is measured on the side
    shard_caps = None
    if a.shard_path == "native":
        tot = filler.totals(reset=True)             # running totals kept on the device by the fills: one read, after the clock
        fwd_pts[0] = float(sum(tot["fwd_per_level"]))
        # capacity check over EVERY shape of the timed region: each commit compared its list with its buffer on the device
        # (counts differ from shape to shape — the last shape's counts say nothing about the others); one read, after the clock
        cut = filler.shard_overflows(reset=True)
        shard_caps = filler.shard_capacities(a.shard_capacity if fixed_caps else None, a.shard_grad_capacity if fixed_caps else None)
        assert cut == 0, (f"{cut} exchange buffer(s) of the timed region were shorter than their list (capacities {shard_caps}): the grids "
                          f"of this run are incomplete — raise --shard-capacity / --shard-grad-capacity")
    # what a rank RECEIVES per shape from the all-gathers (every level's gathered buffer + the gradient buffer, minus its own
    # segment: (world - 1) / world of them; with one rank nothing is exchanged and the figure is what 8 ranks would receive);
    # round 5 summed zero-filled point-indexed buffers instead (ring all-reduce: 2 x (world - 1) / world of the same buffers);
    # SURVEY.md section 8e sized the all-gather at <= 5 MB per thin-shell level
    w_acc = world if world > 1 else 8
    buffers_bytes = (4 * sum(shard_caps[0]) + 12 * shard_caps[1]) if shard_caps else None
    bytes_exchanged_per_shape = int(buffers_bytes * (w_acc - 1) / w_acc) if shard_caps else None
    # the same shapes through the fused single-rank fill, timed the same way: what the sharded path costs by construction
    fused_ms = None
    if rank == 0 or world > 1:
        ff = GridFiller(N)
        ff.fill_grid(make_udf_func(dec, lat[0], sample=0), 2 ** 16, out=(udf, grads), stats=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(B):
            ff.fill_grid(make_udf_func(dec, lat[k], sample=k), 2 ** 16, out=(udf, grads), stats=False)
        torch.cuda.synchronize()
        fused_ms = (time.perf_counter() - t1) * 1e3 / B
    shard_ms = None
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(B):
        if a.shard_path == "native":
            filler.fill_grid_sharded(make_udf_func(dec, lat[k], sample=k), rank=rank, world=world, out=(udf, grads), stats=False,
                                     adaptive=not fixed_caps, capacity=a.shard_capacity if fixed_caps else None,
                                     grad_capacity=a.shard_grad_capacity if fixed_caps else None)
        else:
            filler.fill_grid(ShardedField(make_udf_func(dec, lat[k], sample=k)), 2 ** 22, out=(udf, grads), stats=False)
    torch.cuda.synchronize()
    shard_ms = (time.perf_counter() - t1) * 1e3 / B
    per_rank = None
    if world > 1:
        import torch.distributed as dist
        mine = _host_if_gloo(torch.tensor([local_ms], device="cuda", dtype=torch.float64))
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "ms_per_step": float(v[0]) / a.steps} for r, v in enumerate(allr)]
        tt = _host_if_gloo(torch.tensor([elapsed], device="cuda", dtype=torch.float64))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        return
    shapes = B * a.steps                       # the SAME shapes on every rank: total work is fixed as N grows
    out = {"metric": "shapes/sec end-to-end (1000-step uncond, 512^3 UDF) at 1/2/4/8 GPU", "value": shapes / elapsed, "unit": "shapes/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32 (split-fp16 matrix products, fp32 accumulation)" if a.decoder_precision == "f16x2" else "f32",
           "data": "synthetic (seeded random-init weights, seeded noise)", "rccl_ranks": world,
           "config": {"workload": f"{cfg['name']}; grid-shard mode: reverse loop replicated, every level of every shape's {N}^3 grid split over {world} rank(s) "
                                  "by index range, values returned by ncclAllGather (end point E1)", "mode": "grid-shard", "baseline_config": a.config,
                      "shapes_per_step": B, "resolution": N, "diffusion_steps": T, "decoder_fwd_queries_per_shape": fwd_pts[0] / max(shapes, 1),
                      "shard_path": a.shard_path, "host_syncs_per_shape": (0 if fixed_caps else 1) if a.shard_path == "native" else len(filler.N_levels) + 1,
                      "exchange_capacities": ("fixed (command line)" if fixed_caps else "adaptive: largest counts seen so far x 1.5 (gradient points x 2), one read of the counts per shape, "
                                              "a shape that does not fit is repeated inside the timed region") if a.shard_path == "native" else None,
                      "exchange_capacity_points": {"per_level": shard_caps[0], "gradients": shard_caps[1]} if a.shard_path == "native" else None,
                      "exchange_bytes_per_shape": bytes_exchanged_per_shape if a.shard_path == "native" else None,
                      "exchange_bytes_per_shape_means": (f"bytes a rank receives per shape from the all-gathers at {w_acc} ranks ((world - 1) / world of every gathered buffer); "
                                                         f"round 5's ring all-reduce of zero-filled buffers of the same capacities moved {int(2 * buffers_bytes * (w_acc - 1) / w_acc) if shard_caps else None}") if a.shard_path == "native" else None,
                      "exchange_buffers_cut_in_timed_region": 0 if a.shard_path == "native" else None,
                      "parallelism": (f"grid-shard x{world}: rank r evaluates tiles r, r + {world}, ... of every level's voxel-ordered list; fixed-capacity value "
                                      "segments of capacity / world points all-gathered (ncclAllGather over xGMI: all_gather_into_tensor), 4 B per point per level + 12 B per gradient point")
                                     if a.shard_path == "native" else
                                     f"grid-shard x{world}: all_gather of 4 B per point per level + 12 B per gradient point over xGMI"},
           "grid_ms_per_shape": {"sharded": shard_ms, "fused_single_rank_fill": fused_ms,
                                 "sharded_over_fused": (shard_ms / fused_ms) if fused_ms else None,
                                 "note": "the shapes of the last step again, grids only, after the timed region"}}
    if per_rank is not None:
        out["per_rank"] = per_rank
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
