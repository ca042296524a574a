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

Batches are software-pipelined (--pipeline 1, default): the reverse loop of batch s+1 — a chain of ~115 000
dependent, latency-bound launches — runs on its own HIP stream on half of the CUs while the grids of batch s
(matrix-pipe/power bound) are evaluated on the other half; every one of the K batches runs start to finish
inside the timed region (pipeline fill and drain included).  --pipeline 0 runs loop then grids on one stream.

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
    ap.add_argument("--decoder-blocks", type=int, default=160,
                    help="with --pipeline 1: persistent decoder workgroups per launch while the next batch's reverse loop "
                         "runs on the remaining CUs (the last batch's grids, with nothing left to overlap, use every CU)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1: overlap the reverse loops of the next batches with the grid evaluation of the current one")
    ap.add_argument("--loop-chains", type=int, default=2,
                    help="with --pipeline 1: reverse loops (of different batches) in flight at once, each on its own stream "
                         "and execution context (MDM.replica)")
    ap.add_argument("--unet-precision", choices=["f16x2", "fp32"], default="f16x2",
                    help="denoiser conv arithmetic (include/surfd_hip.h: surfd_unet_set_precision)")
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


def cpu_baseline(T, B, n_fwd, n_grad):
    """The oracle (CPU restatement of the reference's op graph) timed on this host on a bounded
    sample of the same workload, scaled to shapes/s: a few denoiser steps at batch B, 16k forward
    decoder queries, 4k forward+backward queries (the reference's faster chunk size)."""
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
    t0 = time.time()
    odec.sample_udf(f, pts, 16384)
    fwd_rate = 16384 / (time.time() - t0)
    odec.sample_grads(f, pts[:1024], 1024)
    t0 = time.time()
    odec.sample_grads(f, pts[:4096], 4096)
    grad_rate = 4096 / (time.time() - t0)
    per_shape = T * step_s / B + n_fwd / fwd_rate + n_grad / grad_rate
    return {"value": 1.0 / per_shape, "unit": "shapes/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_it} denoiser steps at batch {B} ({step_s * 1e3:.0f} ms/step), 16384 decoder forward queries "
                      f"({fwd_rate:.0f} pts/s), 4096 forward+backward queries ({grad_rate:.0f} pts/s); extrapolated to "
                      f"{T} steps + {n_fwd:.0f} fwd + {n_grad:.0f} grad queries per shape (grid bookkeeping excluded)"}


def main():
    a = parse()
    world, rank, local = setup_dist(a.gpus)
    from surfd_amd import _native as Nn
    from surfd_amd import synth
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
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
    udf = [torch.empty(N, N, N, device="cuda") for _ in range(B)]
    grads = [torch.empty(N, N, N, 3, device="cuda") for _ in range(B)]
    stats = []

    chains = [model] + [model.replica() for _ in range(max(1, a.loop_chains) - 1)] if a.pipeline else [model]
    for m in chains[1:]:
        m.set_precision(a.unet_precision)

    def sample_latents(chain=0):
        return diffusion.p_sample_loop(chains[chain], (B, 1, a.latent), clip_denoised=False, model_kwargs={"y": {}},
                                       noise_stream=noise, fused=True)

    def fill_grids(lat, collect=False):
        dec.bind_latents(lat.reshape(B, a.latent))
        for k in range(B):
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
        """k_steps full passes (every batch: reverse loop + 8 grids), start to finish.  Pipelined mode
        (surfd_amd.parallel.BatchPipeline): batch s+1's reverse loop (latency-bound, few CUs) runs on its own
        stream while batch s's grids (matrix-pipe bound) are evaluated; shapes are independent, so this is the
        steady state of a sampling service."""
        if not a.pipeline:
            for _ in range(k_steps):
                one_step()
        elif k_steps:
            pipe.run(k_steps)

    run_steps(a.warmup)
    barrier(world)
    L.surfd_profile_enable(1)
    t0 = time.perf_counter()
    run_steps(a.steps)
    barrier(world)
    elapsed = time.perf_counter() - t0
    L.surfd_profile_enable(0)
    prof = {}
    for kind, name in [(0, "dec_fwd"), (1, "dec_grad"), (2, "loop")]:
        n, ms = C.c_int64(), C.c_double()
        Nn.check(L.surfd_profile_read(kind, C.byref(n), C.byref(ms)))
        prof[name] = (n.value, ms.value)
    # workload counters (one extra, untimed pass; the counters are deterministic)
    dec.set_grid_blocks(0)
    one_step(collect=True)
    torch.cuda.synchronize()
    n_fwd = sum(sum(s["fwd_per_level"]) for s in stats) / B
    n_grad = sum(s["grad"] for s in stats) / B
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        return
    shapes = world * B * a.steps
    fwd_launches, fwd_ms = prof["dec_fwd"]
    grad_launches, grad_ms = prof["dec_grad"]
    fwd_flop_total = n_fwd * B * a.steps * FWD_FLOP            # this rank, timed region
    algorithmic = fwd_flop_total / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
    if a.decoder_precision == "f16x2":
        # every algorithmic multiply-add is issued as three fp16 MFMA products (xh*wh + xh*wl + xl*wh):
        # price the kernel at what it issues against the fp16 matrix peak
        kname = "decoder_kernel<false, f16x2> (fused encode + 11-layer CBN MLP + sigmoid; split-fp16 operands, fp32 accumulate)"
        achieved, peak = 3.0 * algorithmic, F16_MFMA_PEAK_TF
        dtype = "f32 (decoder matrix products as split fp16x2 on the fp16 MFMA pipe with fp32 accumulation; denoiser fp32 MFMA)"
    else:
        kname = "decoder_kernel<false> (fused encode + 11-layer CBN MLP + sigmoid)"
        achieved, peak = algorithmic, FP32_MFMA_PEAK_TF
        dtype = "f32"
    out = {
        "metric": "shapes/sec end-to-end (1000-step uncond, 512^3 UDF) at 1/2/4/8 GPU",
        "value": shapes / elapsed, "unit": "shapes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic (seeded random-init weights of the reference architectures, seeded noise)",
        "config": {"workload": f"unconditional, {T}-step {'DDPM' if a.diffusion_steps == 1000 else 'DDIM'}, L={a.latent}, "
                               f"{N}^3 coarse-to-fine UDF grid + gradients (end point E1: grids resident in HBM), "
                               f"{B} shapes/GPU (BASELINE configs[2] per-GPU shard)",
                   "shapes_per_gpu": B, "resolution": N, "diffusion_steps": T,
                   "decoder_fwd_queries_per_shape": n_fwd, "decoder_grad_queries_per_shape": n_grad,
                   "decoder_precision": a.decoder_precision,
                   "pipeline": ("reverse loop of batch s+1 overlaps the grids of batch s on two HIP streams; the decoder "
                                f"kernels run on {a.decoder_blocks} of the CUs while a loop is in flight and on all of them "
                                "for the last batch; every batch runs start to finish inside the timed region")
                               if a.pipeline else "none (loop then grids, one stream)",
                   "parallelism": f"shape-parallel x{world}, no data-path collective"},
        "roofline": {"kernel": kname, "bound": "mfma",
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": None, "launches": fwd_launches, "avg_launch_ms": fwd_ms / max(fwd_launches, 1),
                     "flop_per_point": FWD_FLOP, "algorithmic_tflops": algorithmic,
                     "cus": (f"{a.decoder_blocks} of 256 for {a.steps - 1} of {a.steps} batches (the rest run the next batch's "
                             "reverse loop), 256 for the last; peak is the whole chip's") if a.pipeline else "256",
                     "mfma_flop_per_point": (3 if a.decoder_precision == "f16x2" else 1) * FWD_FLOP},
        "breakdown_ms_per_step": {"reverse_loop": prof["loop"][1] / a.steps, "decoder_fwd": fwd_ms / a.steps,
                                  "decoder_fwd_bwd": grad_ms / a.steps,
                                  "decoder_fwd_bwd_tflops": (n_grad * B * a.steps * 2 * FWD_FLOP) / (grad_ms * 1e-3) / 1e12 if grad_ms > 0 else 0.0},
    }
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(T, B, n_fwd, n_grad)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
