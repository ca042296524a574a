#!/usr/bin/env python3
"""K wide reverse loops at once, driven (a) by K host threads (one surfd_sample_loop call each, what
PhasedPipeline did through round 4) or (b) by ONE host thread that hands `chunk` graph replays at a time to each loop's stream in
turn (SpacedDiffusion.fused_loops_interleaved over surfd_sample_loop_begin / _run / _end).  For (K, width) in 2 x 80, 3 x 53,
4 x 40: `reps` consecutive timed runs of each driver, in us per (evaluation, latent), and whether the interleaved results equal
the threaded ones bit for bit.  One JSON line on stdout.
python tools/loop_interleave_sweep.py [T steps per loop] [reps] [L]        (CONFIGS=2x80,4x80 CHUNKS=1 DESIGN=80 to change the sweep)"""
import json, os, sys, threading, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
from surfd_amd.diffusion import create_gaussian_diffusion
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
CONFIGS = [tuple(int(v) for v in c.split("x")) for c in os.environ.get("CONFIGS", "2x80,3x53,4x40").split(",")]      # loops x width
CHUNKS = [int(c) for c in os.environ.get("CHUNKS", "1,4").split(",")]
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
diff = create_gaussian_diffusion(args, f"ddim{T}")
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
KMAX = max(k for k, _ in CONFIGS)
chains = [model] + [model.replica() for _ in range(KMAX - 1)]
streams = [torch.cuda.Stream() for _ in range(KMAX)]
dev = torch.cuda.current_device()


def threaded(K, width, noise):
    outs = [None] * K

    def worker(q):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[q]):
            outs[q] = diff.p_sample_loop(chains[q], (width, 1, L), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise[q], fused=True)

    th = [threading.Thread(target=worker, args=(q,)) for q in range(K)]
    [t.start() for t in th]; [t.join() for t in th]
    return outs


def interleaved(K, width, noise, chunk):
    jobs = [{"model": chains[q], "shape": (width, 1, L), "noise_stream": noise[q], "stream": streams[q], "model_kwargs": {"y": {}}} for q in range(K)]
    return diff.fused_loops_interleaved(jobs, sampler="ddpm", clip_denoised=False, chunk=chunk)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, outs


res = {"lib": os.path.basename(os.environ.get("SURFD_LIB", "default")), "T": T, "L": L, "reps": REPS, "unit": "us per (evaluation, latent)", "configs": {}}
for K, width in CONFIGS:
    for m in chains[:K]:
        m.set_wide(int(os.environ.get("DESIGN", width)))
    noise = [synth.synth_noise_batch(diff.num_timesteps, q * width, width, L).cuda() for q in range(K)]
    row = {}
    drivers = [("threads", lambda: threaded(K, width, noise))] + [(f"one_thread_chunk{c}", (lambda c=c: interleaved(K, width, noise, c))) for c in CHUNKS]
    ref = None
    for name, fn in drivers:
        timed(fn)                                                     # untimed: graph capture, workspace
        runs = []
        for _ in range(REPS):
            dt, outs = timed(fn)
            runs.append(round(dt / diff.num_timesteps * 1e6 / (K * width), 2))
        row[name] = runs
        if ref is None:
            ref = [o.clone() for o in outs]
        else:
            row[name + "_bit_equal_to_threads"] = all(torch.equal(a, b) for a, b in zip(ref, outs))
    res["configs"][f"{K}x{width}"] = row
res["saturation"] = int(model.saturation_count())
if "2x80" in res["configs"]:
    base = max(res["configs"]["2x80"]["threads"])
    res["within_5pct_of_2x80_threads"] = {k: {n: max(v) <= 1.05 * base for n, v in r.items() if isinstance(v, list)} for k, r in res["configs"].items()}
print(json.dumps(res), flush=True)
