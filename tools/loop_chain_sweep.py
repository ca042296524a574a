#!/usr/bin/env python3
"""Several fused reverse loops at once on the chip (one stream + execution context + host thread each, as
surfd_amd.parallel.BatchPipeline runs them): aggregate cost per (evaluation x latent) for chains x batch width.
python tools/loop_chain_sweep.py [L] [chains:B,chains:B,...] [T] [wide design batch]   -> markdown table"""
import os, sys, threading, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cases = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[2] if len(sys.argv) > 2 else "1:32,2:32,3:32,2:64,3:64").split(",")]
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
WIDE = int(sys.argv[4]) if len(sys.argv) > 4 else 32
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, diff = create_model_and_diffusion(args)
if T != 1000:
    from surfd_amd.diffusion import create_gaussian_diffusion
    diff = create_gaussian_diffusion(args, f"ddim{T}")
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
maxc = max(c for c, _ in cases)
chains = [model] + [model.replica() for _ in range(maxc - 1)]
for m in chains:
    m.set_wide(WIDE)
streams = [torch.cuda.Stream(priority=-1) for _ in range(maxc)]
print(f"conv form: {'wide, design batch %d' % WIDE if WIDE else 'latency'}; {diff.num_timesteps} steps per loop\n")
print("| L | chains | B per chain | wall ms / evaluation | us / (evaluation x latent) |")
print("|---|---|---|---|---|")
dev = torch.cuda.current_device()
for nchain, B in cases:
    noise = [synth.synth_noise_batch(diff.num_timesteps, q * B, B, L).cuda() for q in range(nchain)]

    def worker(q):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[q]):
            diff.p_sample_loop(chains[q], (B, 1, L), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise[q], fused=True)

    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(q,)) for q in range(nchain)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it:
            best = min(best, dt)
    ms = best / diff.num_timesteps * 1e3
    print(f"| {L} | {nchain} | {B} | {ms:.3f} | {ms * 1e3 / (B * nchain):.1f} |", flush=True)
