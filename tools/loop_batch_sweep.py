#!/usr/bin/env python3
"""One fused 1000-step reverse loop alone on the chip at several batch widths: ms per evaluation and the HBM weight
stream it amounts to.  python tools/loop_batch_sweep.py [L] [B,B,...] [T] [wide design batch, 0 = latency form]   -> markdown table on stdout"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Bs = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16, 24, 32]
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
WIDE = int(sys.argv[4]) if len(sys.argv) > 4 else 0
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, diff = create_model_and_diffusion(args)
if T != 1000:
    from surfd_amd.diffusion import create_gaussian_diffusion
    diff = create_gaussian_diffusion(args, f"ddim{T}")
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
model.set_wide(WIDE)
print(f"conv form: {'wide, design batch %d' % WIDE if WIDE else 'latency'}; {diff.num_timesteps} steps per loop\n")
WEIGHT_BYTES = 553294340
print(f"| L | B | ms / evaluation | us / (evaluation x latent) | weight stream GB/s | frac of 8 TB/s |")
print("|---|---|---|---|---|---|")
for B in Bs:
    noise = synth.synth_noise_batch(diff.num_timesteps, 0, B, L).cuda()
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = diff.p_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it:
            best = min(best, dt)
    ms = best / diff.num_timesteps * 1e3
    gbs = WEIGHT_BYTES / (ms * 1e-3) / 1e9
    print(f"| {L} | {B} | {ms:.3f} | {ms * 1e3 / B:.1f} | {gbs:.0f} | {gbs / 8000:.3f} |", flush=True)
    assert model.saturation_count() == 0
