#!/usr/bin/env python3
"""Times the fused 1000-step reverse loop (B shapes) in both conv precisions.  python tools/loop_time.py [B] [L] [T]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["f16x2", "fp32"]
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, diff = create_model_and_diffusion(args)
if T != 1000:
    from surfd_amd.diffusion import create_gaussian_diffusion
    diff = create_gaussian_diffusion(args, f"ddim{T}")
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
noise = synth.synth_noise_batch(diff.num_timesteps, 0, B, L).cuda()
outs = {}
for mode in modes:
    model.set_precision(mode)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = diff.p_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{mode} run {it}: {t2 - t0:.3f}s for {diff.num_timesteps} steps = {(t2 - t0) / diff.num_timesteps * 1e3:.3f} ms/eval", flush=True)
    outs[mode] = out.clone()
    print(mode, "saturation count", model.saturation_count())
if len(outs) == 2:
    a, b = outs.values()
    print("max |x_final(f16x2) - x_final(fp32)|", (a - b).abs().max().item())
