"""Is the fused reverse loop host-bound?  Compares the time surfd_sample_loop takes to RETURN (host
enqueue) with the time until the stream drains.   python tools/debug_loop_host.py [B] [wide design batch] [T,T,...]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, diff = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
WIDE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
Ts = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1000]
model.set_wide(WIDE)
for T in Ts:          # a short loop fits the hardware queue: its enqueue time is pure host cost per graph launch
    if T != 1000:
        from surfd_amd.diffusion import create_gaussian_diffusion
        diff = create_gaussian_diffusion(args, f"ddim{T}")
    noise = synth.synth_noise_batch(diff.num_timesteps, 0, B, 32).cuda()
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = diff.p_sample_loop(model, (B, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"B {B} T {diff.num_timesteps} run {it}: enqueue returned after {(t1 - t0) * 1e3:.1f} ms ({(t1 - t0) / diff.num_timesteps * 1e6:.0f} us per graph launch), "
              f"stream drained after {(t2 - t0) * 1e3:.1f} ms", flush=True)
