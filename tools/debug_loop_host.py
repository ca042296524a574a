"""Is the fused reverse loop host-bound?  Compares the time surfd_sample_loop takes to RETURN (host
enqueue) with the time until the stream drains."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, diff = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
noise = synth.synth_noise_batch(1000, 0, 8, 32).cuda()
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = diff.p_sample_loop(model, (8, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"run {it}: enqueue returned after {t1 - t0:.3f}s, stream drained after {t2 - t0:.3f}s")
