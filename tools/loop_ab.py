#!/usr/bin/env python3
"""One library build (SURFD_LIB=...) through the reverse-loop timings that matter: the latency form at B latents (what
sample/generate_* runs), one wide loop of W latents alone and two at once (the bench's schedule).  One JSON line on stdout.
python tools/loop_ab.py [T steps per loop] [B] [W] [design batch] [L]"""
import json, os, sys, threading, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
from surfd_amd.diffusion import create_gaussian_diffusion
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = int(sys.argv[3]) if len(sys.argv) > 3 else 80
DESIGN = int(sys.argv[4]) if len(sys.argv) > 4 else 80
L = int(sys.argv[5]) if len(sys.argv) > 5 else 32
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
diff = create_gaussian_diffusion(args, f"ddim{T}")
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
chains = [model, model.replica()]
streams = [torch.cuda.Stream(priority=-1) for _ in range(2)]
dev = torch.cuda.current_device()


def run(nchain, width, wide):
    for m in chains[:nchain]:
        m.set_wide(wide)
    noise = [synth.synth_noise_batch(diff.num_timesteps, q * width, width, L).cuda() for q in range(nchain)]
    outs = [None] * nchain

    def worker(q):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[q]):
            outs[q] = diff.p_sample_loop(chains[q], (width, 1, L), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise[q], fused=True)

    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(q,)) for q in range(nchain)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it:
            best = min(best, dt)
    return best / diff.num_timesteps * 1e3, float(outs[0].double().abs().sum())


res = {"lib": os.path.basename(os.environ.get("SURFD_LIB", "default")), "T": T}
ms, cs = run(1, B, 0); res[f"latency_B{B}_ms"] = round(ms, 4); res["latency_checksum"] = cs
ms, cs = run(1, W, DESIGN); res[f"wide_1x{W}_ms"] = round(ms, 4); res["wide_checksum"] = cs
ms, _ = run(2, W, DESIGN); res[f"wide_2x{W}_ms"] = round(ms, 4); res[f"wide_2x{W}_us_per_eval_latent"] = round(ms * 1e3 / (2 * W), 2)
res["saturation"] = int(model.saturation_count())
print(json.dumps(res), flush=True)
