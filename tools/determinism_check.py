#!/usr/bin/env python3
"""Run-to-run bit stability of one denoiser evaluation: the same input through the same handle N times, how many distinct
outputs.  python tools/determinism_check.py [N] [B] [wide design batch, 0 = latency form]   (SURFD_LIB selects the build)"""
import hashlib, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cases = [(8, 0), (80, 80)] if len(sys.argv) < 3 else [(int(sys.argv[2]), int(sys.argv[3]))]
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
for B, wide in cases:
    model.set_wide(wide)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 1, 32, generator=g).cuda(); t = torch.full((B,), 500, device="cuda")
    seen = {}
    first = None
    for i in range(N):
        out = model(x, t, y={})
        torch.cuda.synchronize()
        h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]
        seen[h] = seen.get(h, 0) + 1
        if first is None:
            first = out.clone()
        elif h != list(seen)[0] and "shown" not in seen:
            d = (out - first).abs()
            print(f"  B={B} wide={wide}: run {i} differs from run 0: max |d| = {float(d.max()):.3e}, {int((d > 0).sum())} of {d.numel()} values, samples touched {sorted(set(torch.nonzero(d.flatten(1).sum(1)).flatten().tolist()))[:12]}")
            seen["shown"] = 0
    seen.pop("shown", None)
    print(f"lib={os.path.basename(os.environ.get('SURFD_LIB', 'default'))} B={B} wide={wide}: {len(seen)} distinct output(s) in {N} runs {sorted(seen.values(), reverse=True)}", flush=True)
