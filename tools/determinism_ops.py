#!/usr/bin/env python3
"""Which conv op is not bit-stable run to run: the denoiser with exactly ONE conv op on the f16x2 kernel (the others on the exact
fp32 kernel), N evaluations of the same input each.   python tools/determinism_ops.py [N] [B] [wide design batch]"""
import hashlib, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import _native as Nn, synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 80
WIDE = int(sys.argv[3]) if len(sys.argv) > 3 else 80
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
model.set_wide(WIDE)
lib, h = model._native()
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, 32, generator=g).cuda(); t = torch.full((B,), 500, device="cuda")
model(x, t, y={})
RANGES = [(int(a), int(b)) for a, b in (r.split("-") for r in sys.argv[4].split(","))] if len(sys.argv) > 4 else [(k, k) for k in range(90)]
for first, last in RANGES:
    op = f"{first}-{last}"
    Nn.check(lib.surfd_unet_debug_only_op(h, first | (last << 16) if last != first else first))
    seen = {}
    for i in range(N):
        out = model(x, t, y={}); torch.cuda.synchronize()
        k = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:10]
        seen[k] = seen.get(k, 0) + 1
    if len(seen) > 1 or len(RANGES) < 90:
        print(f"ops {op}: {len(seen)} distinct outputs in {N} runs", flush=True)
Nn.check(lib.surfd_unet_debug_only_op(h, -1))
print("done")
