#!/usr/bin/env python3
"""One module of the denoiser (its conv launches back to back) on a fixed input, N times: which output elements change run to run.
python tools/determinism_module.py [module] [Cin] [Cout] [Lin] [N] [B] [wide design batch] [L of the model]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import _native as Nn, synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
mod = sys.argv[1] if len(sys.argv) > 1 else "middle_block.0"
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 896
Cout = int(sys.argv[3]) if len(sys.argv) > 3 else 896
Lin = int(sys.argv[4]) if len(sys.argv) > 4 else 4
N = int(sys.argv[5]) if len(sys.argv) > 5 else 200
B = int(sys.argv[6]) if len(sys.argv) > 6 else 80
WIDE = int(sys.argv[7]) if len(sys.argv) > 7 else 80
LM = int(sys.argv[8]) if len(sys.argv) > 8 else 32
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
model.set_wide(WIDE)
L, h = model._native()
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, LM, generator=g).cuda(); t = torch.full((B,), 500, device="cuda")
model(x, t, y={}); torch.cuda.synchronize()          # embedding rows, workspace
xin = torch.randn(B, Cin, Lin, generator=g).cuda().contiguous()
out = torch.empty(B, Cout, Lin, device="cuda")
ref = None
ndiff = 0
for i in range(N):
    Nn.check(L.surfd_unet_debug_run_module(h, mod.encode(), Nn.ptr(xin), Cin, Lin, Nn.ptr(out), Cout, Lin, B, LM, Nn.stream()))
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone(); continue
    d = (out != ref)
    if d.any():
        ndiff += 1
        if ndiff <= 6:
            idx = torch.nonzero(d)
            bs = sorted(set(idx[:, 0].tolist())); cs = sorted(set(idx[:, 1].tolist())); ls = sorted(set(idx[:, 2].tolist()))
            print(f"run {i}: {int(d.sum())} elements differ, max |d| {float((out - ref).abs().max()):.3e}; samples {bs[:16]}; channels {len(cs)} in [{cs[0]}, {cs[-1]}] "
                  f"(tiles {sorted(set(c // 32 for c in cs))[:12]}); positions {ls}", flush=True)
            b0 = bs[0]
            e = (out[b0] - ref[b0]).abs()
            print(f"   sample {b0}: per-channel max |d| (nonzero): {[(c, round(float(e[c].max()), 7)) for c in range(e.shape[0]) if float(e[c].max()) > 0][:24]}", flush=True)
print(f"{mod}: {ndiff} of {N - 1} runs differ from run 0")
