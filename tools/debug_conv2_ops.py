#!/usr/bin/env python3
"""GPU debugging aid for the f16x2 conv kernel: runs the denoiser with exactly ONE conv op on the f16x2 kernel
(all others exact fp32) for every op and prints the output error against the all-fp32 evaluation, then the
all-f16x2 error and per-evaluation timings of both modes.   python tools/debug_conv2_ops.py [B] [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import _native as N, synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
import types

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine",
                             sigma_small=True, clip_value=1.0)
model, diff = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict())
model.cuda().eval()
lib, h = model._native()
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 1, L, generator=g).cuda()
t = torch.randint(0, 1000, (B,), generator=g).cuda()
model.set_precision("fp32")
ref = model(x, t, y={}).clone()
torch.cuda.synchronize()
print("fp32 out absmax", ref.abs().max().item())
model.set_precision("f16x2")
bad = []
for op in range(98):
    N.check(lib.surfd_unet_debug_only_op(h, op))
    try:
        out = model(x, t, y={})
        torch.cuda.synchronize()
        err = (out - ref).abs().max().item()
    except Exception as e:
        err = float("nan"); print("op", op, "EXC", e)
    flag = "" if err < 2e-5 else "   <<<<<<"
    if flag: bad.append(op)
    print(f"op {op:3d} err {err:.3e}{flag}")
N.check(lib.surfd_unet_debug_only_op(h, -1))
out = model(x, t, y={})
torch.cuda.synchronize()
print("ALL f16x2 err", (out - ref).abs().max().item(), "sat", model.saturation_count(), "bad ops", bad)
for mode in ["fp32", "f16x2"]:
    model.set_precision(mode)
    for _ in range(3): model(x, t, y={})
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(50): model(x, t, y={})
    torch.cuda.synchronize()
    print(mode, "ms/forward (eager, incl. embedding prep)", (time.time() - t0) / 50 * 1e3)
