#!/usr/bin/env python3
"""First layer whose output differs run to run: every f16x2 conv launch's output is checksummed on the device
(SURFD_CONV_DEBUG=1 SURFD_CONV2_HASH=1) and the per-launch checksums of N evaluations of the same input are compared.
python tools/determinism_layers.py [N] [B] [wide design batch] [L]"""
import ctypes as C, os, sys, types
os.environ["SURFD_CONV_DEBUG"] = "1"; os.environ["SURFD_CONV2_HASH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 80
WIDE = int(sys.argv[3]) if len(sys.argv) > 3 else 80
LEN = int(sys.argv[4]) if len(sys.argv) > 4 else 32
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
model.set_wide(WIDE)
L, h = model._native()
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, LEN, generator=g).cuda(); t = torch.full((B,), 500, device="cuda")
buf = (C.c_longlong * (4096 * 16))()
model(x, t, y={}); torch.cuda.synchronize(); L.surfd_unet_debug_read(h, buf, 4096)
ref = None
firsts = {}
every = {}
for run in range(N):
    model(x, t, y={}); torch.cuda.synchronize()
    n = L.surfd_unet_debug_read(h, buf, 4096)
    rec = [(buf[i * 16], buf[i * 16 + 11], buf[i * 16 + 12], buf[i * 16 + 13] // 100000, buf[i * 16 + 13] % 100000, buf[i * 16 + 14]) for i in range(n)]
    if ref is None:
        ref = rec; continue
    first = True
    for i, (a, b) in enumerate(zip(ref, rec)):
        if a[0] != b[0]:
            key = (i, a[1:])
            if first:
                firsts[key] = firsts.get(key, 0) + 1
                first = False
            every[key] = every.get(key, 0) + 1
print(f"{N - 1} runs compared with run 0; first differing launch (index, Cout, Cin, Lout, tiles x chunks x KS, KS*100+nch): count")
for k, v in sorted(firsts.items()):
    print("  ", k, v)
if not firsts:
    print("   none: every launch's checksum is identical in every run")
else:
    print(f"every differing launch ({len(ref)} launches per evaluation; a launch downstream of a differing one differs too):")
    for k, v in sorted(every.items())[:24]:
        print("  ", k, v)
