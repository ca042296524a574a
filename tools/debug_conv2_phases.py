"""Developer aid: phase breakdown (us) of every f16x2 conv launch of one denoiser evaluation (workgroup 0).
python tools/debug_conv2_phases.py [B] [wide design batch].  Needs a stamps build (python tools/build_variants.py
stamps=conv_f16x2.hip:-DSURFD_C2_STAMPS, selected with SURFD_LIB=...); run with SURFD_CONV_DEBUG=1."""
import ctypes as C, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import _native as N, synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
WIDE = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # wide form with this design batch (0: latency form)
model.set_wide(WIDE)
x = torch.randn(B, 1, 32, device="cuda"); t = torch.full((B,), 500, device="cuda")
for _ in range(3):
    model(x, t, y={})
torch.cuda.synchronize()
L, h = model._native()
buf = (C.c_longlong * (4096 * 16))()
L.surfd_unet_debug_read(h, buf, 4096)          # discard warm-up
model(x, t, y={}); torch.cuda.synchronize()
n = L.surfd_unet_debug_read(h, buf, 4096)
names = ["issue", "v+stat", "combine", "slab", "mfma1", "rest", "reduce", "splitK", "epilog", "drain"]
print("launch Cout  Cin Lout  WGs KS nch |", " ".join(f"{n_:>7s}" for n_ in names), "  total")
tot = [0.0] * len(names)
for i in range(n):
    s = buf[i * 16:(i + 1) * 16]
    st = list(s[:11]); meta = [s[11], s[12], s[13] // 100000, s[13] % 100000, s[14]]; cyc = s[15]
    if st[0] == 0: continue
    for k in range(1, 11):
        if st[k] == 0: st[k] = st[k - 1]
    d = [(st[k + 1] - st[k]) / 100.0 for k in range(10)]
    for k in range(10): tot[k] += d[k]
    print(f"{i:4d} {meta[0]:5d} {meta[1]:5d} {meta[2]:3d} {meta[3]:5d} {meta[4]//100:2d} {meta[4]%100:3d} |", " ".join(f"{v:7.2f}" for v in d), f"{(st[10]-st[0])/100.0:8.2f}  {cyc / max(st[10]-st[0], 1) / 10.0:5.2f} GHz")
print("sum".ljust(32), "|", " ".join(f"{v:7.1f}" for v in tot), f"{sum(tot):8.1f}")
