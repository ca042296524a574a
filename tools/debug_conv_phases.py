"""Developer aid: per-phase shader-clock breakdown of every conv launch of one denoiser evaluation.
Run with SURFD_CONV_DEBUG=1."""
import ctypes as C, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import _native as N, synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
x = torch.randn(8, 1, 32, device="cuda"); t = torch.full((8,), 500, device="cuda")
for _ in range(3):
    model(x, t, y={})
torch.cuda.synchronize()
L, h = model._native()
buf = (C.c_longlong * (4096 * 16))()
L.surfd_unet_debug_read(h, buf, 4096)          # discard warm-up
model(x, t, y={}); torch.cuda.synchronize()
n = L.surfd_unet_debug_read(h, buf, 4096)
print("launch Cout Cin Lout blocks KS bch |", " ".join(f"{n_:>8s}" for n_ in ["toChunk", "xload", "GN", "slabwr", "mfma", "xwave", "publish", "reduce", "epilog", "total"]))
for i in range(n):
    s = buf[i * 16:(i + 1) * 16]
    st = list(s[:10]); meta = s[10:16]
    for k in range(1, 10):
        if st[k] == 0: st[k] = st[k - 1]
    d = [st[k + 1] - st[k] for k in range(9)]
    print(f"{i:4d} {meta[0]:5d} {meta[1]:5d} {meta[2]:3d} {meta[3]:5d} {meta[4]:3d} {meta[5]:3d} |", " ".join(f"{v:8d}" for v in d), f"{st[9]-st[0]:8d}")
