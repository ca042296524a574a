#!/usr/bin/env python3
"""Forward / forward+reverse decoder kernels alone on the chip: algorithmic TFLOP/s, and the f16x2 result against the
exact-fp32 kernel on the same points.  SURFD_LIB=<variant .so> python tools/dec_time.py [log2 points]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.cbndec import CbnDecoder
from surfd_amd.spec import DecoderConfig
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
dec = CbnDecoder(63, 32, 512, 5)
dec.load_state_dict(synth.synth_decoder_state_dict(DecoderConfig(latent_dim=32)), strict=True)
dec = dec.cuda().eval()
lat = (torch.randn(2, 32, generator=torch.Generator().manual_seed(1)) * 0.8).cuda()
dec.bind_latents(lat)
g = torch.Generator().manual_seed(2)
pts = (torch.rand(1 << lg, 3, generator=g) * 2 - 1).cuda()
gp = pts[: 1 << (lg - 2)].contiguous()
FWD = 5308416


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


dec.set_precision("fp32")
u32 = dec.udf(pts[: 1 << 18], 1).clone()
_, n32 = dec.udf_and_ngrad(gp[: 1 << 16], 1)
n32 = n32.clone()
dec.set_precision("f16x2")
u16 = dec.udf(pts[: 1 << 18], 1)
_, n16 = dec.udf_and_ngrad(gp[: 1 << 16], 1)
cos = (n32 * n16).sum(-1)
import hashlib
digest = hashlib.sha256(u16.cpu().numpy().tobytes()).hexdigest()[:12]
f_ms = timed(lambda: dec.udf(pts, 0))
g_ms = timed(lambda: dec.udf_and_ngrad(gp, 0))
print(f"{os.path.basename(os.environ.get('SURFD_LIB', 'default')):28s} fwd {pts.shape[0] * FWD / f_ms / 1e9:6.1f} TF ({f_ms:.2f} ms)  "
      f"fwd+bwd {gp.shape[0] * 2 * FWD / g_ms / 1e9:6.1f} TF ({g_ms:.2f} ms)  max|udf16-udf32| {float((u16 - u32).abs().max()):.2e}  "
      f"cos>=1-1e-5: {100 * float((cos >= 1 - 1e-5).float().mean()):.3f} %  sat {dec.saturation_count()}  udf16 sha {digest}", flush=True)
