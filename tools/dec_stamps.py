"""Phase timing of the forward decoder kernel (debug build: SURFD_EXTRA_HIPCC_FLAGS=-DSURFD_DEC_STAMPS)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import _native as N, synth
from surfd_amd.cbndec import CbnDecoder
from surfd_amd.spec import DecoderConfig
dec = CbnDecoder(63, 32, 512, 5)
dec.load_state_dict(synth.synth_decoder_state_dict(DecoderConfig(latent_dim=32)), strict=True)
dec = dec.cuda().eval()
dec.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f16x2")
lat = (torch.randn(1, 32) * 0.8).cuda()
dec.bind_latents(lat)
pts = (torch.rand(256 * 64 * 64, 3) * 2 - 1).cuda()
L = N.lib()
out = (C.c_longlong * 8)()
dec.udf(pts, 0); torch.cuda.synchronize()
L.surfd_decoder_debug_stamps(out, 1)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); dec.udf(pts, 0); t1.record(); torch.cuda.synchronize()
L.surfd_decoder_debug_stamps(out, 1)
names = ["fetch+encode", "gemm", "epilogue", "barrier", "output"]
tot = sum(out[i] for i in range(5))
print("kernel ms", t0.elapsed_time(t1), "tiles per WG 64; TF-eq", pts.shape[0] * 5308416 / t0.elapsed_time(t1) / 1e9)
for i, n in enumerate(names):
    print(f"  {n:14s} {out[i]:12d} cycles  {100.0 * out[i] / tot:5.1f} %   per tile {out[i] / 64:9.0f}")
print("  total", tot, "per tile", tot / 64)
