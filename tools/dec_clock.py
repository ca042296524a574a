#!/usr/bin/env python3
"""Shader clock the chip sustains under the forward decoder kernel (4-wave stamps build, SURFD_LIB=...stamps.so,
SURFD_DECODER_FWD8=0) as a function of how many CUs run it: cycles per tile (s_memtime) / wall time per tile."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import _native as N, synth
from surfd_amd.cbndec import CbnDecoder
from surfd_amd.spec import DecoderConfig
dec = CbnDecoder(63, 32, 512, 5)
dec.load_state_dict(synth.synth_decoder_state_dict(DecoderConfig(latent_dim=32)), strict=True)
dec = dec.cuda().eval()
dec.bind_latents((torch.randn(1, 32) * 0.8).cuda())
L = N.lib()
out = (C.c_longlong * 8)()
for blocks in (256, 192, 128, 64):
    pts = (torch.rand(blocks * 64 * 64, 3) * 2 - 1).cuda()          # 64 tiles per workgroup
    dec.set_grid_blocks(blocks if blocks < 256 else 0)
    dec.udf(pts, 0); torch.cuda.synchronize()
    L.surfd_decoder_debug_stamps(out, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); dec.udf(pts, 0); b.record(); torch.cuda.synchronize()
    L.surfd_decoder_debug_stamps(out, 1)
    cyc = sum(out[i] for i in range(5)) / 64
    us = a.elapsed_time(b) * 1e3 / 64
    print(f"{blocks:3d} CUs: {cyc:9.0f} cycles per tile, {us:6.1f} us per tile -> {cyc / us / 1e3:.2f} GHz; {pts.shape[0] * 5308416 / a.elapsed_time(b) / 1e9 / blocks:.3f} TFLOP/s per CU", flush=True)
