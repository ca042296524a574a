import os, sys
sys.path.insert(0, "/root/repo")
import torch
from surfd_amd import synth
from surfd_amd.cbndec import CbnDecoder
from surfd_amd.spec import DecoderConfig
dec = CbnDecoder(63, 32, 512, 5); dec.load_state_dict(synth.synth_decoder_state_dict(DecoderConfig(latent_dim=32)), strict=True); dec = dec.cuda().eval()
dec.bind_latents((torch.randn(1, 32) * 0.8).cuda())
pts = (torch.rand(1 << 22, 3) * 2 - 1).cuda()
for blocks in (256, 224, 192, 128, 64, 32):
    dec.set_grid_blocks(blocks if blocks < 256 else 0)
    dec.udf(pts, 0); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); dec.udf(pts, 0); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    tf = pts.shape[0] * 5308416 / ms / 1e9
    print(f"{os.environ.get('SURFD_DECODER_FWD8','1')} blocks {blocks:3d}: {tf:6.1f} TF  {tf / blocks:5.3f} TF per CU", flush=True)
