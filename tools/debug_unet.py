"""GPU debugging aid: native UNet vs oracle on reduced configurations."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet as ounet
from surfd_amd import synth
from surfd_amd.mdm import MDM
from surfd_amd.spec import UNetConfig

def run(cfg, B, L, tag):
    sd = synth.synth_unet_state_dict(cfg)
    m = MDM(cond_mode="no_cond", unet_cfg=cfg)
    m.load_state_dict(sd, strict=True)
    m.to("cuda")
    g = torch.Generator().manual_seed(B * 100 + L)
    x = torch.randn(B, 1, L, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    out = m(x.cuda(), t.cuda(), y={}).cpu()
    with torch.no_grad():
        ref = ounet.unet_forward(sd, x, t)
    err = (out - ref).abs().amax(dim=(1, 2))
    print(f"{tag:28s} B={B:2d} L={L:2d} max|ref|={float(ref.abs().max()):.3f} err/sample={[f'{e:.1e}' for e in err.tolist()]}", flush=True)

cfgs = {
    "1lvl_noattn": UNetConfig(channel_mult=(1,), num_res_blocks=1, attention_resolutions=()),
    "1lvl_attn": UNetConfig(channel_mult=(1,), num_res_blocks=1, attention_resolutions=(1,)),
    "2lvl_noattn": UNetConfig(channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=()),
    "2lvl_attn": UNetConfig(channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(1, 2)),
    "3lvl_noattn": UNetConfig(channel_mult=(1, 2, 4), num_res_blocks=1, attention_resolutions=()),
    "4lvl_noattn": UNetConfig(channel_mult=(1, 2, 4, 4), num_res_blocks=1, attention_resolutions=()),
    "4lvl_nrb2_noattn": UNetConfig(attention_resolutions=()),
    "full": UNetConfig(),
}
which = sys.argv[1].split(",") if len(sys.argv) > 1 else list(cfgs)
for name in which:
    for B, L in [(int(a), int(b)) for a, b in (p.split('x') for p in os.environ.get('BL', '2x32,3x32,8x32,8x64').split(','))]:
        run(cfgs[name], B, L, name)
