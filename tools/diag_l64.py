#!/usr/bin/env python3
"""Diagnostic: one denoiser evaluation at B x L in the wide form against the exact-fp32 mode and the latency form, per sample.
python tools/diag_l64.py [B] [L] [design]"""
import os, sys, types, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 80
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
D = int(sys.argv[3]) if len(sys.argv) > 3 else 32
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
g = torch.Generator().manual_seed(3)
x = torch.randn(B, 1, L, generator=g).cuda(); t = torch.randint(0, 1000, (B,), generator=g).cuda()
model.set_precision("fp32"); ref = model(x, t, y={}).clone(); model.set_precision("f16x2")
lat = model(x, t, y={}).clone()
model.set_wide(D); wide = model(x, t, y={}).clone(); wide2 = model(x, t, y={}).clone()
e_lat = (lat - ref).abs().amax(dim=(1, 2)); e_w = (wide - ref).abs().amax(dim=(1, 2))
bad = [int(i) for i in torch.nonzero(e_w > 1e-4).flatten()]
print(json.dumps({"lib": os.path.basename(os.environ.get("SURFD_LIB", "default")), "pfn_min": os.environ.get("SURFD_CONV2_PFN_MIN"), "B": B, "L": L, "design": D,
                  "latency_max_err": float(e_lat.max()), "wide_max_err": float(e_w.max()), "wide_repeatable": bool(torch.equal(wide, wide2)),
                  "bad_samples": bad[:40], "n_bad": len(bad), "bad_err": [round(float(e_w[i]), 5) for i in bad[:12]],
                  "bad_positions_of_first": ([int(p) for p in torch.nonzero((wide[bad[0]] - ref[bad[0]]).abs().flatten() > 1e-4).flatten()][:70] if bad else [])}), flush=True)
