#!/usr/bin/env python3
"""Where do the bits of an unstable conv launch change?  Needs a -DSURFD_C2_PROBE build of conv_f16x2.hip (SURFD_LIB=...): every
workgroup of every f16x2 conv launch adds checksums of its phases to eight 64-bit words (operand as loaded, GroupNorm mean /
scale, staged values, slab read back from LDS, weight fragments consumed, accumulators, stored values, epilogue operands).  N
evaluations of the same input are compared word by word with the first: the first (launch, phase) that differs names the
phase in which timing enters, the workgroup numbers say where on the chip.
python tools/probe_phases.py [N] [B] [L] [wide design batch]"""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 80
LEN = int(sys.argv[3]) if len(sys.argv) > 3 else 64
WIDE = int(sys.argv[4]) if len(sys.argv) > 4 else 32
NOPS, NWG, NSLOT = 128, 4096, 8
probe = torch.zeros(NOPS * NWG * NSLOT, dtype=torch.int64, device="cuda")
os.environ["SURFD_CONV2_PROBE_PTR"] = hex(probe.data_ptr())       # read by the library at its first conv launch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
model.set_wide(WIDE)
g = torch.Generator().manual_seed(3)
x = torch.randn(B, 1, LEN, generator=g).cuda(); t = torch.randint(0, 1000, (B,), generator=g).cuda()
model(x, t, y={}); torch.cuda.synchronize()
SLOTS = ["operand", "gn_mean_scale", "staged", "slab_lds", "weights", "acc", "stored", "epi_operands"]
ref = ref_out = None
report = {"lib": os.path.basename(os.environ.get("SURFD_LIB", "default")), "B": B, "L": LEN, "design": WIDE, "runs": N, "first_diffs": [], "out_distinct": 1}
outs = set()
for run in range(N):
    probe.zero_(); torch.cuda.synchronize()
    out = model(x, t, y={}); torch.cuda.synchronize()
    p = probe.view(NOPS, NWG, NSLOT).cpu()
    outs.add(out.cpu().numpy().tobytes())
    if ref is None:
        ref, ref_out = p.clone(), out.clone()
        used = [int(i) for i in torch.nonzero(p.flatten(1).abs().sum(1)).flatten()]
        report["ops_probed"] = len(used)
        continue
    d = (p != ref)
    if not bool(d.any()):
        continue
    ops = [int(i) for i in torch.nonzero(d.flatten(1).any(1)).flatten()]
    op = ops[0]
    slots = [SLOTS[int(k)] for k in torch.nonzero(d[op].any(0)).flatten()]
    per_slot = {}
    for k in torch.nonzero(d[op].any(0)).flatten():
        wgs = [int(w) for w in torch.nonzero(d[op][:, int(k)]).flatten()]
        per_slot[SLOTS[int(k)]] = {"n_wg": len(wgs), "wgs": wgs[:24], "min_wg": min(wgs), "max_wg": max(wgs)}
    nwg_used = int((ref[op].abs().sum(1) != 0).sum())
    report["first_diffs"].append({"run": run, "first_op": op, "later_ops_differing": len(ops) - 1, "workgroups_of_op": nwg_used, "phases": per_slot,
                                  "out_max_abs_diff": float((out - ref_out).abs().max())})
report["out_distinct"] = len(outs)
print(json.dumps(report), flush=True)
