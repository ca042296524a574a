#!/usr/bin/env python3
"""profiles/rNN_pmc_summary.json (tools/profile_round.sh) -> profiles/rNN_pmc_traffic.json, the per-launch HBM traffic
and pipe-utilisation figures bench.py puts into its `roofline.traffic` fields.

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; the counters sit on the L2's fabric side
(Infinity-Cache hits included).  Pipe fractions: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_WAVE_CYCLES) for kernels that run
one wave per SIMD, SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, SQ_WAIT_ANY / SQ_WAVE_CYCLES.

    python tools/pmc_to_traffic.py profiles/r02_pmc_summary.json profiles/r02_pmc_traffic.json
"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
r = json.load(open(src))


def pick(tag, sub):
    hits = [(k, v) for k, v in r.get(tag, {}).items() if sub in k]
    out = {"dispatches": 0}
    for _, v in hits:
        for a, b in v.items():
            out[a] = out.get(a, 0) + b
    return out


WEIGHT_BYTES = 553_294_340          # fp16 planes streamed per denoiser evaluation (bench.py: algorithmic_bytes_per_evaluation)
CONVS_PER_EVAL = 84
FWD = "decoder_fwd8_kernel" if any("decoder_fwd8_kernel" in k for k in r.get("fetch", {})) else "decoder_kernel<false, true>"   # the default forward kernel of the round
dec_f, dec_w, dec_s = (pick(t, FWD) for t in ("fetch", "write", "sq"))
n_dec = max(dec_f["dispatches"], 1)
unet_fetch = sum(pick("fetch", k).get("FETCH_SIZE", 0) for k in ("conv2_kernel", "attn_kernel")) * 1024 * 2
unet_write = sum(pick("write", k).get("WRITE_SIZE", 0) for k in ("conv2_kernel", "attn_kernel")) * 1024
evals = max(pick("fetch", "conv2_kernel")["dispatches"] / CONVS_PER_EVAL, 1)
c2 = pick("sq", "conv2_kernel")
out = {
    "source": (f"{src}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes) over `python bench.py --steps N "
               "--warmup 0 --diffusion-steps 20 ...` (the shortened command of tools/profile_round.sh: same kernels and grid shapes as the "
               "headline run, whose full length under --pmc would be hundreds of thousands of serialised dispatches); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at "
               "64 B); counters sit on the L2's fabric side and include Infinity-Cache hits"),
    "decoder_fwd_kernel": FWD,
    "decoder_fwd_launches_measured": n_dec,
    "decoder_fwd_fetch_bytes_per_launch": dec_f.get("FETCH_SIZE", 0) * 1024 * 2 / n_dec,
    "decoder_fwd_write_bytes_per_launch": dec_w.get("WRITE_SIZE", 0) * 1024 / n_dec,
    "unet_evals_measured": evals,
    "unet_eval_fetch_bytes": unet_fetch / evals,
    "unet_eval_write_bytes": unet_write / evals,
}
if "source_sha256" in r:          # the kernel sources the counters were collected with (tools/profile_round.sh): bench.py quotes
    out["source_sha256"] = r["source_sha256"]      # the profile only while they are unchanged
out["decoder_fwd_hbm_bytes_per_launch"] = out["decoder_fwd_fetch_bytes_per_launch"] + out["decoder_fwd_write_bytes_per_launch"]
out["unet_eval_hbm_bytes"] = out["unet_eval_fetch_bytes"] + out["unet_eval_write_bytes"]
out["unet_fetch_over_algorithmic"] = out["unet_eval_fetch_bytes"] / WEIGHT_BYTES
for name, d in (("decoder_fwd", dec_s), ("decoder_grad", pick("sq", "decoder_kernel<true, true>")), ("conv2", c2)):
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc:
        # SQ_WAVE_CYCLES sums over resident waves (quad-cycles): a kernel with two waves per SIMD counts every SIMD cycle twice
        per_simd = 2 if (name == "decoder_fwd" and FWD == "decoder_fwd8_kernel") else 1
        out[name + "_mfma_busy_frac"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc / per_simd)
        out[name + "_valu_frac"] = d.get("SQ_ACTIVE_INST_VALU", 0) / wc
        out[name + "_wait_frac"] = d.get("SQ_WAIT_ANY", 0) / wc
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
