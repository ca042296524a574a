#!/usr/bin/env python3
"""Oracle <-> reference CPU timing ratio: runs ONLY in the build container, where the reference is
mounted at /root/reference.  It imports the reference's own modules (tools/make_golden.py's import recipe — nothing is
copied), gives them and the oracle the same synthetic weights and the same inputs, and times both on the same cores:

  * one denoiser evaluation (UNetModel.forward through MDM) at batch 1 and 8, L = 32
  * udf_func forward (meshudf.sample_udf) at 2^12 and 2^16 points, chunk = the point count
  * meshudf.sample_grads (autograd through the decoder) at 2^12 and 2^16 points, chunk 2^12
  * GridFiller(64) on the analytic thin-shell field: pure bookkeeping (the field is a few elementwise ops)

The ratio (oracle time / reference time, target 1.0 +- 0.1) is what ties bench.py's `cpu_baseline` — the oracle timed on
the GPU box's host, where the reference does not exist — back to the reference itself.

    python tools/cpu_ratio.py [--threads 8] [--out profiles/r04_cpu_oracle_vs_reference.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))


def best_of(fn, reps):
    fn()                                    # warm-up (allocator, thread pool)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r04_cpu_oracle_vs_reference.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    import make_golden as mg
    from oracle import decoder as odec
    from oracle import gridfiller as ogrid
    from oracle import unet as ounet
    from surfd_amd import synth
    from surfd_amd.spec import DecoderConfig
    R = mg.import_reference(a.ref)
    rows = []

    def row(name, ref_fn, ora_fn, reps, check=None):
        if check is not None:
            check()
        tr, to = best_of(ref_fn, reps), best_of(ora_fn, reps)
        rows.append({"case": name, "reference_s": tr, "oracle_s": to, "oracle_over_reference": to / tr})
        print(f"{name:46s} reference {tr * 1e3:9.2f} ms   oracle {to * 1e3:9.2f} ms   ratio {to / tr:.3f}", flush=True)

    # ---- denoiser ---------------------------------------------------------------------------------------------
    model, _ = mg.build_model(R, "no_cond")
    sd = synth.synth_unet_state_dict()
    for B in (1, 8):
        x = mg.rnd((B, 1, 32), 7 + B)
        t = torch.full((B,), 500)

        def ref_step():
            with torch.no_grad():
                return model(x, t, y={})

        def ora_step():
            with torch.no_grad():
                return ounet.unet_forward(sd, x, t)
        assert float((ref_step() - ora_step()).abs().max()) < 1e-4
        row(f"denoiser evaluation, B={B}, L=32", ref_step, ora_step, 5)
    # ---- decoder ----------------------------------------------------------------------------------------------
    dec = mg.build_decoder(R, 32)
    for prm in dec.parameters():          # as every reference driver does (sample/generate_uncond.py:80-82): no weight gradients in sample_grads
        prm.requires_grad = False
    dsd = synth.synth_decoder_state_dict(DecoderConfig(latent_dim=32))
    lat = mg.rnd((1, 32), 51, 0.8)
    f_ref = mg.make_ref_udf(R, dec, lat)
    f_ora = odec.make_udf_func(dsd, lat)
    for n in (2 ** 12, 2 ** 16):
        pts = torch.rand(n, 3, generator=torch.Generator().manual_seed(n)) * 2 - 1
        assert float((R.meshudf.sample_udf(f_ref, pts[:256], 256) - odec.sample_udf(f_ora, pts[:256], 256)).abs().max()) < 1e-6
        row(f"udf_func forward, {n} points", lambda: R.meshudf.sample_udf(f_ref, pts, n), lambda: odec.sample_udf(f_ora, pts, n), 3)
        row(f"sample_grads, {n} points, chunk 4096", lambda: R.meshudf.sample_grads(f_ref, pts, 2 ** 12),
            lambda: odec.sample_grads(f_ora, pts, 2 ** 12), 2)
    # ---- grid bookkeeping -------------------------------------------------------------------------------------
    def ref_grid():
        return R.meshudf.GridFiller(64).fill_grid(ogrid.analytic_field, 2 ** 30)

    def ora_grid():
        return ogrid.fill_grid(ogrid.analytic_field, 64, 2 ** 30)
    row("GridFiller(64), analytic field (bookkeeping)", ref_grid, ora_grid, 3)

    # ---- the mix bench.py's cpu_baseline extrapolates with (per shape of the headline workload) ---------------------
    by = {r["case"]: r for r in rows}
    T, B = 1000, 8
    n_fwd, n_grad = 10_398_866.0, 51_168.0           # decoder queries per shape, W-real (BENCH_r03.json)
    def per_shape(key):
        step = by["denoiser evaluation, B=8, L=32"][key]
        fwd = by["udf_func forward, 65536 points"][key] / 65536
        grd = by["sample_grads, 65536 points, chunk 4096"][key] / 65536
        return T * step / B + n_fwd * fwd + n_grad * grd
    mix = per_shape("oracle_s") / per_shape("reference_s")
    out = {"what": "oracle (oracle/*.py) vs the imported reference on the same cores of the build container; min over repeats after a warm-up",
           "threads": a.threads, "host_cores": os.cpu_count(), "torch": torch.__version__,
           "rows": rows,
           "headline_mix": {"oracle_over_reference": mix, "reference_shapes_per_s": 1.0 / per_shape("reference_s"),
                            "oracle_shapes_per_s": 1.0 / per_shape("oracle_s"),
                            "mix": f"{T} denoiser evaluations at batch {B} / {B} + {n_fwd:.0f} forward + {n_grad:.0f} gradient decoder queries per shape "
                                   "(grid bookkeeping: see its own row; < 1 % of a shape on either side)"}}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(f"headline mix: oracle / reference = {mix:.3f}   -> {a.out}")


if __name__ == "__main__":
    main()
