#!/usr/bin/env python3
"""What a two-pass gradient path could gain at best ("build it or close it with a measurement").

Measured on one box, kernels alone on all CUs: the 8-wave forward kernel, the 4-wave forward kernel (the forward half of today's
gradient kernel is that arithmetic plus the gate words), the 4-wave forward + reverse-sweep kernel.  The two-pass design =
pass 1: 8-wave forward + 704 B of gate words per point; pass 2: a reverse-sweep-only kernel.  Its time per gradient query is
bounded BELOW by t(8-wave forward) + gate traffic at the measured HBM rate + t(reverse sweep) where the reverse sweep cannot
beat the forward kernel's rate on the same GEMM shapes (same 10 x 512 x 512 products on the transposed planes, same split
arithmetic, plus the gate application).  The script prints the measured rates, the reverse sweep's share of today's kernel
(t_grad - t_fwd4), and the bound.      python tools/grad_two_pass_bound.py [log2 points]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lg = sys.argv[1] if len(sys.argv) > 1 else "22"


def run(env):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "dec_time.py"), lg], env={**os.environ, **env}, text=True)
    line = [l for l in out.splitlines() if " fwd " in l][-1]
    tf = float(line.split(" fwd ")[1].split(" TF")[0]); tg = float(line.split("fwd+bwd ")[1].split(" TF")[0])
    return tf, tg, line.strip()


f8, g, l8 = run({})
f4, g2, l4 = run({"SURFD_DECODER_FWD8": "0"})
FWD = 5308416.0
t_f8, t_f4, t_g = FWD / f8, FWD / f4, 2 * FWD / ((g + g2) / 2)           # ps of chip time per query (TFLOP/s -> flop / (TF) = ps)
t_rev_now = t_g - t_f4                                                     # what the reverse sweep costs inside today's kernel
gate_ps = 704.0 * 2 / 3.0e12 * 1e12                                        # 704 B written + read back per point at ~3 TB/s effective: ps per point
best = t_f8 + gate_ps + t_f8                                               # reverse sweep at the forward kernel's rate: the optimistic end
same = t_f8 + gate_ps + t_rev_now                                          # reverse sweep no faster than it is today
res = {"fwd8_tflops": f8, "fwd4_tflops": f4, "fwd_bwd_tflops_2x_accounting": (g + g2) / 2,
       "ps_per_query": {"fwd8": t_f8, "fwd4": t_f4, "fwd_bwd_now": t_g, "reverse_sweep_now": t_rev_now, "gate_words_704B": gate_ps},
       "two_pass_ps_per_gradient_query": {"reverse_at_forward_rate (upper end)": best, "reverse_as_today": same},
       "two_pass_tflops_2x_accounting": {"upper end": 2 * FWD / best, "reverse_as_today": 2 * FWD / same},
       "gain_over_today": {"upper end": t_g / best - 1.0, "reverse_as_today": t_g / same - 1.0},
       "lines": [l8, l4]}
print(json.dumps(res, indent=1))
