#!/usr/bin/env python3
"""Pipeline configuration sweep on the GPU box: bench.py at a reduced step count over loop chains x CU budget of the
loops, then the decoder's CU share around the best point.  Writes one line per run and the winner to the output
directory.  (Development tool; the chosen values are baked into bench.py's defaults by hand.)"""
import json
import os
import subprocess
import sys

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
os.makedirs(out, exist_ok=True)
log = open(os.path.join(out, "sweep.jsonl"), "a")


def run(chains, cus, blocks):
    cmd = [sys.executable, "bench.py", "--steps", str(steps), "--warmup", "2", "--no-trace", "--no-e2", "--no-cpu-baseline",
           "--loop-chains", str(chains), "--loop-cus", str(cus), "--decoder-blocks", str(blocks)]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        rec = {"chains": chains, "cus": cus, "blocks": blocks, "value": r["value"], "ms_per_step": r["ms_per_step"],
               "loop_alone_ms": r.get("roofline_loop", {}).get("loop_alone_ms_per_eval")}
    except Exception as e:                                   # noqa: BLE001
        rec = {"chains": chains, "cus": cus, "blocks": blocks, "value": 0.0, "error": repr(e)[:300]}
    log.write(json.dumps(rec) + "\n"); log.flush()
    print(rec, flush=True)
    return rec


res = [run(3, 256, 192)]
for chains in (3, 4, 6):
    for cus in (32, 64, 128):
        res.append(run(chains, cus, 192))
best = max(res, key=lambda r: r["value"])
for blocks in (184, 200):
    res.append(run(best["chains"], best["cus"], blocks))
best = max(res, key=lambda r: r["value"])
json.dump(best, open(os.path.join(out, "best.json"), "w"))
print("BEST", best)
