#!/usr/bin/env python3
"""Timeline of one pipelined bench run from a rocprofv3 kernel trace (run on the GPU box):
    rocprofv3 --kernel-trace --output-format csv -d DIR -o kt -- python bench.py ...
    python tools/pipeline_timeline.py DIR OUT.json
Per queue: busy time and span; for the queue that runs the decoder kernels: idle gaps; for the loop queues: when
each batch's reverse loop started and ended (1000 loop_advance kernels per batch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
q = defaultdict(list)
with open(f) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        kind = ("dec" if "decoder_kernel" in n else "adv" if "loop_advance" in n else "conv" if "conv2_kernel" in n else
                "attn" if "attn_kernel" in n else "grid" if ("classify" in n or "fill_kernel" in n or "commit" in n or "emit" in n) else "other")
        q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
t0 = min(v[0][0] for v in q.values() if v)
res = {"queues": {}}
for k, v in q.items():
    v.sort()
    kinds = defaultdict(int)
    busy = 0
    for s, e, kd in v:
        kinds[kd] += 1
        busy += e - s
    info = {"kernels": len(v), "kinds": dict(kinds), "first_ms": (v[0][0] - t0) / 1e6, "last_ms": (v[-1][1] - t0) / 1e6, "busy_ms": busy / 1e6}
    if kinds["dec"] > 10:
        dec = [(s, e) for s, e, kd in v if kd in ("dec", "grid")]
        gaps = []
        for (s0, e0), (s1, e1) in zip(dec, dec[1:]):
            if s1 - e0 > 2e6:
                gaps.append({"at_ms": round((e0 - t0) / 1e6, 1), "gap_ms": round((s1 - e0) / 1e6, 1)})
        info["decoder_gaps_over_2ms"] = gaps
        info["decoder_gap_total_ms"] = round(sum(g["gap_ms"] for g in gaps), 1)
        info["decoder_busy_ms"] = round(sum(e - s for s, e in dec) / 1e6, 1)
    if kinds["adv"] >= steps:
        adv = [(s, e) for s, e, kd in v if kd == "adv"]
        info["loops"] = [{"start_ms": round((adv[i][0] - t0) / 1e6, 1), "end_ms": round((adv[min(i + steps, len(adv)) - 1][1] - t0) / 1e6, 1)}
                         for i in range(0, len(adv), steps)]
    res["queues"][k] = info
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
