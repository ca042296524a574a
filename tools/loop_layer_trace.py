#!/usr/bin/env python3
"""Per-launch view of the reverse loop from a `rocprofv3 --kernel-trace --output-format csv` trace of
tools/loop_batch_sweep.py / loop_chain_sweep.py.

    python tools/loop_layer_trace.py <kernel_trace.csv> [launches per evaluation = 102]

Per queue: the launches of the loop kernels (conv2 / conv / attn / loop_step) in start order, folded onto their
position inside one evaluation: mean duration, workgroups, mean gap to the previous launch's end on the same queue.
Then, over all queues: how many loop kernels were running at once (time-weighted), to tell contention (long kernels)
from starvation (gaps) when several loops share the chip."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
per_eval = int(sys.argv[2]) if len(sys.argv) > 2 else 102
LOOP = ("conv2_kernel", "attn_kernel", "loop_step_kernel", "loop_advance_kernel")      # 84 + 16 + 1 + 1 per evaluation
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if any(k in n for k in LOOP):
            wg = int(r["Workgroup_Size_X"])
            rows.append((int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void surfd::", "").replace("surfd::", ""),
                         int(r["Grid_Size_X"]) // max(wg, 1)))
byq = defaultdict(list)
for q, s, e, n, g in rows:
    byq[q].append((s, e, n, g))
print(f"{len(rows)} loop launches on {len(byq)} queue(s)")
for q, lst in sorted(byq.items()):
    lst.sort()
    n_eval = len(lst) // per_eval
    if n_eval < 3:
        print(f"queue {q}: {len(lst)} launches (fewer than 3 evaluations of {per_eval}) — skipped")
        continue
    # the loop's launches are periodic with period per_eval; drop everything before the last n_eval*per_eval launches
    tail = lst[len(lst) - n_eval * per_eval:]
    dur = [0.0] * per_eval
    gap = [0.0] * per_eval
    name = [""] * per_eval
    wgs = [0] * per_eval
    cnt = 0
    for ev in range(1, n_eval):          # skip the first folded evaluation (gap to an unrelated kernel)
        for i in range(per_eval):
            s, e, n, g = tail[ev * per_eval + i]
            ps, pe, _, _ = tail[ev * per_eval + i - 1]
            dur[i] += (e - s) / 1e3
            gap[i] += (s - pe) / 1e3
            name[i], wgs[i] = n, g
        cnt += 1
    print(f"\nqueue {q}: {n_eval} evaluations; mean per evaluation: kernels {sum(dur) / cnt:.1f} us + gaps {sum(gap) / cnt:.1f} us")
    print("pos kernel                         WGs   dur_us  gap_us")
    for i in range(per_eval):
        print(f"{i:3d} {name[i][:30]:30s} {wgs[i]:5d} {dur[i] / cnt:8.2f} {gap[i] / cnt:7.2f}")
# concurrency histogram over all queues
ev = []
for q, s, e, n, g in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = defaultdict(int)
cur, last = 0, ev[0][0] if ev else 0
for t, d in ev:
    hist[cur] += t - last
    cur += d; last = t
tot = sum(hist.values()) or 1
print("\nloop kernels running at once (share of the traced span): " + ", ".join(f"{k}: {v / tot:.1%}" for k, v in sorted(hist.items())))
