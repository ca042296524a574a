#!/bin/bash
# Where the reverse loop's fabric-side traffic comes from (4.6 x the 553 MB of weights per evaluation): FETCH_SIZE /
# WRITE_SIZE of one wide loop of 80 latents (20 evaluations) in four configurations — default; no split K (SURFD_CONV2_KSMAX=1: no
# partial tiles); no weight prefetch ahead (SURFD_CONV2_PFN=0); neither.  Separate --pmc passes (MI355X_MICROARCH.md); FETCH_SIZE
# doubled (gfx950 tallies 128-byte requests at 64).  Output: gpurun_out/traffic_split/summary.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/traffic_split; rm -rf $O; mkdir -p $O
CMD="python tools/loop_batch_sweep.py 32 80 20 80"
i=0
for cfg in "SURFD_X=0" "SURFD_CONV2_KSMAX=1" "SURFD_CONV2_PFN=0" "SURFD_CONV2_KSMAX=1 SURFD_CONV2_PFN=0"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    env $cfg timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/c${i}_$ctr -o p -- $CMD > $O/c${i}_$ctr.log 2>&1
  done
  echo "$cfg" > $O/c${i}.cfg
  i=$((i+1))
done
python - <<'PY'
import csv, glob, json, os
O = "gpurun_out/traffic_split"
out = []
for i in range(4):
    cfg = open(f"{O}/c{i}.cfg").read().strip()
    row = {"config": cfg}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"{O}/c{i}_{ctr}/**/*counter_collection.csv", recursive=True)
        tot, disp = 0.0, set()
        per = {}
        for r in csv.DictReader(open(files[0])):
            k = r["Kernel_Name"]
            if "conv2_kernel" in k or "attn_kernel" in k:
                tot += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        evals = len(disp) / 100.0            # 84 convolutions + 16 attention cores per evaluation
        row[ctr + "_bytes_per_evaluation"] = tot * 1024 * (2 if ctr == "FETCH_SIZE" else 1) / max(evals, 1)
        row["evaluations"] = evals
    ms = [l for l in open(f"{O}/c{i}_FETCH_SIZE.log") if l.startswith("| 32")]
    out.append(row)
json.dump(out, open(f"{O}/summary.json", "w"), indent=1)
for r in out: print(r)
for d in glob.glob(f"{O}/c*_*SIZE"):
    import shutil; shutil.rmtree(d)
PY
