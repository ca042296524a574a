#!/bin/bash
# PMC passes over the forward decoder kernel (run on the GPU box through gpurun).
# usage: pmc_decoder.sh <f16x2|fp32> [resolution]
mode=${1:-f16x2}; res=${2:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcd_$i -o p -- \
      python bench.py --decoder-precision $mode --resolution $res --steps 1 --warmup 0 --diffusion-steps 10 --no-cpu-baseline > /tmp/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pmc_$i.log; continue; }
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(float); n = 0
f = glob.glob("/tmp/pmcd_$i/**/*counter_collection.csv", recursive=True)
disp = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "decoder_kernel" in k and ("ILb0E" in k or "<false" in k):
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
# durations of the same dispatches from the kernel trace
dur = 0.0
kt = glob.glob("/tmp/pmcd_$i/**/*kernel_trace.csv", recursive=True)
for r in csv.DictReader(open(kt[0])):
    k = r["Kernel_Name"]
    if "decoder_kernel" in k and ("ILb0E" in k or "<false" in k):
        dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
w = agg.get("SQ_WAVE_CYCLES", 0)
print("$mode res $res pass $i: launches", len(disp), "kernel time %.3f s" % dur)
print("   raw", dict(agg))
if w: print("   per wave quad-cycle", {k: round(v / w, 4) for k, v in agg.items()})
if "GRBM_GUI_ACTIVE" in agg and dur: print("   effective clock %.3f GHz (GRBM_GUI_ACTIVE / 8 XCD-instances? raw/dur = %.3f G)" % (agg["GRBM_GUI_ACTIVE"] / dur / 1e9, agg["GRBM_GUI_ACTIVE"] / dur / 1e9))
PY
done
