#!/bin/bash
# PMC passes over the forward decoder kernel (run on the GPU box through gpurun).  usage: pmc_decoder.sh <mode>
mode=${1:-f16x2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  SURFD_DECODER_PRECISION=$mode timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcd_$i -o p -- \
      python bench.py --resolution 128 --steps 1 --warmup 0 --diffusion-steps 10 --no-cpu-baseline > /tmp/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pmc_$i.log; continue; }
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(float)
f = glob.glob("/tmp/pmcd_$i/**/*counter_collection.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "decoder_kernel" in r["Kernel_Name"] and "Lb0E" in r["Kernel_Name"].replace("false", "Lb0E") :
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
w = agg.get("SQ_WAVE_CYCLES", 0)
print("$mode pass $i:", {k: (round(v / w, 4) if w else v) for k, v in agg.items()}, "wave_cycles", w)
PY
done
