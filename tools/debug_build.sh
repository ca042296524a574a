#!/bin/bash
# The debug builds of the C-ABI library in one place (VERDICT r5 "missing" #6): the variants that separated
# "uninitialised register / stale LDS" from "missing wait" from "scheduling" in round 6's hunt (profiles/r06_conv2_instability.md).
#   bash tools/debug_build.sh            builds surfd_amd/lib/variants/libsurfd_hip_dbg_{poison,nan,waits,mfmapad,O1}.so  (CPU box)
#   gpurun --timeout 1500 -- 'bash tools/debug_build.sh run'      runs the 40-evaluation determinism check and the per-sample
#                                                                L = 64 check on each of them (GPU box; output gpurun_out/dbg/)
# A debug library is selected with SURFD_LIB=<path>; surfd_build_config() reports it as unsafe_variants > 0 and the product
# tests refuse it.
cd "$(dirname "$0")/.." || exit 1
V=surfd_amd/lib/variants
if [ "${1:-build}" = "build" ]; then
  python tools/build_variants.py \
    dbg_poison=conv_f16x2.hip:-DSURFD_C2_DBG_POISON=1 \
    dbg_nan=conv_f16x2.hip:-DSURFD_C2_DBG_POISON=2 \
    dbg_waits=conv_f16x2.hip:-mllvm,-amdgpu-waitcnt-forcezero=1 \
    dbg_mfmapad=conv_f16x2.hip:-mllvm,-amdgpu-mfma-padding-ratio=100 \
    dbg_O1=conv_f16x2.hip:-O1
  exit $?
fi
O=gpurun_out/dbg; mkdir -p $O; : > $O/summary.txt
for v in dbg_poison dbg_nan dbg_waits dbg_mfmapad dbg_O1; do
  lib=$PWD/$V/libsurfd_hip_$v.so
  [ -f $lib ] || { echo "$v: not built" >> $O/summary.txt; continue; }
  for c in "8 0" "80 80"; do
    echo "== $v determinism $c: $(SURFD_LIB=$lib timeout 300 python tools/determinism_check.py 40 $c 2>&1 | grep -E 'distinct|differs' | tail -1)" >> $O/summary.txt
  done
  echo "== $v L=64: $(SURFD_LIB=$lib timeout 300 python tools/diag_l64.py 80 64 32 2>&1 | grep -E '^\{' | cut -c1-260)" >> $O/summary.txt
done
cat $O/summary.txt
