#!/bin/bash
# A/B of an environment setting on one library: bash tools/ab_env.sh <variant name or ""> VAR=a VAR=b ...   (latency form at 8 latents, ms per evaluation)
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1; shift
LIB=""; [ -n "$V" ] && LIB=$PWD/surfd_amd/lib/variants/libsurfd_hip_$V.so
O=gpurun_out/ab_env; mkdir -p $O; : > $O/ab.txt
for r in 1 2 3; do for kv in "$@"; do
  env SURFD_LIB=$LIB $kv timeout 300 python tools/loop_batch_sweep.py 32 8 200 0 2>&1 | grep "^| 32" | sed "s|^|$kv |" >> $O/ab.txt; done; done
cat $O/ab.txt
