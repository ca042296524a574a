#!/bin/bash
# wide-form A/B of a variant library: bash tools/ab_wide.sh <name>   (tools/loop_ab.py three times each, alternating)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/abw_$1; mkdir -p $O; : > $O/ab.txt
for r in 1 2 3; do for lib in "" $PWD/surfd_amd/lib/variants/libsurfd_hip_$1.so; do
  env SURFD_LIB=$lib timeout 300 python tools/loop_ab.py 100 8 80 80 2>/dev/null | grep '^{' | cut -c1-330 >> $O/ab.txt; done; done
cat $O/ab.txt
