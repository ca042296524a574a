#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_$1; mkdir -p $O; : > $O/ab.txt
for r in 1 2 3; do for lib in "" $PWD/surfd_amd/lib/variants/libsurfd_hip_$1.so; do
  env SURFD_LIB=$lib timeout 300 python tools/loop_batch_sweep.py 32 8 200 0 2>&1 | grep "^| 32" | sed "s|^|lib=$(basename "$lib") |" >> $O/ab.txt; done; done
env SURFD_LIB=$PWD/surfd_amd/lib/variants/libsurfd_hip_$1.so timeout 300 python tools/loop_ab.py 100 8 80 80 2>/dev/null | grep '^{' >> $O/ab.txt
cat $O/ab.txt
