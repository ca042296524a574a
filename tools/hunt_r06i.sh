#!/bin/bash
# split-K partial tiles requested C2_ZB slices at a time (sum order unchanged): A/B on one box
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/zb; mkdir -p $OUT
V=surfd_amd/lib/variants
for v in zb1 zb4 zb2 zb3 zb8 zb1 zb4 zb3; do
  env SURFD_LIB=$PWD/$V/libsurfd_hip_$v.so timeout 400 python tools/loop_ab.py 100 8 80 80 2>/dev/null | grep '^{' >> $OUT/loop_ab_L32.txt
done
for v in zb1 zb4 zb3; do
  env SURFD_LIB=$PWD/$V/libsurfd_hip_$v.so timeout 400 python tools/loop_ab.py 50 8 80 80 64 2>/dev/null | grep '^{' >> $OUT/loop_ab_L64.txt
done
cat $OUT/loop_ab_L32.txt $OUT/loop_ab_L64.txt
