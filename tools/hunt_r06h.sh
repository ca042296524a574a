#!/bin/bash
# Round 6, eighth call: in-wave GroupNorm with padded reductions as the statistics path of every instantiation
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/hunt8; mkdir -p $OUT
V=surfd_amd/lib/variants
run() { local name=$1 lib=$2; shift 2; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    echo "== $name ($lib ${envs[*]}) :: $*" >> $OUT/summary.txt
    env SURFD_LIB=$PWD/$V/libsurfd_hip_$lib.so "${envs[@]}" timeout 500 "$@" > $OUT/$name.txt 2>&1; echo "rc=$?" >> $OUT/$name.txt
    grep -E '^\{|distinct|differs|rc=' $OUT/$name.txt | cut -c1-330 >> $OUT/summary.txt; }
run l64_agp agp -- python tools/diag_l64.py 80 64 32
run l64_agp2 agp -- python tools/diag_l64.py 80 64 160
run l64_agp3 agp -- python tools/diag_l64.py 40 64 80
run l64_gp gp -- python tools/diag_l64.py 80 64 32
for v in gp1 gp2 gp8; do run det_$v $v -- python tools/determinism_check.py 30 80 80; run det160_$v $v -- python tools/determinism_check.py 30 160 160; done
run det_gp_lat gp -- python tools/determinism_check.py 40 8 0
run det_agp_l64 agp -- python tools/error_structure.py input_blocks.1.1.qkv 224 672 64 64 60 80 32 64
run ab_d0 d0 -- python tools/loop_ab.py 100 8 80 80
run ab_gp gp -- python tools/loop_ab.py 100 8 80 80
run ab_gp8 gp8 -- python tools/loop_ab.py 100 8 80 80
run ab_agp agp -- python tools/loop_ab.py 100 8 80 80
run ab_d0_L64 d0 -- python tools/loop_ab.py 50 8 80 80 64
run ab_agp_L64 agp -- python tools/loop_ab.py 50 8 80 80 64
cat $OUT/summary.txt
