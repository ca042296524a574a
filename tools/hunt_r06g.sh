#!/bin/bash
# Round 6, seventh call: the cross-lane reduction next to matrix-pipe traffic (ubench), and wait states in the GroupNorm reductions
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/hunt7; mkdir -p $OUT
V=surfd_amd/lib/variants
timeout 900 tools/ubench/bin/xlane_raw_test 20000 2 > $OUT/xlane_raw.txt 2>&1; echo "rc=$?" >> $OUT/xlane_raw.txt
run() { local name=$1 lib=$2; shift 2; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    echo "== $name ($lib ${envs[*]}) :: $*" >> $OUT/summary.txt
    env SURFD_LIB=$PWD/$V/libsurfd_hip_$lib.so "${envs[@]}" timeout 500 "$@" > $OUT/$name.txt 2>&1; echo "rc=$?" >> $OUT/$name.txt
    grep -E '^\{|distinct|differs|rc=' $OUT/$name.txt | cut -c1-330 >> $OUT/summary.txt; }
run l64_ap ap -- python tools/diag_l64.py 80 64 32
run l64_ap2 ap -- python tools/diag_l64.py 80 64 32
run es_ap_qkv ap -- python tools/error_structure.py input_blocks.1.1.qkv 224 672 64 64 60 80 32 64
run det_gp gp -- python tools/determinism_check.py 40 80 80
run det_gp160 gp -- python tools/determinism_check.py 40 160 160
run es_gp_qkv gp -- python tools/error_structure.py middle_block.1.qkv 896 2688 4 4 60 80 80 32
cat $OUT/xlane_raw.txt $OUT/summary.txt
