#!/bin/bash
# Round 6, third call: does the register pin (no load into a just-issued MFMA's source registers) matter, or only the timing?
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/hunt3; mkdir -p $OUT
V=surfd_amd/lib/variants
timeout 900 tools/ubench/bin/mfma_war_test 4096 10 > $OUT/mfma_war.txt 2>&1; echo "rc=$?" >> $OUT/mfma_war.txt
run() { local name=$1 lib=$2; shift 2; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    echo "== $name ($lib ${envs[*]}) :: $*" >> $OUT/summary.txt
    env SURFD_LIB=$PWD/$V/libsurfd_hip_$lib.so "${envs[@]}" timeout 400 "$@" > $OUT/$name.txt 2>&1; echo "rc=$?" >> $OUT/$name.txt
    grep -E '^\{|distinct|differs|rc=' $OUT/$name.txt | cut -c1-700 >> $OUT/summary.txt; }
run l64_ab2 ab2 -- python tools/diag_l64.py 80 64 32
run l64_ab2b ab2 -- python tools/diag_l64.py 80 64 32
run det_gb2 gb2 -- python tools/determinism_check.py 30 80 80
run l64_ab_80_160 ab -- python tools/diag_l64.py 80 64 160
run l64_ab_40_80 ab -- python tools/diag_l64.py 40 64 80
run l64_ab_24_32 ab -- python tools/diag_l64.py 24 64 32
run det_gb_b gb -- python tools/determinism_check.py 40 80 80
run det_gb_c gb -- python tools/determinism_check.py 40 160 160
run det_gb_d gb -- python tools/determinism_check.py 40 53 80
run l64_a0 a0 -- python tools/diag_l64.py 80 64 32
run det_g0 g0 -- python tools/determinism_check.py 30 80 80
cat $OUT/mfma_war.txt $OUT/summary.txt
