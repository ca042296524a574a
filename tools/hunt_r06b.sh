#!/bin/bash
# Round 6, second call: microbenchmarks for the two candidate mechanisms + the operand-pipelined K loop on the known-bad builds
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/hunt2; mkdir -p $OUT
V=surfd_amd/lib/variants
timeout 600 tools/ubench/bin/mfma_war_test 4096 10 > $OUT/mfma_war.txt 2>&1; echo "rc=$?" >> $OUT/mfma_war.txt
timeout 600 tools/ubench/bin/lds_stage_test 200 5 > $OUT/lds_stage.txt 2>&1; echo "rc=$?" >> $OUT/lds_stage.txt
run() { local name=$1 lib=$2; shift 2; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    echo "== $name ($lib ${envs[*]}) :: $*" >> $OUT/summary.txt
    env SURFD_LIB=$PWD/$V/libsurfd_hip_$lib.so "${envs[@]}" timeout 400 "$@" > $OUT/$name.txt 2>&1; echo "rc=$?" >> $OUT/$name.txt
    grep -E '^\{|distinct|differs|rc=' $OUT/$name.txt | cut -c1-700 >> $OUT/summary.txt; }
run l64_ab ab -- python tools/diag_l64.py 80 64 32
run l64_ab2 ab -- python tools/diag_l64.py 80 64 32
run det_gb gb -- python tools/determinism_check.py 30 80 80
run det_db db -- python tools/determinism_check.py 30
run det_d0 d0 -- python tools/determinism_check.py 30
run ab_d0 d0 -- python tools/loop_ab.py 100 8 80 80
run ab_db db -- python tools/loop_ab.py 100 8 80 80
run ab_d0_L64 d0 -- python tools/loop_ab.py 60 8 80 80 64
run ab_db_L64 db -- python tools/loop_ab.py 60 8 80 80 64
cat $OUT/mfma_war.txt $OUT/lds_stage.txt $OUT/summary.txt
