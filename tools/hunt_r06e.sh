#!/bin/bash
# Round 6, fifth call: which of the three split products is wrong in a bad workgroup (host-side prediction vs observed error)
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/hunt5; mkdir -p $OUT
V=surfd_amd/lib/variants
timeout 600 tools/ubench/bin/lds_stage_test 100 3 > $OUT/lds_stage.txt 2>&1; echo "rc=$?" >> $OUT/lds_stage.txt
run() { local name=$1 lib=$2; shift 2; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    echo "== $name ($lib ${envs[*]}) :: $*" >> $OUT/summary.txt
    env SURFD_LIB=$PWD/$V/libsurfd_hip_$lib.so "${envs[@]}" timeout 500 "$@" > $OUT/$name.txt 2>&1; echo "rc=$?" >> $OUT/$name.txt
    grep -E '^\{|rc=' $OUT/$name.txt | cut -c1-300 >> $OUT/summary.txt; }
run es_a0_qkv a0 -- python tools/error_structure.py input_blocks.1.1.qkv 224 672 64 64 40 80 32 64
run es_a0_qkv2 a0 -- python tools/error_structure.py input_blocks.2.1.qkv 224 672 64 64 40 80 32 64
run es_g0_qkv g0 -- python tools/error_structure.py middle_block.1.qkv 896 2688 4 4 60 80 80 32
run es_g0_qkv8 g0 -- python tools/error_structure.py input_blocks.7.1.qkv 896 2688 8 8 60 80 80 32
cat $OUT/lds_stage.txt $OUT/summary.txt
