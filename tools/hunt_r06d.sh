#!/bin/bash
# Round 6, fourth call: the shape of the errors (one conv on a fixed input against the exact-fp32 kernel) + VALU write-after-read ubench
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/hunt4; mkdir -p $OUT
V=surfd_amd/lib/variants
timeout 900 tools/ubench/bin/mfma_valu_war_test 2048 3 > $OUT/mfma_valu_war.txt 2>&1; echo "rc=$?" >> $OUT/mfma_valu_war.txt
run() { local name=$1 lib=$2; shift 2; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    echo "== $name ($lib ${envs[*]}) :: $*" >> $OUT/summary.txt
    env SURFD_LIB=$PWD/$V/libsurfd_hip_$lib.so "${envs[@]}" timeout 500 "$@" > $OUT/$name.txt 2>&1; echo "rc=$?" >> $OUT/$name.txt
    grep -E '^\{|distinct|differs|rc=|first differing|^   \(' $OUT/$name.txt | cut -c1-1500 >> $OUT/summary.txt; }
run layers_a0 a0 -- python tools/determinism_layers.py 16 80 32 64
run es_a0_qkv a0 -- python tools/error_structure.py input_blocks.1.1.qkv 224 672 64 64 60 80 32 64
run es_a0_conv1 a0 -- python tools/error_structure.py input_blocks.1.0.in_layers.2 224 224 64 64 60 80 32 64
run es_a0_proj a0 -- python tools/error_structure.py input_blocks.1.1.proj_out 224 224 64 64 60 80 32 64
run layers_g0 g0 -- python tools/determinism_layers.py 16 80 80 32
run es_g0_mid g0 -- python tools/error_structure.py middle_block.0.in_layers.2 896 896 4 4 60 80 80 32
run es_g0_out0 g0 -- python tools/error_structure.py output_blocks.0.0.in_layers.2 1792 896 4 4 60 80 80 32
cat $OUT/mfma_valu_war.txt; cat $OUT/summary.txt
