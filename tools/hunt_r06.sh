#!/bin/bash
# Round 6: the "second workgroup on a CU" instability of conv2_kernel — debug-variant matrix on the GPU box.
# Output: gpurun_out/hunt/*.txt   (libraries: tools/build_variants.py, names below)
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/hunt; mkdir -p $OUT
V=surfd_amd/lib/variants
run() { # name, lib, env..., -- cmd
    local name=$1 lib=$2; shift 2
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    echo "== $name ($lib ${envs[*]}) :: $*" >> $OUT/summary.txt
    env SURFD_LIB=$PWD/$V/libsurfd_hip_$lib.so "${envs[@]}" timeout 300 "$@" > $OUT/$name.txt 2>&1
    echo "rc=$?" >> $OUT/$name.txt
    grep -E '^\{|distinct|differs|rc=' $OUT/$name.txt | cut -c1-600 >> $OUT/summary.txt
}
for v in a0 a_fz a_pad a_s2v a_p1 a_p2 a_O1; do run l64_$v $v -- python tools/diag_l64.py 80 64 32; done
run l64_a0_1percu a0 SURFD_CONV2_LDS_EXTRA=20000 -- python tools/diag_l64.py 80 64 32
run l64_a0_again a0 -- python tools/diag_l64.py 80 64 32
run probe_a a_probe -- python tools/probe_phases.py 8 80 64 32
run probe_d64 d_probe -- python tools/probe_phases.py 8 80 64 32
for v in g0 g_fz g_pad g_s2v g_p1; do run det_$v $v -- python tools/determinism_check.py 30 80 80; done
run probe_g g_probe -- python tools/probe_phases.py 12 80 32 80
run probe_d32 d_probe -- python tools/probe_phases.py 8 80 32 80
cat $OUT/summary.txt
