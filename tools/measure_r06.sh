#!/bin/bash
# Round 6 measurement set (one gpurun call): gate, the driver's command + rocprof + PMC passes, the other configurations,
# grid-shard mode, the gradient two-pass bound.  Outputs under gpurun_out/r06/.
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
bash tools/gate.sh r06_final > $O/gate.log 2>&1
PROF_DIR=$O/prof bash tools/profile_round.sh > $O/profile_round.log 2>&1
for c in c2 c4 c5; do timeout 900 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 900 python bench.py --mode grid-shard --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_gridshard_native_1rank.json 2> $O/bench_gridshard.err
SURFD_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --mode grid-shard --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_gridshard_native_2ranks_gloo_one_gpu.json 2> $O/bench_gridshard2.err
timeout 600 python tools/grad_two_pass_bound.py 22 > $O/grad_two_pass_bound.json 2> $O/grad_two_pass_bound.err
timeout 300 python tools/loop_ab.py 100 8 80 80 > $O/loop_ab_L32.json 2>&1
timeout 300 python tools/loop_ab.py 50 8 80 80 64 > $O/loop_ab_L64.json 2>&1
ls -la $O $O/prof | head -60
tail -5 $O/gate.log
