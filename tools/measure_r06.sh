#!/bin/bash
# Round 6 measurement set in ONE gpurun call, in the order that lets the driver line quote its own round's counters:
#   gate (full GPU suite + determinism)  ->  PMC passes on the shortened command  ->  profiles/r06_pmc_{summary,traffic}.json on the box
#   ->  the driver's command (its roofline.traffic now names this profile; kernel sources identical by construction)
#   ->  the same command under rocprofv3 --kernel-trace --stats  ->  C2 / C4 / C5, grid-shard, W-trace E2
# Outputs under gpurun_out/r06/ (copy to profiles/ with the r06_ prefix).
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
bash tools/gate.sh r06_final > $O/gate.log 2>&1
PMC_ONLY=1 PROF_DIR=$O/prof bash tools/profile_round.sh > $O/profile_round.log 2>&1
cp $O/prof/pmc_summary.json profiles/r06_pmc_summary.json && python tools/pmc_to_traffic.py profiles/r06_pmc_summary.json profiles/r06_pmc_traffic.json > /dev/null
cp profiles/r06_pmc_summary.json profiles/r06_pmc_traffic.json $O/
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-trace --no-e2 --no-strict --no-trace-e2 > $O/bench_under_rocprof.json 2> $O/kt.log; find $O/kt -name '*kernel_trace.csv' -delete )
for c in c2 c4 c5; do timeout 900 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 900 python bench.py --mode grid-shard --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_gridshard_native_1rank.json 2> $O/bench_gridshard.err
timeout 900 python bench.py --workload trace --endpoint e2 --steps 20 --warmup 2 --no-cpu-baseline --no-strict --no-trace-e2 > $O/bench_trace_e2_timed.json 2> $O/bench_trace.err
timeout 300 python tools/loop_ab.py 100 8 80 80 2>/dev/null | grep '^{' > $O/loop_ab_L32.json
ls -la $O $O/prof $O/kt | head -60
tail -6 $O/gate.log
