#!/bin/bash
# final lines of the round (after the PMC traffic profile of the final kernel sources is committed: roofline.traffic quotes it)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06b; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
for c in c2 c4 c5; do timeout 900 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 900 python bench.py --workload trace --endpoint e2 --steps 20 --warmup 2 --no-cpu-baseline --no-strict --no-trace-e2 > $O/bench_trace_e2_timed.json 2> $O/bench_trace.err
ls -la $O
