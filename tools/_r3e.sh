O=gpurun_out/r3e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_decoder_grid.py -x -q -m gpu -k "band_mesher or two_gpus" -s > $O/pytest_band.log 2>&1; echo "pytest rc=$?"; grep -E "band mesher at|passed|failed|Error" $O/pytest_band.log | head
B="--no-trace --no-cpu-baseline"
timeout 500 python bench.py --steps 6 --warmup 1 --endpoint e2 $B > $O/bench_e2.json 2> $O/bench_e2.err; echo "e2 rc=$?"; tail -3 $O/bench_e2.err; python -c "
import json; d=json.load(open('$O/bench_e2.json')); print(d['value'], d['e2'])"
timeout 500 python bench.py --steps 6 --warmup 1 --workload trace $B --no-e2 > $O/bench_trace.json 2> $O/bench_trace.err; echo "trace rc=$?"; tail -3 $O/bench_trace.err; python -c "
import json; d=json.load(open('$O/bench_trace.json')); print(d['value'], d['time_share'], d['breakdown_ms_per_step'])"
timeout 500 python bench.py --steps 6 --warmup 1 --workload trace --endpoint e2 $B > $O/bench_trace_e2.json 2> $O/bench_trace_e2.err; echo "trace e2 rc=$?"; tail -3 $O/bench_trace_e2.err; python -c "
import json; d=json.load(open('$O/bench_trace_e2.json')); print(d['value'], d['e2'])"
for c in c2 c4 c5; do
timeout 500 python bench.py --steps 4 --warmup 1 --config $c $B --no-e2 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; tail -3 $O/bench_$c.err; python -c "
import json; d=json.load(open('$O/bench_$c.json')); print(d['value'], d['ms_per_step'], d['time_share'], d['roofline']['achieved'], d['config']['decoder_fwd_queries_per_shape'], d['roofline_loop']['one_loop_alone_ms_per_evaluation'], d['roofline_loop']['latents_per_loop'])"
done
timeout 300 python bench.py --steps 1 --warmup 0 --mode grid-shard --batch 2 > $O/bench_gridshard.json 2> $O/bench_gridshard.err; echo "gs rc=$?"; tail -3 $O/bench_gridshard.err; cut -c1-600 $O/bench_gridshard.json
