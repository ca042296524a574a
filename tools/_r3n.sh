O=gpurun_out/r3n; mkdir -p $O
timeout 300 python tools/loop_chain_sweep.py 32 2:80,3:54,4:40 1000 32 > $O/chains_default.md 2>/dev/null
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/loop_chain_sweep.py 32 2:80,3:54,4:40 1000 32 > $O/chains_hwq8.md 2>/dev/null
cat $O/chains_default.md $O/chains_hwq8.md
B="--no-trace --no-cpu-baseline --no-e2"
for c in c2 c4 c5; do
timeout 500 python bench.py --steps 20 --warmup 2 --config $c $B > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; python -c "
import json; d=json.load(open('$O/bench_$c.json')); print(d['value'], d['ms_per_step'], d['time_share'], d['roofline']['achieved'], d['config']['decoder_fwd_queries_per_shape'], d['roofline_loop']['one_loop_alone_ms_per_evaluation'], d['roofline_loop']['latents_per_loop'])"
done
