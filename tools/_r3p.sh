O=gpurun_out/r3p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decoder_grid.py -x -q -m gpu > $O/pytest_grid.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_grid.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-trace --no-e2 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['time_share'], d['breakdown_ms_per_step'])"
timeout 400 python bench.py --steps 20 --warmup 2 --workload trace --endpoint e2 --no-trace --no-cpu-baseline > $O/bench_trace_e2.json 2> $O/bench_trace_e2.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench_trace_e2.json')); print(d['value'], d['ms_per_step'], d['time_share'], d['e2'])"
