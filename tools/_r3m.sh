O=gpurun_out/r3m; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['time_share'], d['breakdown_ms_per_step'], d['w_trace'], d['e2'], d['cpu_baseline']['value'])"
timeout 400 python bench.py --steps 20 --warmup 2 --workload trace --endpoint e2 --no-trace --no-cpu-baseline > $O/bench_trace_e2.json 2> $O/bench_trace_e2.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench_trace_e2.json')); print(d['value'], d['ms_per_step'], d['time_share'], d['e2'])"
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
