O=gpurun_out/r3final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['time_share'], d['e2']['status'][:40], d['cpu_baseline']['value'])"
