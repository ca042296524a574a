O=gpurun_out/r3d; mkdir -p $O
timeout 400 python bench.py --steps 6 --warmup 1 --timeline --no-trace --no-e2 --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err; echo "rc=$?"; tail -4 $O/bench_short.err; cut -c1-400 $O/bench_short.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --timeline > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc=$?"; tail -4 $O/bench_driver.err; cut -c1-300 $O/bench_driver.json
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
