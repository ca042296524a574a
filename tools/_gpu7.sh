set -u
mkdir -p gpurun_out/r2f
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2f/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r2f/smoke.log
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r2f/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2f/pytest_gpu.log
bash tools/profile_round.sh > gpurun_out/r2f/profile.log 2>&1; echo "profile rc=$?"
tail -c 600 gpurun_out/r02prof_b/bench_driver_cmd.json
