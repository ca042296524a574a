set -u
mkdir -p gpurun_out/r2s
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2s/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r2s/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2s/pytest_gpu.log
timeout 1100 python tools/sweep_pipeline.py gpurun_out/r2s 10 2>&1 | tail -16
