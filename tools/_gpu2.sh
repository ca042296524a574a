set -u
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_decoder_grid.py tests/test_attention.py -x -q -m gpu -s > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "directions|passed|failed|Error|error" gpurun_out/r2g/pytest.log | tail -30
timeout 300 python bench.py --steps 1 --warmup 0 --diffusion-steps 20 --no-cpu-baseline --no-e2 > gpurun_out/r2g/trace.json 2> gpurun_out/r2g/trace.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r2g/trace.json') if l.startswith('{')][-1])
print(json.dumps(r.get('w_trace'),indent=0)[:1500]); print(r['breakdown_ms_per_step'])
PY
