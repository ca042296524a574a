set -u
mkdir -p gpurun_out/r2v
for v in ${VARIANTS:-O C}; do
  cp surfd_amd/lib/variants/lib_$v.so surfd_amd/lib/libsurfd_hip.so
  timeout 300 python bench.py --steps 1 --warmup 0 --diffusion-steps 20 --no-cpu-baseline --no-e2 > gpurun_out/r2v/trace_$v.json 2> gpurun_out/r2v/trace_$v.err
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
try:
    r=json.loads([l for l in open(f'gpurun_out/r2v/trace_{v}.json') if l.startswith('{')][-1])
    w=r['w_trace']; print(v, 'fwd TF', round(w['fwd_algorithmic_tflops'],1), 'fwd+bwd TF', round(w['fwd_bwd_algorithmic_tflops'],1), 'real fwd ms/step', round(r['breakdown_ms_per_step']['decoder_fwd'],1), 'roof', round(r['roofline']['achieved'],1))
except Exception as e:
    print(v,'failed',e); print(open(f'gpurun_out/r2v/trace_{v}.err').read()[-800:])
PY
done
if [ -n "${TESTLIB:-}" ]; then
  cp surfd_amd/lib/variants/lib_$TESTLIB.so surfd_amd/lib/libsurfd_hip.so
  timeout 600 python -m pytest tests/test_gpu_decoder_grid.py -x -q -m gpu 2>&1 | tail -4
fi
