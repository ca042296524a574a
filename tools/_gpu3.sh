set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2t
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python bench.py --steps 8 --warmup 0 --no-trace --no-e2 --no-cpu-baseline > gpurun_out/r2t/bench.json 2> gpurun_out/r2t/bench.err
tail -c 400 gpurun_out/r2t/bench.json | head -c 400; echo
python tools/pipeline_timeline.py /tmp/kt gpurun_out/r2t/timeline.json > gpurun_out/r2t/timeline.txt 2>&1; tail -5 gpurun_out/r2t/timeline.txt
