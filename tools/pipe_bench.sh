#!/bin/bash
# pipelined vs sequential bench, decoder grid sizes.  usage: pipe_bench.sh "<bench args>" ...
for cfg in "$@"; do
  echo "== $cfg"; timeout 300 python bench.py $cfg --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), 'shapes/s', round(d['ms_per_step']), 'ms/step', d['breakdown_ms_per_step'], round(d['roofline']['frac'],3))"
done
