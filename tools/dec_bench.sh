#!/bin/bash
# forward decoder kernel, both arithmetics, on a short bench (256^3 grids, 10-step DDIM)
for mode in ${@:-fp32 f16x2}; do
echo "== $mode"; timeout 200 python bench.py --decoder-precision $mode --resolution 256 --diffusion-steps 10 --steps 2 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['breakdown_ms_per_step'], d['roofline']['algorithmic_tflops'], d['roofline']['frac'])"
done
