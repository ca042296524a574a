#!/bin/bash
# first runs of the sixteen-wave latency form: the denoiser's GPU tests, then the loop timing with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/lat16_${1:-0}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q 2>&1 | tail -25 > $O/pytest_unet.txt
for e in 1 0; do SURFD_CONV2_LAT16=$e timeout 300 python tools/loop_batch_sweep.py 32 8,4,2,16 200 0 2>&1 | grep "^| 32" | sed "s/^/lat16=$e /" >> $O/sweep.txt; done
cat $O/pytest_unet.txt $O/sweep.txt
