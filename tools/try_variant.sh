#!/bin/bash
# a variant build (tools/build_variants.py) through the denoiser's GPU tests and the latency-form A/B: bash tools/try_variant.sh <name>
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/try_$1; mkdir -p $O
SURFD_LIB=$PWD/surfd_amd/lib/variants/libsurfd_hip_$1.so timeout 900 python -m pytest tests/test_gpu_unet.py -x -q 2>&1 | tail -8 > $O/pytest_unet.txt
bash tools/ab_variant.sh $1 > $O/ab.txt 2>&1
cat $O/pytest_unet.txt $O/ab.txt
