O=gpurun_out/r3t; mkdir -p $O
SURFD_CONV2_WT2=1 timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "wide or phased" > $O/pytest_wt2.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_wt2.log
timeout 300 python tools/loop_chain_sweep.py 32 1:80,2:80 1000 32 2>/dev/null | tail -3
SURFD_CONV2_WT2=1 timeout 300 python tools/loop_chain_sweep.py 32 1:80,2:80,3:54 1000 32 2>/dev/null | tail -4
SURFD_CONV2_WT2=1 timeout 300 python tools/loop_chain_sweep.py 64 1:80,2:80 1000 32 2>/dev/null | tail -3
timeout 300 python tools/loop_chain_sweep.py 64 2:80 1000 32 2>/dev/null | tail -2
