mkdir -p gpurun_out/r3b
timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "wide" > gpurun_out/r3b/pytest_wide.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest_wide.log
tail -5 gpurun_out/r3b/pytest_wide.log
timeout 300 python tools/loop_batch_sweep.py 32 8,16,32,64 1000 32 > gpurun_out/r3b/l32_wide32.md 2> gpurun_out/r3b/l32_wide32.err
SURFD_CONV2_WIDE_PREF=1 timeout 300 python tools/loop_batch_sweep.py 32 32,64 1000 32 > gpurun_out/r3b/l32_wide32_pref.md 2> gpurun_out/r3b/l32_wide32_pref.err
timeout 300 python tools/loop_batch_sweep.py 32 32,64 1000 8 > gpurun_out/r3b/l32_wide8.md 2> gpurun_out/r3b/l32_wide8.err
timeout 300 python tools/loop_batch_sweep.py 32 64 1000 128 > gpurun_out/r3b/l32_wide128.md 2> gpurun_out/r3b/l32_wide128.err
cat gpurun_out/r3b/*.md
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3b/prof -- python $GRAFT_REPO_ROOT/tools/loop_batch_sweep.py 32 64 50 32 > $GRAFT_REPO_ROOT/gpurun_out/r3b/prof.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r3b/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-200 {} | head -12'
find $GRAFT_REPO_ROOT/gpurun_out/r3b/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $GRAFT_REPO_ROOT/gpurun_out/r3b/prof -name "*.db" -delete
