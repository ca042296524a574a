#!/bin/bash
# latency form at 8 latents: phase stamps of every conv launch (workgroup 0) and a per-launch kernel trace of one fused loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lat${1:-0}; mkdir -p $O
SURFD_LIB=$PWD/surfd_amd/lib/variants/libsurfd_hip_stamps.so SURFD_CONV_DEBUG=1 timeout 300 python tools/debug_conv2_phases.py 8 0 > $O/stamps_latency_B8.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/loop_batch_sweep.py 32 8 30 0 > $O/sweep_under_rocprof.txt 2>&1
f=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/loop_layer_trace.py $f 100 > $O/layer_trace_B8.txt 2>&1
rm -rf $O/kt
timeout 300 python tools/loop_batch_sweep.py 32 8 200 0 > $O/sweep_B8.txt 2>&1
tail -5 $O/sweep_B8.txt; tail -3 $O/stamps_latency_B8.txt; head -12 $O/layer_trace_B8.txt
