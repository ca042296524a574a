#!/bin/bash
# latency form at 8 latents: the split-K sizing knobs of launch_conv2 re-swept on the round-6 kernel (one box, ms per evaluation)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ks_sweep; mkdir -p $O; : > $O/sweep.txt
run() { env "$@" timeout 300 python tools/loop_batch_sweep.py 32 8 200 0 2>&1 | grep "^| 32" | sed "s/^/$* /" >> $O/sweep.txt; }
run SURFD_X=0
run SURFD_CONV2_FILL=384
run SURFD_CONV2_FILL=640
run SURFD_CONV2_FILL=768
run SURFD_CONV2_NOSPLIT_ABOVE=120
run SURFD_CONV2_NOSPLIT_ABOVE=300
run SURFD_CONV2_NOSPLIT_ABOVE=400
run SURFD_CONV2_KSMAX=8
run SURFD_CONV2_KSMAX=24
run SURFD_CONV2_PREF=1
run SURFD_X=1
cat $O/sweep.txt
