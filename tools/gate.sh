#!/bin/bash
# The gate in front of every kernel commit (round 5 validated kernel commits with the bench, which cannot see a 1e-3
# error): the FULL oracle suite on the GPU + run-to-run bit stability of every instantiation.  Runs on the GPU box:
#     gpurun --timeout 2400 -- 'bash tools/gate.sh [label]'
# and leaves gpurun_out/gate_<label>.txt (copy it to profiles/ with the commit it gates).
cd "$(dirname "$0")/.." || exit 1
LABEL=${1:-$(date +%H%M)}
OUT=gpurun_out/gate_$LABEL.txt; mkdir -p gpurun_out
{
  echo "gate $LABEL: $(python -c 'from surfd_amd import _native as N; print(N.lib().surfd_build_config().decode())' 2>&1 | tail -1)"
  echo "sha256 of the kernel sources:"; sha256sum surfd_amd/csrc/*.hip surfd_amd/csrc/*.h surfd_amd/csrc/*.cpp | cut -c1-16,65-
  echo "== pytest -m gpu (full suite)"
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25
  echo "pytest rc=${PIPESTATUS[0]}"
  echo "== determinism_check: 40 evaluations of one input, every form"
  for c in "8 0" "80 80" "160 160"; do timeout 300 python tools/determinism_check.py 40 $c 2>&1 | grep -E "distinct|differs"; done
  for c in "80 64 32" "40 64 80"; do timeout 300 python tools/diag_l64.py $c 2>&1 | grep -E '^\{' | cut -c1-330; done
} > $OUT 2>&1
cat $OUT
