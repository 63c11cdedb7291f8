#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3e
for ce in 4 8 16; do
  echo "== MPM_RESORT_EVERY=$ce"
  MPM_RESORT_EVERY=$ce MPM_AB_ROUNDS=2 timeout -k 10 300 python scratch/ab_run.py new 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3e/ce.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r3e/tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r3e/tests.log
