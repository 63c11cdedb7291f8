#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3c
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r3c/tests.log
MPM_AB_ROUNDS=3 timeout -k 10 500 python scratch/ab_run.py r02 new noldsf > gpurun_out/r3c/ab.log 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids gpurun_out/r3c/ab.log
