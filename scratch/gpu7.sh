#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3g
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3g/tests.log 2>&1; echo "tests rc=$?"
grep -v "^E    " gpurun_out/r3g/tests.log | tail -8
MPM_AB_ROUNDS=2 timeout -k 10 300 python scratch/ab_run.py r02 2>&1 | grep -v amdgpu.ids
MPM_HIP_LIBRARY=$PWD/drake_amd/libmpm_hip.so MPM_AB_ROUNDS=1 timeout -k 10 300 python scratch/ab_run.py cur cur 2>&1 | grep -v amdgpu.ids
