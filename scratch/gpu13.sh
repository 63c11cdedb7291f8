#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3l
MPM_AB_ROUNDS=2 timeout -k 10 600 python scratch/ab_run.py fem0 fem1 fem2 2>&1 | grep -v amdgpu.ids | cut -c1-230
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3l/tests.log 2>&1; echo "tests rc=$?"
grep -v "^E    " gpurun_out/r3l/tests.log | tail -22
