#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3d
MPM_AB_ROUNDS=2 timeout -k 10 600 python scratch/ab_run.py r02 new noact nofold noirng allold > gpurun_out/r3d/ab.log 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids gpurun_out/r3d/ab.log
