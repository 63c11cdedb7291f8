#!/bin/bash
# first GPU call of round 3: correctness of the new face layout, then same-box A/B of the P2G variants
set -o pipefail
mkdir -p gpurun_out/r3a
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3a/summary.txt
tail -3 gpurun_out/r3a/tests.log
timeout -k 10 500 python scratch/ab_run.py r02 new nohoist_noldsf r02p2g hoist > gpurun_out/r3a/ab.log 2>&1; echo "ab rc=$?" | tee -a gpurun_out/r3a/summary.txt
cat gpurun_out/r3a/ab.log
MPM_HIP_LIBRARY=$PWD/drake_amd/variants/libmpm_hip_diag.so MPM_DBG=4 timeout -k 10 120 python scratch/p2g_diag.py > gpurun_out/r3a/diag.log 2>&1; echo "diag rc=$?" | tee -a gpurun_out/r3a/summary.txt
cat gpurun_out/r3a/diag.log
