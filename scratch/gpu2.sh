#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3b
MPM_HIP_LIBRARY=$PWD/drake_amd/variants/libmpm_hip_diag.so MPM_DBG=4 timeout -k 10 120 python scratch/p2g_diag.py > gpurun_out/r3b/diag.log 2>&1; echo "diag rc=$?"
cat gpurun_out/r3b/diag.log
