#!/bin/bash
for v in fem0 fem1 fem2; do
MPM_HIP_LIBRARY=$PWD/drake_amd/variants/libmpm_hip_$v.so MPM_PRECISION_REPORT_ONLY=1 timeout -k 10 600 python -m pytest tests/test_precision_gpu.py -m gpu -q > /dev/null 2>&1
cp gpurun_out/precision_report.txt gpurun_out/precision_$v.txt
done
