#!/bin/bash
for v in fem0 fem1; do echo "== $v"; MPM_HIP_LIBRARY=$PWD/drake_amd/variants/libmpm_hip_$v.so timeout -k 10 300 python scratch/prec256.py 2>&1 | grep -v amdgpu; done
