#!/bin/bash
for v in fem0 fem1; do echo "== $v"; MPM_HIP_LIBRARY=$PWD/drake_amd/variants/libmpm_hip_$v.so timeout -k 10 300 python scratch/prec_scene.py 6 1e-3 3,0.5,0.5,20 -1,0.5,0.3,20 2>&1 | grep -v amdgpu; done
