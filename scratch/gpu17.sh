#!/bin/bash
for k in 1 2 4 8 1000; do MPM_RESORT_EVERY=$k timeout -k 10 200 python scratch/idle_cost.py 2>&1 | grep -v amdgpu; done
