#!/bin/bash
MPM_AB_ROUNDS=2 timeout -k 10 600 python scratch/ab_run.py base p2g10 2>&1 | grep -v amdgpu.ids | cut -c1-230
