#!/bin/bash
MPM_AB_ROUNDS=3 timeout -k 10 600 python scratch/ab_run.py r02 cur2 p2gr02 2>&1 | grep -v amdgpu.ids
