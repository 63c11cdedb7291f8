#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3f
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r3f/tests.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
MPM_ROCTX=1 timeout -k 10 200 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f/roctx -- python3 $R/scratch/roctx_probe.py > $R/gpurun_out/r3f/roctx.log 2>&1; echo "roctx rc=$?"
tail -3 $R/gpurun_out/r3f/roctx.log
ls $R/gpurun_out/r3f/roctx/*/ | head -20
