#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3n
for i in 1 2; do timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r3n/full$i.log 2>&1; echo "full run $i rc=$?"; tail -1 gpurun_out/r3n/full$i.log; done
T="tests/test_run_substeps_gpu.py tests/test_parity_gpu.py tests/test_capacity_gpu.py tests/test_configs_gpu.py tests/test_contact_gpu.py tests/test_precision_gpu.py"
MPM_QUIET_FACTOR=1 timeout -k 10 500 python -m pytest $T -x -q -m gpu > gpurun_out/r3n/quiet1.log 2>&1; echo "quiet=1 rc=$?"; tail -1 gpurun_out/r3n/quiet1.log
MPM_ITEM_GROUPS=2 timeout -k 10 500 python -m pytest $T -x -q -m gpu > gpurun_out/r3n/items2.log 2>&1; echo "item_groups=2 rc=$?"; tail -1 gpurun_out/r3n/items2.log
