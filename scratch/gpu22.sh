#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3k
T="tests/test_run_substeps_gpu.py tests/test_parity_gpu.py tests/test_capacity_gpu.py tests/test_configs_gpu.py tests/test_contact_gpu.py"
MPM_QUIET_FACTOR=1 timeout -k 10 500 python -m pytest $T -x -q -m gpu > gpurun_out/r3k/quiet1.log 2>&1; echo "quiet=1 rc=$?"; tail -1 gpurun_out/r3k/quiet1.log
MPM_ITEM_GROUPS=2 timeout -k 10 500 python -m pytest $T -x -q -m gpu > gpurun_out/r3k/items2.log 2>&1; echo "item_groups=2 rc=$?"; tail -1 gpurun_out/r3k/items2.log
MPM_POISON=1 timeout -k 10 500 python -m pytest $T -x -q -m gpu > gpurun_out/r3k/poison.log 2>&1; echo "poison rc=$?"; tail -1 gpurun_out/r3k/poison.log
