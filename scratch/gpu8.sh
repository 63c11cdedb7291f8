#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3h
timeout -k 10 900 python -m pytest tests/test_precision_gpu.py tests/test_contact_gpu.py -m gpu -q -s > gpurun_out/r3h/tests.log 2>&1; echo "tests rc=$?"
grep -v "^E    " gpurun_out/r3h/tests.log | tail -60
