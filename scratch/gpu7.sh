#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3g
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3g/tests.log 2>&1; echo "tests rc=$?"
grep -v "^E    " gpurun_out/r3g/tests.log | tail -24
