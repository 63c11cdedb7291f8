#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3h
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -v "^E    " gpurun_out/r3h/tests.log | tail -8
[ $rc -eq 0 ] && timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r3h/bench.json
