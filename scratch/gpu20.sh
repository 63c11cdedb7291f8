#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3j
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3j/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -v "^E    " gpurun_out/r3j/tests.log | tail -5
[ $rc -eq 0 ] && for i in 1 2 3; do timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-contact-leg 2> gpurun_out/r3j/bench$i.err | tee gpurun_out/r3j/bench$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['timed_region'], d['steady_state'], d['roofline']['frac'], d['roofline']['substep_frac'])"; done
