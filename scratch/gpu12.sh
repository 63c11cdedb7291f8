#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3k
timeout -k 10 600 python -m pytest tests/test_domain_gpu.py tests/test_dist_gpu.py tests/test_chain_native_gpu.py -m gpu -q > gpurun_out/r3k/tests.log 2>&1; echo "tests rc=$?"
grep -v "^E    " gpurun_out/r3k/tests.log | tail -5
MPM_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 --config cloth_8m --dt 2e-4 > gpurun_out/r3k/b8m.json 2> gpurun_out/r3k/b8m.err; echo "bench 8m x2 rc=$?"; tail -2 gpurun_out/r3k/b8m.err | cut -c1-300; cut -c1-1200 gpurun_out/r3k/b8m.json
