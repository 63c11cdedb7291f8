#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3k
timeout -k 10 600 python -m pytest tests/test_domain_gpu.py tests/test_dist_gpu.py tests/test_chain_native_gpu.py -m gpu -q > gpurun_out/r3k/tests.log 2>&1; echo "tests rc=$?"
grep -v "^E    " gpurun_out/r3k/tests.log | tail -8
MPM_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r3k/b2.json 2> gpurun_out/r3k/b2.err; echo "bench2 rc=$?"; tail -2 gpurun_out/r3k/b2.err; cut -c1-900 gpurun_out/r3k/b2.json
MPM_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 4 --steps 20 --warmup 5 > gpurun_out/r3k/b4.json 2> gpurun_out/r3k/b4.err; echo "bench4 rc=$?"; tail -2 gpurun_out/r3k/b4.err; cut -c1-700 gpurun_out/r3k/b4.json
