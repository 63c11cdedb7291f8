#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3i
export MPM_BENCH_SHARE_GPU=1
timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r3i/self2.json 2> gpurun_out/r3i/self2.err; echo "self-launch rc=$?"; cut -c1-700 gpurun_out/r3i/self2.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r3i/torchrun2.json 2> gpurun_out/r3i/torchrun2.err; echo "torchrun rc=$?"; cut -c1-700 gpurun_out/r3i/torchrun2.json
unset MPM_BENCH_SHARE_GPU
timeout -k 10 400 python bench.py --config cloth_8m --dt 2e-4 --steps 20 --warmup 5 --no-cpu-baseline --no-contact-leg > gpurun_out/r3i/8m.json 2> gpurun_out/r3i/8m.err; echo "8m rc=$?"; cut -c1-900 gpurun_out/r3i/8m.json
