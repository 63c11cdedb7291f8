cd $GRAFT_REPO_ROOT
export MPM_BENCH_SHARE_GPU=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4m_torchrun2.json 2> gpurun_out/r4m_torchrun2.err; echo "torchrun rc $?"
tail -c 600 gpurun_out/r4m_torchrun2.json; echo; grep -i "native RCCL\|RCCL halo\|transport\|error" gpurun_out/r4m_torchrun2.err | head -8
python3 -c "
import json; d=json.load(open('gpurun_out/r4m_torchrun2.json')); print(d['transport'], d['value'], d['ms_per_step'], d['timed_region'])"
