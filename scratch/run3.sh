cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_world_gpu.py tests/test_domain_gpu.py tests/test_dist_gpu.py tests/test_chain_native_gpu.py -q -p no:faulthandler > gpurun_out/r4h_world.log 2>&1; echo "world rc $?" >> gpurun_out/r4h_world.log
grep -i "passed\|failed\|error\|rc " gpurun_out/r4h_world.log | head
for n in 2 4; do
MPM_BENCH_SHARE_GPU=1 MPM_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus $n --steps 25 --warmup 5 > gpurun_out/r4h_b$n.json 2> gpurun_out/r4h_b$n.err; echo "bench $n rc $?"
python3 -c "
import json,sys
d=json.load(open('gpurun_out/r4h_b$n.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus','transport')}, d['timed_region'], d['config']['geometry'])"
grep -i "error\|fail\|Traceback" gpurun_out/r4h_b$n.err | head -5
done
