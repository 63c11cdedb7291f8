cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_world_gpu.py tests/test_domain_gpu.py tests/test_dist_gpu.py tests/test_chain_native_gpu.py -q -s -p no:faulthandler > gpurun_out/r4c_world.log 2>&1; echo "world rc $?" >> gpurun_out/r4c_world.log
grep -v "^\s*$" gpurun_out/r4c_world.log | grep -i "world x\|passed\|failed\|error\|rc \|assert" | head -40
for n in 2 4; do
MPM_BENCH_SHARE_GPU=1 MPM_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus $n --steps 25 --warmup 5 > gpurun_out/r4c_b$n.json 2> gpurun_out/r4c_b$n.err; echo "bench $n rc $?"; tail -c 1500 gpurun_out/r4c_b$n.json; grep -i "error\|fail\|Traceback" gpurun_out/r4c_b$n.err | head -5
done
MPM_BENCH_SHARE_GPU=1 MPM_BENCH_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 4 --config cloth_8m --dt 2e-4 --steps 10 --warmup 3 > gpurun_out/r4c_b8m.json 2> gpurun_out/r4c_b8m.err; echo "bench 8m rc $?"; tail -c 1500 gpurun_out/r4c_b8m.json; grep -i "error\|fail\|Traceback" gpurun_out/r4c_b8m.err | head -5
timeout -k 10 300 python scratch/chain_cost.py 2>&1 | grep "us/substep"
