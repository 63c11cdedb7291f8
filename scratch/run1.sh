set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_world_gpu.py tests/test_domain_gpu.py tests/test_dist_gpu.py tests/test_chain_native_gpu.py -x -q > gpurun_out/r4a_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4a_tests.log
tail -30 gpurun_out/r4a_tests.log
timeout -k 10 300 python scratch/chain_cost.py > gpurun_out/r4a_chain.log 2>&1; tail -8 gpurun_out/r4a_chain.log
timeout -k 10 600 python scratch/share_scaling.py > gpurun_out/r4a_share.log 2>&1; cat gpurun_out/r4a_share.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; tail -3 gpurun_out/r4a_bench.err; cat gpurun_out/r4a_bench.json | head -c 3000
