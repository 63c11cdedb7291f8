cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_world_gpu.py tests/test_domain_gpu.py tests/test_dist_gpu.py tests/test_chain_native_gpu.py -q -s -p no:faulthandler > gpurun_out/r4b_world.log 2>&1; echo "world rc $?" >> gpurun_out/r4b_world.log
grep -v "^\s*$" gpurun_out/r4b_world.log | grep -i "world x\|passed\|failed\|error\|rc \|assert" | head -60
timeout -k 10 600 python scratch/share_scaling.py > gpurun_out/r4b_share.log 2>&1; grep "items<=48" gpurun_out/r4b_share.log
