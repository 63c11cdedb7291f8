cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_ieee_variant_gpu.py tests/test_world_gpu.py "tests/test_contact_gpu.py::test_five_newton_iterations_match_oracle" -q -s -p no:faulthandler > gpurun_out/r4e.log 2>&1; echo "rc $?" >> gpurun_out/r4e.log
grep -v "^\s*$" gpurun_out/r4e.log | grep -i "IEEE\|config 2\|world x\|passed\|failed\|error\|rc \|assert" | head -40
