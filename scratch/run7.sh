cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_chain_native_gpu.py tests/test_ieee_variant_gpu.py "tests/test_contact_gpu.py::test_twenty_newton_iterations_against_the_float_noise_of_the_iteration" "tests/test_contact_gpu.py::test_five_newton_iterations_match_oracle" tests/test_domain_gpu.py -q -s -p no:faulthandler > gpurun_out/r4f.log 2>&1; echo "rc $?" >> gpurun_out/r4f.log
grep -v "^\s*$" gpurun_out/r4f.log | grep -i "20 iter\|scene\|config\|passed\|failed\|error\|rc \|assert" | head -40
