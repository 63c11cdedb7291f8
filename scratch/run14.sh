cd $GRAFT_REPO_ROOT
MPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/drake_amd/variants/libmpm_hip_both.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_precision_gpu.py -q -p no:faulthandler 2>&1 | grep -E "^FAILED|Error|assert " | head -20
