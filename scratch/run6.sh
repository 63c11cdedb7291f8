cd $GRAFT_REPO_ROOT
timeout -k 10 600 python scratch/ct20.py 2>&1 | grep "iters"
timeout -k 10 1100 python -m pytest tests/test_ieee_variant_gpu.py -q -s -p no:faulthandler 2>&1 | grep -i "scene\|config\|passed\|failed\|error\|assert" | head -20
