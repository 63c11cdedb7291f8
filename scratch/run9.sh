cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -m gpu -q -p no:faulthandler > gpurun_out/r4g_suite.log 2>&1; echo "suite rc $?" >> gpurun_out/r4g_suite.log
tail -12 gpurun_out/r4g_suite.log
