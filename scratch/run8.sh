cd $GRAFT_REPO_ROOT
MPM_AB_ROUNDS=2 timeout -k 10 900 python scratch/ab_run.py cur dyn 2>&1 | grep '^{'
