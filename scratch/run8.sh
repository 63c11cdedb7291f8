cd $GRAFT_REPO_ROOT
export MPM_AB_ROUNDS=2
timeout -k 10 600 python scratch/ab_run.py now peel fix50 both 2>&1 | grep '^{'
