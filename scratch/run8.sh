cd $GRAFT_REPO_ROOT
export MPM_AB_ROUNDS=1
timeout -k 10 300 python scratch/ab_run.py now noprio 2>&1 | grep '^{'
for ig in 32 40 64; do echo "item groups $ig"; MPM_ITEM_GROUPS=$ig timeout -k 10 300 python scratch/ab_run.py now 2>&1 | grep '^{'; done
for wg in 640 768 1024; do echo "p2g wgs $wg"; MPM_P2G_WGS=$wg timeout -k 10 300 python scratch/ab_run.py now 2>&1 | grep '^{'; done
