cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s/$/ [default]/"
SBMC_CONV3X3_STREAMK=0 timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s/$/ [SBMC_CONV3X3_STREAMK=0]/"
done
