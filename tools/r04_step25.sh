cd $GRAFT_REPO_ROOT
timeout 400 python tools/rank_cost.py --ipc-self --host-time 8 2>&1 | grep "^world\|^host"
timeout 400 python tools/rank_cost.py --ipc-self --host-time 8 2>&1 | grep "^world\|^host"
SBMC_HIP_PW_GWS=1 timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s/$/ [GWS=1]/"
