cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04w
timeout 400 python tools/rank_cost.py --ipc-self --host-time --quarter 8 2>&1 | grep "^world\|^host\|Error\|error" | head
SBMC_RANK_COST_CPROFILE=gpurun_out/r04w/rank8_host_cprofile.txt timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world\|^host"
timeout 400 python tools/rank_cost.py --host-time 1 2>&1 | grep "^world\|^host"
