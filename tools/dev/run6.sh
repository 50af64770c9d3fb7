cd $GRAFT_REPO_ROOT
bash tools/gpu_suite.sh
for r in 1 2; do timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world"; done
timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world"
python tools/bench_pw_scaled.py 2>&1 | tail -9
