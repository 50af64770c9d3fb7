cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r04c/tests.txt
timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" > gpurun_out/r04c/rank_cost.txt
timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" >> gpurun_out/r04c/rank_cost.txt
cat gpurun_out/r04c/tests.txt gpurun_out/r04c/rank_cost.txt
