cd $GRAFT_REPO_ROOT
for r in 1 2; do timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world"; done
timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world"
bash tools/prof_rank.sh 8 --ipc-self > gpurun_out/prof_rank.log 2>&1
python tools/prof_rank_cat.py gpurun_out/q/rank8_stats.csv 5 > gpurun_out/r05_rank8_categories.txt; cp gpurun_out/q/rank8_stats.csv gpurun_out/r05_rank8_kernel_stats.csv; rm -rf gpurun_out/q
head -50 gpurun_out/r05_rank8_categories.txt | cut -c1-160
