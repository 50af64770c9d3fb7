cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | tee gpurun_out/r04a/rank_cost.txt
timeout 400 python tools/rank_cost.py 8 2>&1 | grep "^world" | tee -a gpurun_out/r04a/rank_cost.txt
timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | tee -a gpurun_out/r04a/rank_cost.txt
bash tools/prof_rank.sh 8 --ipc-self > gpurun_out/r04a/prof_rank.log 2>&1
cp gpurun_out/q/rank8_stats.csv gpurun_out/r04a/rank8_kernel_stats.csv
python tools/prof_rank_cat.py gpurun_out/r04a/rank8_kernel_stats.csv | tee gpurun_out/r04a/rank8_categories.txt
rm -rf gpurun_out/q
