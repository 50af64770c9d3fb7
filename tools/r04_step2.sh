cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_wbank.py tests/test_gpu_halo.py tests/test_gpu_conv3x3.py tests/test_dist_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/r04b/tests.txt
timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | tee gpurun_out/r04b/rank_cost.txt
timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | tee -a gpurun_out/r04b/rank_cost.txt
SBMC_WBANK=0 timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed 's/$/ [SBMC_WBANK=0]/' | tee -a gpurun_out/r04b/rank_cost.txt
bash tools/prof_rank.sh 8 --ipc-self > gpurun_out/r04b/prof_rank.log 2>&1
cp gpurun_out/q/rank8_stats.csv gpurun_out/r04b/rank8_kernel_stats.csv
python tools/prof_rank_cat.py gpurun_out/r04b/rank8_kernel_stats.csv > gpurun_out/r04b/rank8_categories.txt
head -20 gpurun_out/r04b/rank8_categories.txt
rm -rf gpurun_out/q
