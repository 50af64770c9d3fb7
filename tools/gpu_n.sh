#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/n/db
timeout 900 python -m pytest tests/test_gpu_slab.py tests/test_dist_gpu.py tests/test_gpu_backbone.py -q -x > gpurun_out/n/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/n/tests.log
# find records for the slab shapes of 2 / 4 / 8 ranks at 1280x720 (an edge rank and an interior rank each)
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/n/db
cp sbmc_amd/miopen_db/*.ufdb.txt gpurun_out/n/db/
for spec in "2 0" "4 0" "4 1" "8 0" "8 1" "8 4" "8 7"; do
  set -- $spec
  MIOPEN_FIND_MODE=1 timeout 900 python tools/make_miopen_db.py --layout nhwc --ranks $1 --rank $2 2>&1 | grep "find + one step"
done
ls -la gpurun_out/n/db; wc -l gpurun_out/n/db/*.txt
cp gpurun_out/n/db/*.ufdb.txt sbmc_amd/miopen_db/
unset MIOPEN_USER_DB_PATH
timeout 600 python tools/rank_cost.py 1 2 4 8 > gpurun_out/n/rank_cost.txt 2>&1; grep world gpurun_out/n/rank_cost.txt
SBMC_UNET_LAYOUT=nchw timeout 600 python tools/rank_cost.py 8 > gpurun_out/n/rank_cost_nchw.txt 2>&1; grep world gpurun_out/n/rank_cost_nchw.txt
