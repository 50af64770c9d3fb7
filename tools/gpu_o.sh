#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/o/db
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/o/db
cp sbmc_amd/miopen_db/*.ufdb.txt gpurun_out/o/db/
for spec in "8 0" "8 1" "8 4" "8 7"; do
  set -- $spec
  MIOPEN_FIND_MODE=1 timeout 1200 python tools/make_miopen_db.py --layout nhwc --ranks $1 --rank $2 --4k 2>&1 | grep "find + one step"
done
# inference at 4 / 32 spp uses the same U-net shapes as training (bs = 1): nothing more to find at 720p
wc -l gpurun_out/o/db/*.txt
cp gpurun_out/o/db/*.ufdb.txt sbmc_amd/miopen_db/
unset MIOPEN_USER_DB_PATH
timeout 900 python tools/rank_cost.py --4k 8 > gpurun_out/o/rank_cost_4k.txt 2>&1; grep world gpurun_out/o/rank_cost_4k.txt
SBMC_UNET_LAYOUT=nchw timeout 900 python tools/rank_cost.py --4k 8 > gpurun_out/o/rank_cost_4k_nchw.txt 2>&1; grep world gpurun_out/o/rank_cost_4k_nchw.txt
