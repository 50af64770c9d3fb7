#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k/db
timeout 900 python -m pytest tests/test_gpu_backbone.py -q -x > gpurun_out/k/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/k/tests.log
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/k/db
MIOPEN_FIND_MODE=1 timeout 1500 python tools/make_miopen_db.py --layout nhwc 2>&1 | grep -v "MIOpen(HIP)" | tail -4
ls -la gpurun_out/k/db
# now a FAST-mode bench that sees these records through the package's installer
mkdir -p sbmc_amd/miopen_db && cp gpurun_out/k/db/*.ufdb.txt sbmc_amd/miopen_db/
unset MIOPEN_USER_DB_PATH
for layout in nhwc nchw auto; do
  SBMC_UNET_LAYOUT=$layout timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-stages > gpurun_out/k/bench_$layout.json 2> gpurun_out/k/bench_$layout.err
  python -c "import json;d=json.load(open('gpurun_out/k/bench_$layout.json'));print('$layout', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
