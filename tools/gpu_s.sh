#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s/db
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/s/db
cp sbmc_amd/miopen_db/*.ufdb.txt gpurun_out/s/db/
MIOPEN_FIND_MODE=1 timeout 1500 python tools/make_miopen_db.py --layout nhwc --fp16 2>&1 | grep -v "MIOpen(HIP)" | tail -3
wc -l gpurun_out/s/db/*.txt
cp gpurun_out/s/db/*.ufdb.txt sbmc_amd/miopen_db/
unset MIOPEN_USER_DB_PATH
for layout in auto nchw; do
SBMC_UNET_LAYOUT=$layout timeout 600 python bench.py --fp16-activations --steps 8 --warmup 4 --no-cpu-baseline --no-stages 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('train fp16 $layout', d['value'], d['ms_per_step'])"
SBMC_UNET_LAYOUT=$layout timeout 600 python bench.py --workload infer --spp 32 --fp16-activations --steps 8 --warmup 4 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('infer32 fp16 $layout', d['value'], d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "fp16 or autocast" 2>&1 | tail -2
