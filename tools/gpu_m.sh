#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m
timeout 900 python -m pytest tests/test_gpu_backbone.py -q -x > gpurun_out/m/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/m/tests.log
for layout in auto nchw; do
  SBMC_UNET_LAYOUT=$layout timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-stages > gpurun_out/m/bench_$layout.json 2> gpurun_out/m/bench_$layout.err
  python -c "import json;d=json.load(open('gpurun_out/m/bench_$layout.json'));print('$layout', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
timeout 600 python bench.py --workload infer --spp 8 --steps 10 --warmup 4 > gpurun_out/m/infer8.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/m/infer8.json'));print('infer8', d['value'], d['ms_per_step'])"
