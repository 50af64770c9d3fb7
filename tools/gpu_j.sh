#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/j/bench_nocpu.json 2> gpurun_out/j/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/j/bench_nocpu.json"))
print(d["value"], d["ms_per_step"], json.dumps(d["stages"]["splat_all_samples"]), json.dumps(d["roofline"])[:400])
print(json.dumps(d["kernels"]))
PY
bash tools/prof.sh r02 > gpurun_out/j/prof.log 2>&1; echo "prof rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/j/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/j/gpu_tests.log
bash tools/gpu_i.sh
