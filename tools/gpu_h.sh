#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h
( time timeout 1500 python bench.py > gpurun_out/h/r02_bench.json 2> gpurun_out/h/r02_bench.err ); echo "bench rc=$?"
head -c 600 gpurun_out/h/r02_bench.json; echo
timeout 900 python bench.py --workload infer --spp 4 > gpurun_out/h/r02_bench_infer4.json 2>/dev/null; head -c 300 gpurun_out/h/r02_bench_infer4.json; echo
timeout 900 python bench.py --workload infer --spp 8 > gpurun_out/h/r02_bench_infer8.json 2>/dev/null; head -c 300 gpurun_out/h/r02_bench_infer8.json; echo
timeout 900 python bench.py --workload infer --spp 32 > gpurun_out/h/r02_bench_infer32.json 2>/dev/null; head -c 300 gpurun_out/h/r02_bench_infer32.json; echo
timeout 900 python scripts/bench_ops.py > gpurun_out/h/r02_bench_ops.jsonl 2>/dev/null; tail -3 gpurun_out/h/r02_bench_ops.jsonl
bash tools/prof.sh r02 > gpurun_out/h/prof.log 2>&1; echo "prof rc=$?"; tail -5 gpurun_out/h/prof.log
