#!/bin/bash
# The round-end evidence in ONE GPU call (same box): the whole GPU suite + smoke(), the default bench line, tools/prof.sh (rocprofv3
# kernel statistics of the same commands + the PMC passes) and the step's categories:   tools/grun --timeout 3300 "bash tools/final_evidence.sh"
cd $GRAFT_REPO_ROOT
bash tools/gpu_suite.sh
mkdir -p gpurun_out/final
( time timeout 1500 python bench.py > gpurun_out/final/r05_bench.json 2> gpurun_out/final/bench.err ); head -c 400 gpurun_out/final/r05_bench.json; echo
bash tools/prof.sh r05 > gpurun_out/final/prof.log 2>&1; cp gpurun_out/profiles_r05/* gpurun_out/final/; rm -rf gpurun_out/prof_r05 gpurun_out/profiles_r05
python tools/prof_rank_cat.py gpurun_out/final/r05_model_kernel_stats.csv 11 > gpurun_out/final/r05_model_categories.txt; head -14 gpurun_out/final/r05_model_categories.txt
grep -E "splat_(bwd|fwd)_strip" gpurun_out/final/r05_model_kernel_stats.csv | cut -c1-160
