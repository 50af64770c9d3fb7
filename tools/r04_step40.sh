cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04x
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r04x/gpu_suite.txt 2>&1; tail -5 gpurun_out/r04x/gpu_suite.txt
bash tools/measure_round.sh r04
