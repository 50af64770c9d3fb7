cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_gpu_bench_cli.py tests/test_gpu_halo.py tests/test_gpu_wbank.py -q 2>&1 | tail -15 > gpurun_out/r04d/tests.txt
cat gpurun_out/r04d/tests.txt
( time timeout 1500 python bench.py > gpurun_out/r04d/bench.json 2> gpurun_out/r04d/bench.err ) 2>&1 | tail -3
head -c 600 gpurun_out/r04d/bench.json; echo
tail -3 gpurun_out/r04d/bench.err
