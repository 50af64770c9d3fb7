cd $GRAFT_REPO_ROOT
o=gpurun_out/r04p; mkdir -p $o
for i in 1 2; do
SBMC_BENCH_BACKEND=gloo SBMC_BENCH_SINGLE_DEVICE=1 OMP_NUM_THREADS=8 timeout 1200 python bench.py --gpus 8 --steps 3 --warmup 2 --no-cpu-baseline --no-stages > $o/bench8_$i.json 2> $o/bench8_$i.err.txt; echo "8 ranks on one GPU run $i rc=$?"
grep -v "^\[Gloo\|^MIOpen\|socket.cpp\|amdgpu.ids\|^W0\|^\*\*\*\|OMP_NUM" $o/bench8_$i.err.txt | tail -25
python - <<PY
import json
for l in open("$o/bench8_$i.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["ms_per_step"], d["transport"], d["transport_note"], d["validation"], [r["handshake_ms"] for r in d["per_rank"]])
PY
done
timeout 900 python -m pytest tests/test_gpu_bench_cli.py tests/test_dist_gpu.py -q 2>&1 | tail -3
