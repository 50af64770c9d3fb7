cd $GRAFT_REPO_ROOT
o=gpurun_out/r04u; mkdir -p $o
for i in 1 2 3 4; do
SBMC_HALO_TIMEOUT_S=20 SBMC_BENCH_BACKEND=gloo SBMC_BENCH_SINGLE_DEVICE=1 OMP_NUM_THREADS=8 timeout 900 python bench.py --gpus 8 --steps 3 --warmup 2 --no-cpu-baseline --no-stages > $o/bench8_$i.json 2> $o/bench8_$i.err.txt; echo "run $i rc=$?"
python - <<PY
import json
for l in open("$o/bench8_$i.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["ms_per_step"], d["transport"], d["transport_note"], d["validation"]["rel_diff"])
PY
grep -c "gave up\|HaloTimeout" $o/bench8_$i.err.txt
done
