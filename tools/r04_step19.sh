cd $GRAFT_REPO_ROOT
o=gpurun_out/r04s; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_backbone.py -q -k "wide or regressor" 2>&1 | tail -3
timeout 600 python bench.py --no-stages --no-cpu-baseline --steps 6 > $o/bench_quick.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("$o/bench_quick.json") if l.startswith("{")][0])
print(d["ms_per_step"]); print({k:v["avg_ms"] for k,v in d["pointwise_layers"].items()})
PY
