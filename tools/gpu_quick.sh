#!/bin/bash
# A subset of the GPU suite + one bench line:   tools/grun --timeout 1500 bash tools/gpu_quick.sh "<pytest args>" [bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/quick
( time timeout 1200 python -m pytest $1 -q -rf -x ) > gpurun_out/quick/tests.txt 2>&1; grep -E "^(FAILED|ERROR)|^E  |passed|failed" gpurun_out/quick/tests.txt | tail -30
shift
if [ -n "$1" ]; then timeout 900 python bench.py "$@" > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/quick/bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step") if k in d}, d.get("roofline", {}).get("frac"))
    for k in ("stages",):
        if k in d: print(k, d[k])
except Exception as e:
    print("bench failed:", e); print(open("gpurun_out/quick/bench.err").read()[-2000:])
PY
fi
