#!/bin/bash
# Why does the in-run launch time of the 8-sample splat kernels differ between bench runs (4.42 vs 5.01 ms) while rocprofv3 of
# `--workload splat` on the same box reads 4.43?   tools/grun --timeout 1500 bash tools/roofline_variance.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rv
pick() { python - "$1" <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); k=d.get('kernels',{})
        print(sys.argv[1].split('/')[-1], 'ms/step', d.get('ms_per_step'), {n:(v['avg_ms'],v['GBps']) for n,v in k.items()})
P
}
export SBMC_BENCH_DUMP_KERNELS=1
timeout 600 python bench.py --workload splat --no-cpu-baseline > gpurun_out/rv/a_splat_only.json 2> gpurun_out/rv/a.err; pick gpurun_out/rv/a_splat_only.json
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/rv/b_default.json 2> gpurun_out/rv/b.err; pick gpurun_out/rv/b_default.json
SBMC_BENCH_SPLAT_COOLDOWN=15 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/rv/c_cooldown.json 2> gpurun_out/rv/c.err; pick gpurun_out/rv/c_cooldown.json
timeout 600 python bench.py --workload splat --no-cpu-baseline > gpurun_out/rv/d_splat_only.json 2> gpurun_out/rv/d.err; pick gpurun_out/rv/d_splat_only.json
