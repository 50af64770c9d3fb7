#!/bin/bash
# A/B of alternative library builds (sbmc_amd/libab_*.so) on the splat workload, interleaved rounds
for i in 1 2 3; do for l in "" $(ls sbmc_amd/libab_*.so 2>/dev/null); do SBMC_HIP_LIB=$l python bench.py --workload splat --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('${l:-base}', d['value'], d['kernels']['splat_update_fwd']['avg_ms'], d['kernels']['splat_update_bwd']['avg_ms'])"; done; done
