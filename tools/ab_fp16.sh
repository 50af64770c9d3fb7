cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for lib in "" "$PWD/.ab/libsbmc_head.so"; do
  SBMC_HIP_LIB=$lib timeout 400 python bench.py --fp16-activations --no-cpu-baseline --no-stages --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d['pointwise_layers']; print('${lib:-tree}'[-20:], d['ms_per_step'], p['pointwise_bwd_f16 128x128']['avg_ms'], p['pointwise_bwd_f16 128x93 (no gx)']['avg_ms'])"
done; done
