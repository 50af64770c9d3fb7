"""Backward of the regressor's / an embedding's first two 1x1 layers at 1280x720 x 8 spp: the fused pair
(csrc/pointwise_chain_bwd.hip) beside the two layer-by-layer launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch as th
from sbmc_amd import functions as funcs
dev = th.device("cuda")
HW, S = 1280 * 720, 8
th.manual_seed(0)
for name, cin, t_mode, acts, dx in (("128+ctx->128->128 (gx, gt)", 128, 2, (2, 2), True), ("93+gf->128->128 (no gx)", 93, 1, (1, 1), False)):
    x = th.randn(S, cin, HW, device=dev).requires_grad_(dx); funcs.ensure_amax(x)
    t = (th.randn(1, 128, HW, device=dev) if t_mode == 2 else th.randn(1, 128, device=dev)).requires_grad_(True)
    wb = []
    k = cin
    for _ in range(2):
        wb += [(th.randn(128, k, device=dev) / k ** 0.5).requires_grad_(True), (th.randn(128, device=dev) * 0.1).requires_grad_(True)]
        k = 128
    cfg = tuple((a, 0.01) for a in acts)
    gy = th.randn(S, 128, HW, device=dev); funcs.ensure_amax(gy)
    for knob in ("1", "0"):
        os.environ["SBMC_PW_CHAIN_BWD"] = knob
        y = funcs.PointwiseChain.apply(x, t, S, False, cfg, *wb)
        leaves = ([x] if dx else []) + [t] + wb
        def run():
            return th.autograd.grad(y, leaves, gy, retain_graph=True)
        for _ in range(2): r = run()
        th.cuda.synchronize(); a, c = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True); a.record()
        for _ in range(5): r = run()
        c.record(); th.cuda.synchronize()
        print("%-30s SBMC_PW_CHAIN_BWD=%s: %.3f ms per backward" % (name, knob, a.elapsed_time(c) / 5), flush=True)
        if knob == "1": keep = [v.clone() for v in r]
        else:
            print("    max rel diff fused vs separate:", ["%.1e" % ((u - v).abs().max() / v.abs().max()).item() for u, v in zip(keep, r)])
    del x, t, wb, gy, y, r, keep
    th.cuda.empty_cache()
