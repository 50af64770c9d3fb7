import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch as th
from sbmc_amd import functions as funcs
from test_gpu_pointwise_chain import _make, _chain64
dev = th.device("cuda")
for (b, s, cin, couts, hw, t_mode, acts, mean) in [(4, 2, 128, (128, 128, 128), 204, 2, (1, 1, 0), True), (4, 2, 128, (128, 128, 128), 204, 2, (1, 1, 0), False), (3, 3, 128, (128, 128), 260, 2, (2, 2), False), (4, 2, 93, (128, 128, 128), 200, 1, (1, 1, 0), True)]:
    x, t, layers = _make(b, s, cin, couts, hw, t_mode, acts, 1.0, dev, 77)
    x.requires_grad_(True); t.requires_grad_(True)
    wb = []
    for (w, bias, _, _) in layers:
        wb += [w.requires_grad_(True), bias.requires_grad_(True)]
    cfg = tuple((a, sl) for (_, _, a, sl) in layers)
    out = funcs.PointwiseChain.apply(x, t, s, mean, cfg, *wb)
    y, m = out if mean else (out, None)
    th.manual_seed(1)
    gy = th.randn_like(y); gm = th.randn_like(m) if mean else None
    leaves = [x, t] + wb
    got = th.autograd.grad([y] + ([m] if mean else []), leaves, [gy] + ([gm] if mean else []))
    # separate layers
    cur = x
    for l, (w, bias, act, slope) in enumerate(layers):
        last = l + 1 == len(layers)
        if last and mean:
            cur, mm = funcs.PointwiseLayerMean.apply(cur, w, bias, None, 1, act, slope, s)
        else:
            cur = funcs.PointwiseLayer.apply(cur, w, bias, t if l == 0 else None, s if l == 0 else 1, act, slope)
    sep = th.autograd.grad([cur] + ([mm] if mean else []), leaves, [gy] + ([gm] if mean else []))
    x64, t64 = x.detach().double().requires_grad_(True), t.detach().double().requires_grad_(True)
    wb64 = [p.detach().double().requires_grad_(True) for p in wb]
    l64 = [(wb64[2 * l], wb64[2 * l + 1], layers[l][2], layers[l][3]) for l in range(len(layers))]
    y64 = _chain64(x64, t64, s, l64)[-1]
    outs64, gr64 = [y64], [gy.double()]
    if mean:
        outs64.append(y64.view(b // s, s, couts[-1], hw).mean(1)); gr64.append(gm.double())
    ref = th.autograd.grad(outs64, [x64, t64] + wb64, gr64)
    print("case", couts, "mean", mean)
    for i, (a, c, r) in enumerate(zip(got, sep, ref)):
        sc = r.abs().max().item()
        print("  grad %d  chain err %.2e   separate err %.2e" % (i, (a.double() - r).abs().max().item() / sc, (c.double() - r).abs().max().item() / sc))
