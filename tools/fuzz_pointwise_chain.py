"""Randomised sweep of the fused 1x1 chains (csrc/pointwise_chain.hip) against float64: random layer counts, widths, plane sizes
(ragged tiles), context forms, activations, magnitudes; the wide forward and the pair backward with them.
    python tools/fuzz_pointwise_chain.py [--seconds N] [--seed S]"""
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sbmc_amd import _lib, functions as funcs  # noqa: E402
from test_gpu_pointwise_chain import _chain64  # noqa: E402

dev = th.device("cuda")
L = _lib.lib()
seconds = float(sys.argv[sys.argv.index("--seconds") + 1]) if "--seconds" in sys.argv else 60.0
seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 0
g = th.Generator().manual_seed(seed)
ri = lambda lo, hi: int(th.randint(lo, hi + 1, (1,), generator=g))
worst = {"chain": 0.0, "wide": 0.0}
n = {"chain": 0, "wide": 0}
t0 = time.time()
while time.time() - t0 < seconds:
    s = ri(1, 4)
    b = s * ri(1, 3)
    hw = 4 * ri(1, 700)
    spread = 10.0 ** ri(-4, 2)
    th.manual_seed(ri(0, 10 ** 6))
    if ri(0, 3) == 0:
        cin, cout, act = ri(1, 128), ri(129, 512), ri(0, 2)
        x = th.randn(b, cin, hw, device=dev) * spread
        w = th.randn(cout, cin, device=dev) / cin ** 0.5 / spread
        bias = th.randn(cout, device=dev)
        y = th.full((b, cout, hw), float("nan"), device=dev)
        amax = th.zeros(1, dtype=th.int32, device=dev)
        _lib.check(L.sbmc_pointwise_wide_fwd_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(amax), b, cin, cout,
                                                 hw, act, 0.01, _lib.current_stream(dev)), "wide")
        ref = _chain64(x, None, 1, [(w, bias, act, 0.01)])[0]
        e = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        assert e <= 1e-5, ("wide", b, cin, cout, hw, act, e)
        assert amax.item() == y.abs().max().reshape(1).view(th.int32).item(), \
            ("wide amax", b, cin, cout, hw, act, amax.view(th.float32).item(), y.abs().max().item())
        worst["wide"] = max(worst["wide"], e)
        n["wide"] += 1
        continue
    nl, cin, t_mode = ri(2, 3), ri(1, 128), ri(0, 2)
    couts = [ri(1, 128) for _ in range(nl)]
    acts = [ri(0, 2) for _ in range(nl)]
    x = th.randn(b, cin, hw, device=dev) * spread
    layers, k = [], cin
    for l in range(nl):
        w = th.randn(couts[l], k, device=dev) / k ** 0.5 / (spread if l == 0 else 1.0)
        layers.append((w, th.randn(couts[l], device=dev) * 0.3, acts[l], 0.01))
        k = couts[l]
    t = None if t_mode == 0 else (th.randn(b // s, couts[0], device=dev) if t_mode == 1 else th.randn(b // s, couts[0], hw, device=dev))
    train, mean = bool(ri(0, 1)), bool(ri(0, 1))
    ys, signs, amaxes, ymean = funcs.pointwise_chain_forward(x, t, s, layers, store_mid=train, want_signs=train, mean=mean)
    ref = _chain64(x, t, s, layers)
    for l in range(nl):
        if ys[l] is None:
            continue
        e = ((ys[l].double() - ref[l]).abs().max() / ref[l].abs().max().clamp(min=1e-300)).item()
        assert e <= 1e-5, ("chain", b, s, cin, couts, hw, t_mode, acts, train, l, e)
        assert amaxes[l].item() == ys[l].abs().max().reshape(1).view(th.int32).item(), \
            ("chain amax", b, s, cin, couts, hw, t_mode, acts, train, mean, l, amaxes[l].view(th.float32).item(), ys[l].abs().max().item())
        worst["chain"] = max(worst["chain"], e)
        if signs[l] is not None:
            bits = ((signs[l].unsqueeze(-1) >> th.arange(32, device=dev)) & 1).reshape(b, couts[l], -1)[..., :hw].bool()
            assert th.equal(bits, ys[l] > 0)
    if mean:
        m = ref[-1].view(b // s, s, couts[-1], hw).mean(1)
        assert ((ymean.double() - m).abs().max() / m.abs().max().clamp(min=1e-300)).item() <= 1e-5
    n["chain"] += 1
print("fuzz_pointwise_chain: %d chains (worst %.2e of the output's scale), %d wide forwards (worst %.2e) in %.0f s, seed %d: none beyond 1e-5"
      % (n["chain"], worst["chain"], n["wide"], worst["wide"], time.time() - t0, seed))
