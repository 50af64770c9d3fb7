"""Two or three per-sample 1 x 1 layers in one pass (round 6; csrc/pointwise_chain.hip through the C ABI
sbmc_pointwise_chain_fwd_f32 and functions.PointwiseChain) against float64 of the same chain (reference
sbmc/modules.py:154-175 as built at sbmc/models.py:79-102), against the layer-by-layer kernels, and through autograd."""
import pytest
import torch as th

pytestmark = pytest.mark.gpu


def _chain64(x, t, s, layers, masks=None):
    """float64 of the chain; -> [y_0, y_1, ...].  masks[l] (or None): the activation's decisions (pre-activation > 0) taken
    from elsewhere instead of the float64 pre-activation's own sign -- a gradient is only comparable between two evaluations
    that took the SAME decisions, and a pre-activation within fp32 rounding of the kink decides differently in float64."""
    outs = []
    cur = x.double()
    for l, (w, bias, act, slope) in enumerate(layers):
        pre = th.matmul(w.double(), cur) + bias.double().view(1, -1, 1)
        if l == 0 and t is not None:
            tt = t.double().repeat_interleave(s, 0)
            pre = pre + (tt.unsqueeze(-1) if tt.dim() == 2 else tt)
        pos = pre > 0 if (masks is None or masks[l] is None) else masks[l]
        cur = th.where(pos, pre, pre * (0.0 if act == 1 else slope)) if act != 0 else pre
        outs.append(cur)
    return outs


def _make(b, s, cin, couts, hw, t_mode, acts, spread, dev, seed):
    th.manual_seed(seed)
    x = th.randn(b, cin, hw, device=dev) * spread
    layers = []
    k = cin
    for l, (c, a) in enumerate(zip(couts, acts)):
        w = th.randn(c, k, device=dev) / k ** 0.5
        if l == 0:
            w = w / spread
        layers.append((w, th.randn(c, device=dev) * 0.3, a, 0.01))
        k = c
    t = None
    if t_mode == 1:
        t = th.randn(b // s, couts[0], device=dev)
    elif t_mode == 2:
        t = th.randn(b // s, couts[0], hw, device=dev)
    return x, t, layers


CASES = [
    # b, s, cin, couts, hw, t_mode, acts, mean
    (8, 8, 128, (128, 128, 128), 64 * 9 + 20, 2, (1, 1, 0), True),     # a later embedding: per-pixel context, sample mean
    (4, 2, 93, (128, 128, 128), 516, 1, (1, 1, 0), True),              # the first embedding: 93 features, per-image context
    (6, 3, 128, (128, 128), 720, 2, (2, 2), False),                    # the regressor's first two layers
    (3, 1, 32, (25, 40, 17), 100, 0, (2, 1, 2), False),                # narrow layers, ragged everything
    (2, 1, 64, (96, 128), 260, 1, (1, 0), False),
    (2, 2, 128, (128, 128, 128), 4, 0, (1, 2, 1), True),               # a plane of four pixels
    (5, 1, 128, (128, 128, 128), 64 * 40, 0, (1, 1, 1), False),        # more tiles than one round of the grid at small b
    # (tools/fuzz_pointwise_chain.py's finds: layer 0's magnitude word carried a wave's maximum of layer 1 -- the words of
    # consecutive layers went through one shared array without a barrier between them)
    (6, 3, 113, (22, 74, 128), 1880, 0, (0, 1, 0), False),
    (3, 1, 124, (19, 124, 102), 1704, 1, (2, 0, 2), False),
    (9, 3, 106, (98, 28, 7), 592, 1, (1, 1, 0), False),
    (3, 3, 101, (10, 51, 83), 872, 2, (1, 0, 0), False),
]


@pytest.mark.parametrize("spread", [1.0, 1e-4])
@pytest.mark.parametrize("store_mid", [False, True])
@pytest.mark.parametrize("b,s,cin,couts,hw,t_mode,acts,mean", CASES)
def test_chain_forward_vs_float64(b, s, cin, couts, hw, t_mode, acts, mean, store_mid, spread):
    """Every output the pass writes -- intermediates (training form), the last layer, the mean, sign words, magnitude
    words -- against float64 at 1e-5 of each tensor's scale; spread: nothing may depend on the input's magnitude."""
    from sbmc_amd import functions as funcs
    dev = th.device("cuda")
    x, t, layers = _make(b, s, cin, couts, hw, t_mode, acts, spread, dev, b * 1000 + hw)
    ref = _chain64(x, t, s, layers)
    ys, signs, amaxes, ymean = funcs.pointwise_chain_forward(x, t, s, layers, store_mid=store_mid, want_signs=store_mid,
                                                             mean=mean)
    th.cuda.synchronize()
    for l, r in enumerate(ref):
        last = l + 1 == len(ref)
        if not (last or store_mid):
            assert ys[l] is None
            continue
        scale = r.abs().max().item()
        err = (ys[l].double() - r).abs().max().item()
        assert err <= 1e-5 * scale, (l, err / scale)
        assert amaxes[l].item() == ys[l].abs().max().reshape(1).view(th.int32).item(), "magnitude word is not max |y|"
        if signs[l] is not None:
            # bit i of word j of a row = (y > 0) of pixel 32 j + i; elements within rounding of the kink may differ from float64
            bits = ((signs[l].unsqueeze(-1) >> th.arange(32, device=dev)) & 1).reshape(b, couts[l], -1)[..., :hw].bool()
            assert th.equal(bits, ys[l] > 0)
    if mean:
        m = ref[-1].view(b // s, s, couts[-1], hw).mean(1)
        assert (ymean.double() - m).abs().max().item() <= 1e-5 * m.abs().max().item()


def test_chain_equals_the_separate_layers_closely():
    """The fused pass and the layer-by-layer kernels compute the same chain from the same weights: both within 1e-5 of
    float64, and of each other within 2e-6 of the output's scale (different scales, same products)."""
    from sbmc_amd import functions as funcs
    dev = th.device("cuda")
    b, s, cin, couts, hw = 8, 4, 128, (128, 128, 128), 64 * 30
    x, t, layers = _make(b, s, cin, couts, hw, 2, (1, 1, 0), 1.0, dev, 5)
    ys, _, _, ymean = funcs.pointwise_chain_forward(x, t, s, layers, mean=True)
    cur = x
    for l, (w, bias, act, slope) in enumerate(layers):
        if l + 1 < len(layers):
            cur = funcs.PointwiseLayer.apply(cur, w, bias, t if l == 0 else None, s if l == 0 else 1, act, slope)
        else:
            cur, m = funcs.PointwiseLayerMean.apply(cur, w, bias, None, 1, act, slope, s)
    scale = cur.abs().max().item()
    assert (ys[-1] - cur).abs().max().item() <= 2e-6 * scale
    assert (ymean - m).abs().max().item() <= 2e-6 * scale


@pytest.mark.parametrize("b,s,cin,couts,hw,t_mode,acts,mean", [
    (4, 2, 128, (128, 128, 128), 64 * 3 + 12, 2, (1, 1, 0), True),
    (4, 2, 93, (128, 128, 128), 200, 1, (1, 1, 0), True),
    (3, 3, 128, (128, 128), 260, 2, (2, 2), False),
])
def test_chain_autograd_vs_float64(b, s, cin, couts, hw, t_mode, acts, mean):
    """functions.PointwiseChain: outputs and EVERY gradient (input, context term, weights, biases) against float64 autograd of
    the same chain, 1e-5 of each gradient's scale."""
    from sbmc_amd import functions as funcs
    dev = th.device("cuda")
    x, t, layers = _make(b, s, cin, couts, hw, t_mode, acts, 1.0, dev, 77)
    x.requires_grad_(True)
    t.requires_grad_(True)
    wb = []
    for (w, bias, _, _) in layers:
        wb += [w.requires_grad_(True), bias.requires_grad_(True)]
    cfg = tuple((a, sl) for (_, _, a, sl) in layers)
    out = funcs.PointwiseChain.apply(x, t, s, mean, cfg, *wb)
    y, m = out if mean else (out, None)
    gy = th.randn_like(y)
    gm = th.randn_like(m) if mean else None
    leaves = [x, t] + wb
    got = th.autograd.grad([y] + ([m] if mean else []), leaves, [gy] + ([gm] if mean else []))

    # the decisions the pass took (a raw launch of the same chain in its training form)
    mids = funcs.pointwise_chain_forward(x.detach(), t.detach(), s, [(w.detach(), bb.detach(), a, sl) for (w, bb, a, sl) in layers],
                                         store_mid=True)[0]
    masks = [(m > 0) if a != 0 else None for m, (_, _, a, _) in zip(mids, layers)]
    x64, t64 = x.detach().double().requires_grad_(True), t.detach().double().requires_grad_(True)
    wb64 = [p.detach().double().requires_grad_(True) for p in wb]
    l64 = [(wb64[2 * l], wb64[2 * l + 1], layers[l][2], layers[l][3]) for l in range(len(layers))]
    y64 = _chain64(x64, t64, s, l64, masks)[-1]
    outs64, gr64 = [y64], [gy.double()]
    if mean:
        outs64.append(y64.view(b // s, s, couts[-1], hw).mean(1))
        gr64.append(gm.double())
    ref = th.autograd.grad(outs64, [x64, t64] + wb64, gr64)
    assert (y.double() - y64).abs().max().item() <= 1e-5 * y64.abs().max().item()
    for i, (a, r) in enumerate(zip(got, ref)):
        scale = r.abs().max().item()
        assert a.shape == r.shape
        assert (a.double() - r).abs().max().item() <= 1e-5 * scale, (i, (a.double() - r).abs().max().item() / scale)


def test_chain_is_what_the_embedding_runs():
    """modules.pointwise_chain_with_context hands a three-layer 1x1 ConvChain to ONE fused pass (and the regressor's first
    two layers + its wide layer), with the same result as the layer-by-layer path (SBMC_PW_CHAIN=0)."""
    import os
    from sbmc_amd import functions as funcs, modules as ops
    dev = th.device("cuda")
    th.manual_seed(3)
    chain = ops.ConvChain(128 + 128, 128, width=128, depth=3, ksize=1, pad=False).to(dev)
    chain.pointwise_as_gemm = True
    reg = ops.ConvChain(128 + 128, 441, width=128, depth=3, ksize=1, pad=False, activation="leaky_relu").to(dev)
    reg.pointwise_as_gemm = True
    per_sample = th.randn(1, 4, 128, 24, 40, device=dev)
    context = th.randn(1, 128, 24, 40, device=dev)
    store = []
    funcs.enable_kernel_timing(store)
    try:
        mean_out = []
        a = ops.pointwise_chain_with_context(chain, per_sample, context, mean_out)
        k = ops.pointwise_chain_with_context(reg, per_sample, context)
        th.cuda.synchronize()
    finally:
        funcs.enable_kernel_timing(None)
    names = sorted(n for n, _, _ in store)
    assert sum(n.startswith("pointwise_chain_fwd 128x128x128") for n in names) == 1, names
    assert sum(n.startswith("pointwise_chain_fwd 128x128<") for n in names) == 1, names
    os.environ["SBMC_PW_CHAIN"] = "0"
    try:
        mean_ref = []
        a0 = ops.pointwise_chain_with_context(chain, per_sample, context, mean_ref)
        k0 = ops.pointwise_chain_with_context(reg, per_sample, context)
    finally:
        del os.environ["SBMC_PW_CHAIN"]
    for got, ref in ((a, a0), (mean_out[0], mean_ref[0]), (k, k0)):
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()


def test_chain_at_1280x720x4spp_equals_the_separate_layers_and_itself():
    """BASELINE configs[2]'s size (every workgroup walks ~450 tiles: the prefetch of the tile after next, its waits and the
    store traffic at full rate), training form: every output against the layer-by-layer kernels, and bit-equal between two
    launches."""
    from sbmc_amd import functions as funcs
    dev = th.device("cuda")
    b, s, cin, couts, hw = 4, 4, 128, (128, 128, 128), 1280 * 720
    x, t, layers = _make(b, s, cin, couts, hw, 2, (1, 1, 0), 1.0, dev, 11)
    ys, signs, amaxes, ymean = funcs.pointwise_chain_forward(x, t, s, layers, store_mid=True, want_signs=True, mean=True)
    ys2, signs2, _, ymean2 = funcs.pointwise_chain_forward(x, t, s, layers, store_mid=True, want_signs=True, mean=True)
    for a, c in zip(ys + signs[:2] + [ymean], ys2 + signs2[:2] + [ymean2]):
        assert th.equal(a, c)
    del ys2, signs2, ymean2
    cur = x
    for l, (w, bias, act, slope) in enumerate(layers):
        if l + 1 < len(layers):
            cur = funcs.PointwiseLayer.apply(cur, w, bias, t if l == 0 else None, s if l == 0 else 1, act, slope)
        else:
            cur, m = funcs.PointwiseLayerMean.apply(cur, w, bias, None, 1, act, slope, s)
        scale = cur.abs().max().item()
        assert th.isfinite(ys[l]).all()
        assert (ys[l] - cur).abs().max().item() <= 3e-6 * scale, l
    assert (ymean - m).abs().max().item() <= 3e-6 * scale


@pytest.mark.parametrize("spread", [1.0, 1e-4])
@pytest.mark.parametrize("b,cin,cout,hw,act", [(2, 128, 441, 64 * 37 + 20, 0), (1, 128, 441, 48, 2), (3, 96, 200, 1000, 1),
                                               (2, 128, 512, 4096, 0), (1, 33, 129, 132, 0), (2, 64, 300, 64 * 600, 0)])
def test_wide_forward_vs_float64(b, cin, cout, hw, act, spread):
    """sbmc_pointwise_wide_fwd_f32 (pw_wide_fwd_kernel: the 441-channel logits layer, all row tiles from one staged tile)
    against float64 at 1e-5 of the output's scale; the magnitude word; layers of 2, 3 and 4 row tiles, ragged planes."""
    from sbmc_amd import _lib
    L = _lib.lib()
    dev = th.device("cuda")
    th.manual_seed(b * 100 + cout)
    x = th.randn(b, cin, hw, device=dev) * spread
    w = th.randn(cout, cin, device=dev) / cin ** 0.5 / spread
    bias = th.randn(cout, device=dev)
    assert L.sbmc_pointwise_wide_fwd_supported(cin, cout, hw)
    y = th.full((b, cout, hw), float("nan"), device=dev)
    amax = th.zeros(1, dtype=th.int32, device=dev)
    _lib.check(L.sbmc_pointwise_wide_fwd_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(amax), b, cin, cout,
                                             hw, act, 0.01, _lib.current_stream(dev)), "wide_fwd")
    ref = _chain64(x, None, 1, [(w, bias, act, 0.01)])[0]
    assert (y.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert amax.item() == y.abs().max().reshape(1).view(th.int32).item()


def test_wide_forward_at_full_size_equals_the_row_tile_kernel(monkeypatch):
    """1280 x 720 x 4 samples, 128 -> 441: the staged-once kernel against csrc/pointwise.hip's row-tile kernel (SBMC_PW_WIDE_FWD=0),
    and bit-equal between two launches."""
    from sbmc_amd import functions as funcs
    dev = th.device("cuda")
    th.manual_seed(9)
    x = th.randn(4, 128, 1280 * 720, device=dev)
    funcs.ensure_amax(x)
    w = th.randn(441, 128, device=dev) / 128 ** 0.5
    bias = th.randn(441, device=dev)
    with th.no_grad():
        a = funcs.PointwiseLayer.apply(x, w, bias, None, 1, 0, 0.0)
        a2 = funcs.PointwiseLayer.apply(x, w, bias, None, 1, 0, 0.0)
        assert th.equal(a, a2)
        del a2
        monkeypatch.setenv("SBMC_PW_WIDE_FWD", "0")
        c = funcs.PointwiseLayer.apply(x, w, bias, None, 1, 0, 0.0)
    assert funcs.known_amax(a) is not None and funcs.known_amax(a).item() == funcs.known_amax(c).item()
    assert (a - c).abs().max().item() <= 3e-6 * c.abs().max().item()


def _word(t):
    return t.abs().max().reshape(1).view(th.int32).clone()


@pytest.mark.parametrize("loose", [1.0, 23.0])
@pytest.mark.parametrize("b,s,cin,hw,t_mode,act1,act0,dx", [
    (4, 2, 128, 64 * 5 + 12, 2, 2, 2, True),      # the regressor's first two layers: per-pixel context gradient
    (8, 8, 128, 64 * 3, 2, 1, 1, True),           # an embedding's first two layers
    (4, 2, 93, 200, 1, 1, 1, False),              # the first embedding: network input (no data gradient), per-image context
    (3, 1, 64, 64 * 40 + 4, 0, 1, 2, True),       # several tiles per workgroup at small b
    (2, 1, 128, 4, 0, 0, 1, True),                # a plane of four pixels; the upper layer linear
])
def test_chain_pair_backward_vs_float64(b, s, cin, hw, t_mode, act1, act0, dx, loose):
    """sbmc_pointwise_chain_bwd_f32 (pw_chain_bwd_kernel): the backward of two consecutive 128-channel layers in one pass --
    gx, both weight and bias gradients, the context gradient -- against float64 autograd of the same two layers taking the
    same activation decisions, 1e-5 of each gradient's scale (weight / bias sums: helpers.close_sum).  loose: the words may
    be bounds."""
    from helpers import close_sum
    from sbmc_amd import _lib
    L = _lib.lib()
    dev = th.device("cuda")
    x, t, layers = _make(b, s, cin, (128, 128), hw, t_mode, (act0, act1), 1.0, dev, b * 7 + hw)
    (w0, b0, _, sl0), (w1, b1, _, sl1) = layers
    y = _chain64(x, t, s, layers)
    y0 = y[0].float()                                   # what the forward stored (its decisions: y0 > 0)
    y1 = y[1].float()
    wpr = (hw + 31) // 32
    bits = (y1 > 0)
    pad = th.zeros(b, 128, wpr * 32, dtype=th.bool, device=dev)
    pad[..., :hw] = bits
    signs1 = (pad.view(b, 128, wpr, 32).long() << th.arange(32, device=dev)).sum(-1)
    signs1 = th.where(signs1 >= 2 ** 31, signs1 - 2 ** 32, signs1).to(th.int32).contiguous()
    th.manual_seed(3)
    gy = th.randn(b, 128, hw, device=dev) * 1e-2
    gy[:, :, ::5] *= 20.0
    groups = L.sbmc_pointwise_chain_bwd_groups(b, s, t_mode, hw)
    nb = b // s if t_mode == 1 else 1
    gx = th.full((b, cin, hw), float("nan"), device=dev) if dx else None
    gwp1, gbp1 = th.full((groups, 128, 128), float("nan"), device=dev), th.full((groups, 128), float("nan"), device=dev)
    gwp0, gbp0 = th.full((groups, 128, cin), float("nan"), device=dev), th.full((groups, nb, 128), float("nan"), device=dev)
    gt = th.full((b // s, 128, hw), float("nan"), device=dev) if t_mode == 2 else None
    words = [(_word(v).view(th.float32) * loose).view(th.int32) for v in (gy, y0, x)]
    gxmax = th.zeros(1, dtype=th.int32, device=dev) if dx else None
    _lib.check(L.sbmc_pointwise_chain_bwd_f32(
        _lib.ptr(gy), _lib.ptr(signs1) if act1 else None, _lib.ptr(y0), _lib.ptr(w1), _lib.ptr(x), _lib.ptr(w0),
        _lib.ptr(gx) if dx else None, _lib.ptr(gwp1), _lib.ptr(gbp1), _lib.ptr(gwp0), _lib.ptr(gbp0),
        _lib.ptr(gt) if gt is not None else None, _lib.ptr(words[0]), _lib.ptr(words[1]), _lib.ptr(words[2]),
        _lib.ptr(gxmax) if dx else None, b, s, cin, hw, t_mode, act1, sl1, act0, sl0, _lib.current_stream(dev)), "chain_bwd")
    # float64 with the same decisions
    gz1 = gy.double() * (th.where(bits, 1.0, 0.0 if act1 == 1 else sl1) if act1 else 1.0)
    gy0 = th.einsum("ok,bop->bkp", w1.double(), gz1)
    gz0 = gy0 * (th.where(y0 > 0, 1.0, 0.0 if act0 == 1 else sl0) if act0 else 1.0)
    y0d, xd = y0.double(), x.double()
    close_sum(gwp1.sum(0), th.einsum("bop,bkp->ok", gz1, y0d), th.einsum("bop,bkp->ok", gz1.abs(), y0d.abs()), what="gw1")
    close_sum(gbp1.sum(0), gz1.sum((0, 2)), gz1.abs().sum((0, 2)), what="gb1")
    close_sum(gwp0.sum(0), th.einsum("bop,bkp->ok", gz0, xd), th.einsum("bop,bkp->ok", gz0.abs(), xd.abs()), what="gw0")
    close_sum(gbp0.sum((0, 1)), gz0.sum((0, 2)), gz0.abs().sum((0, 2)), what="gb0")
    if t_mode == 1:
        close_sum(gbp0.sum(0), gz0.view(b // s, s, 128, hw).sum((1, 3)), gz0.abs().view(b // s, s, 128, hw).sum((1, 3)), what="gt")
    elif t_mode == 2:
        r = gz0.view(b // s, s, 128, hw).sum(1)
        assert (gt.double() - r).abs().max().item() <= 1e-5 * r.abs().max().item()
    if dx:
        r = th.einsum("ok,bop->bkp", w0.double(), gz0)
        assert (gx.double() - r).abs().max().item() <= 1e-5 * r.abs().max().item()
        assert gxmax.item() == _word(gx).item()


def test_chain_autograd_with_the_fused_pair_backward(monkeypatch):
    """SBMC_PW_CHAIN_BWD=1 (off by default: slower at full size, functions._chain_pair_backward): PointwiseChain's backward
    takes layers 1 and 0 in one pass -- same gradients as layer by layer."""
    from sbmc_amd import functions as funcs
    dev = th.device("cuda")
    b, s, cin, hw = 4, 2, 128, 64 * 6 + 8
    x, t, layers = _make(b, s, cin, (128, 128), hw, 2, (2, 2), 1.0, dev, 21)
    x.requires_grad_(True)
    t.requires_grad_(True)
    funcs.ensure_amax(x)
    wb = []
    for (w, bias, _, _) in layers:
        wb += [w.requires_grad_(True), bias.requires_grad_(True)]
    cfg = tuple((a, sl) for (_, _, a, sl) in layers)
    gy = th.randn(b, 128, hw, device=dev)
    funcs.ensure_amax(gy)
    res = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("SBMC_PW_CHAIN_BWD", knob)
        calls = []
        funcs.enable_kernel_timing(calls)
        try:
            y = funcs.PointwiseChain.apply(x, t, s, False, cfg, *wb)
            res[knob] = th.autograd.grad(y, [x, t] + wb, gy)
        finally:
            funcs.enable_kernel_timing(None)
        assert any(n.startswith("pointwise_chain_bwd") for n, _, _ in calls) == (knob == "1")
    for a, c in zip(res["1"], res["0"]):
        assert (a - c).abs().max().item() <= 5e-6 * c.abs().max().item()
