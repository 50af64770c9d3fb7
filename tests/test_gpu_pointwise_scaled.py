"""The 1 x 1 layers in the 3 x 3 kernels' number format (round 5; csrc/pointwise.hip pw_fwd_s_kernel / pw_bwd_kernel <.., F2>:
two f16 planes under power-of-two scales taken from magnitude words, three of the four partial products) through the
C ABI (sbmc_pointwise_{fwd,bwd}_scaled_f32) against float64 of the same layer, and the magnitude words themselves."""
import pytest
import torch as th

pytestmark = pytest.mark.gpu


def _word(t):
    """bit pattern of max |t| as the device word a producing pass would have left"""
    return t.abs().max().reshape(1).view(th.int32).clone()


def _layer64(x, w, bias, t, s, act, slope):
    pre = th.matmul(w.double(), x.double()) + bias.double().view(1, -1, 1)
    if t is not None:
        tt = t.double().repeat_interleave(s, 0)
        pre = pre + (tt.unsqueeze(-1) if tt.dim() == 2 else tt)
    if act == 1:
        return th.relu(pre), pre
    if act == 2:
        return th.where(pre > 0, pre, pre * slope), pre
    return pre, pre


CASES = [
    # b, s, cin, cout, hw, t_mode, act, mean
    (2, 1, 128, 128, 1024, 0, 2, False),
    (8, 8, 128, 128, 64 * 9 + 20, 0, 1, True),        # the embedding's last layer with the sample mean
    (6, 3, 128, 128, 720, 2, 1, False),               # first layer of a later embedding: per-pixel context
    (6, 3, 128, 128, 516, 2, 2, True),
    (4, 2, 96, 128, 516, 1, 1, False),                # first layer of the first embedding
    (3, 1, 32, 25, 100, 0, 0, False),
    (2, 1, 64, 96, 260, 1, 2, False),
    (2, 1, 128, 441, 64 * 5 + 8, 0, 0, False),        # the logits layer: four row tiles
    (1, 1, 128, 128, 4, 0, 2, False),
]


@pytest.mark.parametrize("spread", [1.0, 1e-4])
@pytest.mark.parametrize("b,s,cin,cout,hw,t_mode,act,mean", CASES)
def test_scaled_forward_vs_float64(b, s, cin, cout, hw, t_mode, act, mean, spread):
    """spread: magnitude of the input (the scale is a power of two from the word: nothing may depend on it)."""
    from sbmc_amd import _lib
    L = _lib.lib()
    dev = th.device("cuda")
    th.manual_seed(b * 1000 + hw)
    x = th.randn(b, cin, hw, device=dev) * spread
    x[0, 0, 0] = 7.5 * spread                          # the largest magnitude, known
    w = th.randn(cout, cin, device=dev) / cin ** 0.5 / spread
    bias = th.randn(cout, device=dev)
    t = None
    if t_mode == 1:
        t = th.randn(b // s, cout, device=dev)
    elif t_mode == 2:
        t = th.randn(b // s, cout, hw, device=dev)
    slope = 0.01
    ref, pre = _layer64(x, w, bias, t, s, act, slope)
    for scaled in (True, False):
        y = th.full((b, cout, hw), float("nan"), device=dev)
        wpr = (hw + 31) // 32
        signs = th.zeros(b, cout, wpr, dtype=th.int32, device=dev) if act != 0 else None
        ymean = th.full((b // s, cout, hw), float("nan"), device=dev) if mean else None
        xmax = _word(x) if scaled else None
        amax = th.zeros(1, dtype=th.int32, device=dev)
        rc = L.sbmc_pointwise_fwd_scaled_f32(
            _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(t) if t is not None else None, _lib.ptr(y),
            _lib.ptr(signs) if signs is not None else None, _lib.ptr(ymean) if mean else None, s if mean else 1,
            _lib.ptr(xmax) if scaled else None, _lib.ptr(amax), b, s, cin, cout, hw, t_mode, act, slope,
            _lib.current_stream(dev))
        _lib.check(rc, "fwd_scaled")
        scale = ref.abs().max().item()
        # the pre-activation is a sum of cin products: held to 1e-5 of sum |w||x| per element would be generous; as the
        # other kernels, 1e-5 of the tensor's scale
        err = (y.double() - ref).abs().max().item()
        assert err <= 1e-5 * scale, ("scaled" if scaled else "three planes", err / scale)
        assert amax.item() == _word(y).item(), "magnitude word is not max |y|"
        if mean:
            m = ref.view(b // s, s, cout, hw).mean(1)
            assert (ymean.double() - m).abs().max().item() <= 1e-5 * scale
        if signs is not None:
            bits = (signs.unsqueeze(-1) >> th.arange(32, device=dev, dtype=th.int32)) & 1
            bits = bits.reshape(b, cout, wpr * 32)[..., :hw].bool()
            assert int((bits != (y > 0)).sum()) == 0


def test_scaled_forward_with_a_loose_word_and_zeros():
    """The word is an upper BOUND (a mean's word is its parent's; a bound 100 x too large costs 7 bits of the 2^-39 floor,
    nothing else); an all-zero input has word 0."""
    from sbmc_amd import _lib
    L = _lib.lib()
    dev = th.device("cuda")
    th.manual_seed(5)
    b, cin, cout, hw = 2, 128, 128, 512
    x = th.randn(b, cin, hw, device=dev)
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    bias = th.randn(cout, device=dev)
    ref, _ = _layer64(x, w, bias, None, 1, 0, 0.0)
    for xx, word in ((x, _word(x * 100)), (th.zeros_like(x), th.zeros(1, dtype=th.int32, device=dev))):
        y = th.empty(b, cout, hw, device=dev)
        amax = th.zeros(1, dtype=th.int32, device=dev)
        rc = L.sbmc_pointwise_fwd_scaled_f32(_lib.ptr(xx), _lib.ptr(w), _lib.ptr(bias), None, _lib.ptr(y), None, None, 1,
                                             _lib.ptr(word), _lib.ptr(amax), b, 1, cin, cout, hw, 0, 0, 0.0,
                                             _lib.current_stream(dev))
        _lib.check(rc, "fwd_scaled")
        want = ref if xx is x else bias.double().view(1, -1, 1).expand(b, cout, hw)
        assert (y.double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()


BWD_CASES = [
    # b, s, cin, cout, hw, t_mode, act, gm, dx
    (2, 1, 128, 128, 1024, 0, 2, False, True),
    (8, 8, 128, 128, 64 * 9 + 20, 0, 0, True, True),  # embedding's last (linear) layer with the mean's gradient
    (8, 8, 128, 128, 64 * 3 + 4, 0, 1, True, True),
    (6, 3, 128, 128, 720, 2, 1, False, True),         # per-pixel context gradient
    (4, 2, 96, 128, 516, 1, 1, False, False),         # first layer of the first embedding: no data gradient
    (3, 1, 32, 25, 100, 0, 0, False, True),
    (2, 1, 64, 96, 260, 1, 2, False, True),
]


@pytest.mark.parametrize("spread", [1.0, 1e-3])
@pytest.mark.parametrize("b,s,cin,cout,hw,t_mode,act,gm,dx", BWD_CASES)
def test_scaled_backward_vs_float64(b, s, cin, cout, hw, t_mode, act, gm, dx, spread):
    from sbmc_amd import _lib
    L = _lib.lib()
    dev = th.device("cuda")
    th.manual_seed(b * 77 + hw)
    slope = 0.01
    x = th.randn(b, cin, hw, device=dev)
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    bias = th.randn(cout, device=dev)
    t = None
    if t_mode == 1:
        t = th.randn(b // s, cout, device=dev)
    elif t_mode == 2:
        t = th.randn(b // s, cout, hw, device=dev)
    gy = th.randn(b, cout, hw, device=dev) * spread
    gmean = th.randn(b // s, cout, hw, device=dev) * spread if gm else None
    # forward on the GPU for the sign bits the backward reads
    y = th.empty(b, cout, hw, device=dev)
    wpr = (hw + 31) // 32
    signs = th.zeros(b, cout, wpr, dtype=th.int32, device=dev)
    _lib.check(L.sbmc_pointwise_fwd_scaled_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(t) if t is not None else None,
                                               _lib.ptr(y), _lib.ptr(signs), None, 1, _lib.ptr(_word(x)), None, b, s, cin, cout,
                                               hw, t_mode, act, slope, _lib.current_stream(dev)), "fwd")
    # float64 of the same layer with the SAME activation decisions (y > 0 as the forward stored them)
    gtot = gy.double()
    if gm:
        gtot = gtot + (gmean.double() / s).repeat_interleave(s, 0)
    if act == 0:
        gz = gtot
    else:
        gz = th.where(y > 0, gtot, gtot * (slope if act == 2 else 0.0))
    gx64 = th.matmul(w.double().t(), gz)
    gw64 = th.einsum("bop,bkp->ok", gz, x.double())
    gb64 = gz.sum((0, 2))
    gt64 = None
    if t_mode == 1:
        gt64 = gz.view(b // s, s, cout, hw).sum((1, 3))
    elif t_mode == 2:
        gt64 = gz.view(b // s, s, cout, hw).sum(1)
    for scaled in (True, False):
        groups = L.sbmc_pointwise_bwd_groups(b, s, 1 if (gm and t_mode == 0) else t_mode, hw)    # (as functions.py)
        nb = b // s if t_mode == 1 else 1
        gx = th.full((b, cin, hw), float("nan"), device=dev) if dx else None
        gwp = th.empty(groups, cout, cin, device=dev)
        gbp = th.empty(groups, nb, cout, device=dev)
        gt = th.empty(b // s, cout, hw, device=dev) if t_mode == 2 else None
        # (the three-plane kernels with a context / mean gradient keep no running maximum: SBMC_HIP_EINVAL if asked)
        want_word = dx and (scaled or not (gm or t_mode == 2))
        gxmax = th.zeros(1, dtype=th.int32, device=dev) if want_word else None
        rc = L.sbmc_pointwise_bwd_scaled_f32(
            _lib.ptr(gy), _lib.ptr(signs) if act != 0 else None, _lib.ptr(x), _lib.ptr(w), _lib.ptr(gx) if dx else None,
            _lib.ptr(gwp), _lib.ptr(gbp), _lib.ptr(gt) if gt is not None else None, _lib.ptr(gmean) if gm else None, s,
            _lib.ptr(_word(gy)) if scaled else None, _lib.ptr(_word(gmean)) if (scaled and gm) else None,
            _lib.ptr(_word(x)) if scaled else None, _lib.ptr(gxmax) if want_word else None,
            b, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev))
        _lib.check(rc, "bwd_scaled")
        what = "scaled" if scaled else "three planes"
        if dx:
            assert (gx.double() - gx64).abs().max().item() <= 1e-5 * gx64.abs().max().item(), what
            if want_word:
                assert gxmax.item() == _word(gx).item(), what
            elif not scaled:
                w0 = th.zeros(1, dtype=th.int32, device=dev)
                assert L.sbmc_pointwise_bwd_scaled_f32(
                    _lib.ptr(gy), _lib.ptr(signs) if act != 0 else None, _lib.ptr(x), _lib.ptr(w), _lib.ptr(gx), _lib.ptr(gwp),
                    _lib.ptr(gbp), _lib.ptr(gt) if gt is not None else None, _lib.ptr(gmean) if gm else None, s, None, None,
                    None, _lib.ptr(w0), b, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev)) == -1
        gw = gwp.double().sum(0)
        assert (gw - gw64).abs().max().item() <= 1e-5 * gw64.abs().max().item(), what
        per_image = gbp.double().sum(0)
        assert (per_image.sum(0) - gb64).abs().max().item() <= 1e-5 * max(gb64.abs().max().item(), 1e-30), what
        if t_mode == 1:
            assert (per_image - gt64).abs().max().item() <= 1e-5 * gt64.abs().max().item(), what
        if t_mode == 2:
            assert (gt.double() - gt64).abs().max().item() <= 1e-5 * gt64.abs().max().item(), what


def test_words_travel_through_a_chain_and_its_backward():
    """modules.pointwise_chain_with_context: every layer behind the first finds its input's word (the two-plane form
    runs), the chain's output carries one, and in the backward every layer finds its gradient's word too."""
    from sbmc_amd import functions as F, modules
    th.manual_seed(3)
    bs, S, cs, cp, h, w = 1, 2, 128, 128, 12, 20
    chain = modules.ConvChain(cs + cp, 128, ksize=1, width=128, depth=3, pad=False).cuda()
    chain.pointwise_as_gemm = True
    per = th.randn(bs, S, cs, h, w, device="cuda").requires_grad_()
    ctx = th.randn(bs, cp, h, w, device="cuda")
    mean_out = []
    out = modules.pointwise_chain_with_context(chain, per, ctx, mean_out)
    assert F.known_amax(out) is not None and F.known_amax(mean_out[0]) is not None
    assert F.known_amax(out).item() == out.detach().abs().max().reshape(1).view(th.int32).item()
    nxt = F.tagged_view(out, bs, S, 128, h, w)
    assert F.known_amax(nxt) is not None
    # a second chain consumes it: its first layer must see the word (x of the saved context)
    chain2 = modules.ConvChain(128 + cp, 25, ksize=1, width=128, depth=3, pad=False, activation="leaky_relu").cuda()
    chain2.pointwise_as_gemm = True
    seen = []
    orig = F.known_amax

    def spy(t):
        r = orig(t)
        seen.append((tuple(t.shape), r is not None))
        return r
    F.known_amax = spy
    try:
        out2 = modules.pointwise_chain_with_context(chain2, nxt, ctx)
        fwd_seen = list(seen)
        del seen[:]
        (out2.sum() + mean_out[0].sum()).backward()
    finally:
        F.known_amax = orig
    # every layer of chain2 found its input's word (round 6: its three layers are ONE fused pass, which looks its input's word up
    # for the backward's sake -- and the context term's, which has none: one absmax pass over it, 1 / S of an activation)
    assert all(ok for shape, ok in fwd_seen if shape[0] == bs * S), fwd_seen
    # backward: the last layer's gradient (from .sum()) has no word; every layer behind it gets one from its successor
    bwd_gy = [ok for shape, ok in seen if len(shape) == 3]
    assert sum(bwd_gy) >= 4, seen
    assert per.grad is not None and th.isfinite(per.grad).all()


@pytest.mark.parametrize("loose", [1.0, 37.0])
@pytest.mark.parametrize("B,cin,cout,hw", [(2, 128, 441, 64 * 37 + 20), (1, 128, 441, 48), (3, 96, 200, 1000), (2, 128, 512, 4096),
                                          (1, 64, 129, 260), (9, 128, 441, 64 * 300)])
def test_wide_layer_whole_backward_vs_float64(B, cin, cout, hw, loose):
    """sbmc_pointwise_wide_bwd_f32 (pw_wide_bwd2_kernel): gx, gw and gbias of the 441-channel layer in one pass over the
    logit gradient, against float64.  loose: the word of gz may be a BOUND (the splat's backward leaves one, tens of
    times the true maximum at worst)."""
    from sbmc_amd import _lib
    L = _lib.lib()
    dev = th.device("cuda")
    th.manual_seed(B * 31 + cout)
    gz = th.randn(B, cout, hw, device=dev) * 1e-3
    gz[:, :, ::7] *= 30.0                              # a wide spread of magnitudes, as logit gradients have
    x = th.randn(B, cin, hw, device=dev).relu_()
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    groups = L.sbmc_pointwise_gw_wide_groups(B, hw)
    gwp = th.full((groups, cout, cin), float("nan"), device=dev)
    gbp = th.full((groups, cout), float("nan"), device=dev)
    gx = th.full((B, cin, hw), float("nan"), device=dev)
    ws = th.empty(L.sbmc_pointwise_wide_bwd_ws_bytes(), dtype=th.uint8, device=dev)
    gxmax = th.zeros(1, dtype=th.int32, device=dev)
    gword = (gz.abs().max() * loose).reshape(1).view(th.int32).clone()
    _lib.check(L.sbmc_pointwise_wide_bwd_f32(_lib.ptr(gz), _lib.ptr(x), _lib.ptr(w), _lib.ptr(gx), _lib.ptr(gwp), _lib.ptr(gbp),
                                             _lib.ptr(ws), _lib.ptr(gword), _lib.ptr(_word(x)), _lib.ptr(gxmax), B, cin, cout, hw,
                                             _lib.current_stream(dev)), "wide_bwd")
    gx64 = th.matmul(w.double().t(), gz.double())
    gw64 = th.einsum("bop,bkp->ok", gz.double(), x.double())
    gb64 = gz.double().sum((0, 2))
    assert (gx.double() - gx64).abs().max().item() <= 1e-5 * gx64.abs().max().item()
    assert (gwp.double().sum(0) - gw64).abs().max().item() <= 1e-5 * gw64.abs().max().item()
    assert (gbp.double().sum(0) - gb64).abs().max().item() <= 1e-5 * gb64.abs().max().item()
    assert gxmax.item() == _word(gx).item()


def test_splat_backward_leaves_a_bound_of_the_logit_gradient():
    """functions.SplatAll.backward tags d_kernels with a word >= max |d_kernels| (from per-pixel quantities, no pass over the
    gradient), and not absurdly loose: within 2^6 of the true maximum on this input."""
    from sbmc_amd import functions as F
    th.manual_seed(9)
    bs, S, c, h, w, k = 1, 3, 3, 30, 150, 21
    data = (th.rand(bs, S, c, h, w) * 4).cuda().requires_grad_()
    kern = (th.randn(bs, S, k * k, h, w) * 2).cuda().requires_grad_()
    sr, sw, mw = F.SplatAll.apply(data, kern)
    grads = th.autograd.grad([sr, sw, mw], [kern], [th.randn_like(sr), th.randn_like(sw), th.randn_like(mw) * 0.1])
    word = F.known_amax(grads[0])
    assert word is not None
    bound = word.view(th.float32).item()
    true = grads[0].abs().max().item()
    assert true <= bound <= 64.0 * true, (true, bound)


def test_regressor_backward_runs_in_one_pass():
    """Multisteps' kernel regressor at the production width: the 441-channel layer's backward is the fused kernel (no
    library GEMM, no second pass), reached through the words of the splat's bound and the forward's magnitude."""
    from sbmc_amd import functions as F, modules
    th.manual_seed(4)
    bs, S, h, w = 1, 2, 24, 40
    chain = modules.ConvChain(256, 441, ksize=1, width=128, depth=3, pad=False, activation="leaky_relu", output_type="linear").cuda()
    chain.pointwise_as_gemm = True
    per = th.randn(bs, S, 128, h, w, device="cuda").requires_grad_()
    ctx = th.randn(bs, 128, h, w, device="cuda")
    # (the per-sample input carries a word, as an embedding's output does)
    F.tag_amax(per, per.detach().abs().max().reshape(1).view(th.int32).clone())
    rad = th.rand(bs, S, 3, h, w, device="cuda")
    calls = []
    F.enable_kernel_timing(calls)
    try:
        kernels = modules.pointwise_chain_with_context(chain, per, ctx)
        kernels = F.tagged_view(kernels, bs, S, 441, h, w)
        sr, sw, _ = F.SplatAll.apply(rad, kernels)
        (sr / (sw + 1e-8)).sum().backward()
    finally:
        F.enable_kernel_timing(None)
    names = [c[0] for c in calls]
    assert any(n.startswith("pointwise_wide_bwd") for n in names), names
    assert not any(n.startswith("pointwise_gw_wide") for n in names), names
    # against float64 of the same graph
    ref = modules.ConvChain(256, 441, ksize=1, width=128, depth=3, pad=False, activation="leaky_relu", output_type="linear").double()
    ref.load_state_dict({k: v.double().cpu() for k, v in chain.state_dict().items()})
    pd = per.detach().double().cpu().requires_grad_()
    xin = th.cat([pd, ctx.double().cpu().unsqueeze(1).expand(bs, S, 128, h, w)], 2).reshape(bs * S, 256, h, w)
    k64 = ref(xin).view(bs, S, 441, h, w)
    from helpers import ProgressiveFP64
    upd = ProgressiveFP64()
    st = (None, None, None)
    for s in range(S):
        st = upd(rad[:, s].double().cpu(), k64[:, s], *st)
    (st[0] / (st[1] + 1e-8)).sum().backward()
    assert (per.grad.double().cpu() - pd.grad).abs().max().item() <= 1e-5 * pd.grad.abs().max().item()
    for (n, a), (_, b) in zip(chain.named_parameters(), ref.named_parameters()):
        assert (a.grad.double().cpu() - b.grad).abs().max().item() <= 1e-5 * b.grad.abs().max().item(), n
