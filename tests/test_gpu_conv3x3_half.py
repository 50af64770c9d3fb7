"""The U-nets' 3 x 3 convolutions on HALF activations (csrc/conv3x3.hip in its one-plane form,
functions.Conv3x3BiasActHalfNHWC; "fp16 activations", BASELINE configs[4]).  Not an operator of the reference's own
(there: cuDNN behind nn.Conv2d under autocast, sbmc/modules.py:154-175), so the checker is torch's float64 convolution
of the SAME half-rounded inputs and half-rounded weights -- torch.autocast(float16) semantics: exact products of half
values, fp32 sums, ONE rounding to half on the way out (2^-11 relative)."""
import pytest
import torch as th
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
HALF_ULP = 2.0 ** -11


def _cl(t):
    return t.contiguous(memory_format=th.channels_last)


def _check_half(a, ref, what, extra=0.0):
    """|a - ref| <= one half rounding of ref (+ fp32 summation noise relative to the tensor's scale)"""
    a, ref = a.double(), ref.double()
    bound = HALF_ULP * 1.01 * ref.abs() + (1e-5 + extra) * ref.abs().max()
    bad = (a - ref).abs() > bound
    assert not bad.any(), "%s: %d / %d beyond one half rounding, worst %.3e of scale" % (
        what, int(bad.sum()), ref.numel(), ((a - ref).abs().max() / ref.abs().max()).item())


@pytest.mark.parametrize("shape", [(1, 128, 128, 32, 48), (2, 128, 256, 21, 37), (1, 384, 128, 17, 16), (1, 512, 512, 9, 20),
                                   (1, 128, 128, 1, 5)])
@pytest.mark.parametrize("act,slope", [(0, 0.0), (2, 0.01)])
def test_half_convolution_forward_and_gradients(shape, act, slope):
    from sbmc_amd import functions as funcs
    b, cin, cout, h, w = shape
    g = th.Generator().manual_seed(7 + cin + h)
    x = _cl(th.randn(b, cin, h, w, generator=g).to(DEV).half()).requires_grad_()
    wt = (th.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(DEV).requires_grad_()
    bias = th.randn(cout, generator=g).to(DEV).requires_grad_()
    gy = _cl(th.randn(b, cout, h, w, generator=g).to(DEV).half())
    y = funcs.Conv3x3BiasActHalfNHWC.apply(x, wt, bias, act, slope)
    assert y.dtype == th.float16 and y.is_contiguous(memory_format=th.channels_last)
    gx, gw, gb = th.autograd.grad(y, (x, wt, bias), gy)
    assert gx.dtype == th.float16 and gw.dtype == th.float32 and gb.dtype == th.float32
    # float64 of the same half values
    xd = x.detach().double().requires_grad_()
    wd = wt.detach().half().double().requires_grad_()           # the weight rounded to half once
    bd = bias.detach().double().requires_grad_()
    pre = F.conv2d(xd, wd, bd, padding=1)
    yd = pre if act == 0 else F.leaky_relu(pre, slope)
    _check_half(y, yd, "y")
    # the adjoint: gz = gy * act'(pre), rounded to half once (what the activation's backward hands on under autocast)
    gz = gy.double() if act == 0 else th.where(pre > 0, gy.double(), gy.double() * slope)
    gz = gz.half().double()
    gxd, gwd = th.autograd.grad(pre, (xd, wd), gz)
    # (a pre-activation within fp32 rounding of zero may take the other branch: a handful of elements at most)
    extra = 0.0 if act == 0 else 2e-3
    _check_half(gx, gxd, "gx", extra)
    e = (gw.double() - gwd).abs().max().item() / gwd.abs().max().item()
    assert e <= 1e-5 + extra, ("gw", e)
    e = (gb.double() - gz.sum((0, 2, 3))).abs().max().item() / gz.sum((0, 2, 3)).abs().max().item()
    assert e <= 1e-5 + extra, ("gbias", e)


def test_half_convolution_against_the_fp32_kernels_on_the_same_values():
    """The one-plane form against the three-product fp32 form (functions.Conv3x3BiasActNHWC) fed the same half-rounded
    values: they differ by the output's rounding to half only."""
    from sbmc_amd import functions as funcs
    g = th.Generator().manual_seed(3)
    x = _cl(th.randn(1, 256, 40, 56, generator=g).to(DEV).half())
    wt = (th.randn(256, 256, 3, 3, generator=g) * 0.02).to(DEV).half().float()
    bias = th.randn(256, generator=g).to(DEV)
    yh = funcs.Conv3x3BiasActHalfNHWC.apply(x, wt, bias, 2, 0.01)
    yf, _ = funcs.Conv3x3BiasActNHWC.apply(_cl(x.float()), wt, bias, 2, 0.01)
    _check_half(yh, yf, "half vs fp32 kernels")


def test_unet_under_autocast_runs_on_the_half_kernels(monkeypatch):
    """An Autoencoder of the model's shape under torch.autocast(float16): every 3 x 3 convolution goes through the
    half kernels (no MIOpen convolution is called), forward within half accuracy of the fp32 network, all parameter
    gradients finite and within 2 % of the fp32 network's (half activations: ~1e-3 per layer)."""
    from sbmc_amd import modules as ops, functions as funcs
    th.manual_seed(0)
    net = ops.Autoencoder(128, 128, num_levels=3, increase_factor=2.0, num_convs=3, width=128, ksize=3,
                          output_type="leaky_relu", pooling="max").to(DEV)
    for m in net.modules():
        if isinstance(m, ops.ConvChain):
            m.fuse_bias_act = True
    x = th.randn(1, 128, 48, 64, device=DEV)
    calls = {"n": 0}
    real = funcs.Conv3x3BiasActHalfNHWC.forward

    def counted(ctx, *a):
        calls["n"] += 1
        return real(ctx, *a)
    monkeypatch.setattr(funcs.Conv3x3BiasActHalfNHWC, "forward", staticmethod(counted))
    monkeypatch.setattr(F, "conv2d", lambda *a, **k: (_ for _ in ()).throw(AssertionError("MIOpen convolution called")))
    with th.autocast("cuda", dtype=th.float16):
        yh = net(x.half())
    assert calls["n"] == 15 and yh.dtype == th.float16
    # (a loss scale, as torch's GradScaler applies under float16 autocast: the mean's 1 / N would push the half
    # gradients of the activations below the half range)
    (yh.float().square().mean() * 65536.0).backward()
    gh = {k: p.grad.clone() / 65536.0 for k, p in net.named_parameters()}
    monkeypatch.undo()
    net.zero_grad()
    yf = net(x)
    yf.square().mean().backward()
    e = ((yh.float() - yf).abs().max() / yf.abs().max()).item()
    assert e < 2e-2, e
    for k, p in net.named_parameters():
        assert th.isfinite(gh[k]).all(), k
    num = sum(((gh[k] - p.grad) ** 2).sum().item() for k, p in net.named_parameters())
    den = sum((p.grad ** 2).sum().item() for k, p in net.named_parameters())
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


@pytest.mark.parametrize("shape", [(1, 128, 36, 52), (2, 8, 12, 20), (1, 132, 7, 8)])
def test_half_layout_changes_are_exact(shape):
    """planar <-> channels-last of half tensors by the LDS-tile transpose (csrc/nhwc_ops.hip transpose2d_h_kernel):
    a permutation, equal to torch's copy bit for bit, forward and backward."""
    from sbmc_amd import functions as funcs
    x = th.randn(*shape, device=DEV).half().requires_grad_()
    assert funcs.ToChannelsLast.supported(x)
    y = funcs.ToChannelsLast.apply(x)
    assert y.is_contiguous(memory_format=th.channels_last) and th.equal(y, x.detach())
    z = funcs.FromChannelsLast.apply(y)
    assert z.is_contiguous() and th.equal(z, x.detach())
    g = th.randn(*shape, device=DEV).half()
    z.backward(g)
    assert th.equal(x.grad, g)
