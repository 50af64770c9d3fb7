"""The split-precision 3 x 3 convolution (csrc/conv3x3.hip) against torch's float64 convolution: forward, data
gradient (the same kernel on mirrored weights) and weight gradient, through the C ABI and through the autograd
function the U-nets use.  The fp32 library convolution (MIOpen) is the yardstick: ours must be no further from
float64 than twice its error."""
import os

import pytest
import torch as th
import torch.nn.functional as F

from sbmc_amd import _lib
from sbmc_amd import functions as funcs

pytestmark = pytest.mark.gpu


def _dev():
    if not th.cuda.is_available():
        pytest.skip("needs a GPU")
    return th.device("cuda")


def _cl(t):
    return t.contiguous(memory_format=th.channels_last)


def _err(a, ref):
    return (a.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-300)


def _yardstick(x, w, gy):
    """float64 results and the fp32 library's distance from them (forward, data gradient, weight gradient)."""
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, padding=1)
    gxd, gwd = th.autograd.grad(yd, (xd, wd), gy.double())
    xf, wf = _cl(x).clone().requires_grad_(True), _cl(w).clone().requires_grad_(True)
    yf = F.conv2d(xf, wf, padding=1)
    gxf, gwf = th.autograd.grad(yf, (xf, wf), _cl(gy))
    return (yd.detach(), gxd, gwd), (_err(yf, yd.detach()), _err(gxf, gxd), _err(gwf, gwd))


SHAPES = [
    # b, cin, cout, h, w
    (1, 128, 128, 16, 16),        # exactly one tile
    (1, 128, 128, 37, 53),        # ragged: edge tiles in both directions, strips that end inside a stage
    (2, 128, 256, 24, 40),        # batch of two, two output-channel tiles
    (1, 384, 128, 19, 70),        # the U-net's skip concatenation
    (1, 256, 128, 5, 3),          # smaller than a tile
    (1, 384, 256, 47, 33),        # a last strip of ONE column: shifted right by the tap, nothing of the row is left
    (2, 128, 128, 9, 65),
    (1, 128, 128, 1, 1),
]


@pytest.mark.parametrize("b,cin,cout,h,w", SHAPES)
def test_function_matches_float64(b, cin, cout, h, w):
    dev = _dev()
    g = th.Generator(device="cpu").manual_seed(b * 1000 + cin + h * w)
    x = (th.randn(b, cin, h, w, generator=g) * 2.0).to(dev)
    wt = (th.randn(cout, cin, 3, 3, generator=g) * 0.03).to(dev)
    gy = th.randn(b, cout, h, w, generator=g).to(dev)
    (yd, gxd, gwd), lib_err = _yardstick(x, wt, gy)
    conv = th.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    xs = _cl(x).clone().requires_grad_(True)
    ws = wt.clone().requires_grad_(True)
    if h * w == 1:
        # (a 1 x 1 image is contiguous in both orders: the module path keeps such tensors planar)
        assert not funcs.Conv3x3NHWC.supported(xs, conv)
        return
    assert funcs.Conv3x3NHWC.supported(xs, conv)
    y = funcs.Conv3x3NHWC.apply(xs, ws)
    assert y.is_contiguous(memory_format=th.channels_last)
    gx, gw = th.autograd.grad(y, (xs, ws), _cl(gy))
    for name, got, ref, yard in (("y", y, yd, lib_err[0]), ("gx", gx, gxd, lib_err[1]), ("gw", gw, gwd, lib_err[2])):
        e = _err(got, ref)
        assert e <= max(2.0 * yard, 2e-7), "%s: %.3e of the largest value (fp32 library: %.3e)" % (name, e, yard)


def test_weight_gradient_is_reproducible():
    """The pixel ranges' partial sums are added in a fixed order: two runs agree to the bit."""
    dev = _dev()
    x = _cl(th.randn(1, 128, 45, 77, device=dev))
    wt = th.randn(128, 128, 3, 3, device=dev, requires_grad=True)
    gy = _cl(th.randn(1, 128, 45, 77, device=dev))
    xs = x.clone().requires_grad_(True)
    a = th.autograd.grad(funcs.Conv3x3NHWC.apply(xs, wt), (xs, wt), gy)
    b = th.autograd.grad(funcs.Conv3x3NHWC.apply(xs, wt), (xs, wt), gy)
    assert th.equal(a[0], b[0]) and th.equal(a[1], b[1])


@pytest.mark.parametrize("xs,ws", [(1e-9, 1.0), (3e4, 1e-6), (1.0, 1e3)])
def test_scales_follow_the_tensors(xs, ws):
    """Gradients are ~1e-9, activations can be large: the power-of-two scales come from the tensors themselves."""
    dev = _dev()
    g = th.Generator(device="cpu").manual_seed(5)
    x = (th.randn(1, 128, 20, 33, generator=g) * xs).to(dev)
    wt = (th.randn(128, 128, 3, 3, generator=g) * ws).to(dev)
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    y = funcs.Conv3x3NHWC.apply(_cl(x), wt)
    lib = F.conv2d(_cl(x), _cl(wt), padding=1)
    assert _err(y, ref) <= max(2.0 * _err(lib, ref), 2e-7)


def test_wide_dynamic_range_inside_a_tensor():
    """Values 2^-17 and more below the tensor's largest keep an ABSOLUTE accuracy of 2^-39 of that largest value
    (csrc/conv3x3.hip): a region of tiny values next to large ones stays within 1e-5 of ITS OWN scale down to
    ~1e-6 of the tensor's maximum."""
    dev = _dev()
    g = th.Generator(device="cpu").manual_seed(9)
    x = th.randn(1, 128, 32, 64, generator=g)
    x[..., 32:] *= 1e-5                       # the right half: five decades below
    x = x.to(dev)
    wt = (th.randn(128, 128, 3, 3, generator=g) * 0.05).to(dev)
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    y = funcs.Conv3x3NHWC.apply(_cl(x), wt)
    far = (slice(None), slice(None), slice(None), slice(40, None))        # pixels that see only the small half
    assert _err(y[far], ref[far]) <= 1e-5
    assert _err(y, ref) <= 2e-6


def test_zero_input_gives_zeros():
    dev = _dev()
    x = _cl(th.zeros(1, 128, 16, 20, device=dev))
    wt = th.randn(128, 128, 3, 3, device=dev)
    assert funcs.Conv3x3NHWC.apply(x, wt).abs().max().item() == 0.0


def test_abi_rejects_unsupported_shapes():
    _dev()
    L = _lib.lib()
    assert L.sbmc_conv3x3_supported(1, 8, 8, 128, 128) == 1
    assert L.sbmc_conv3x3_supported(1, 8, 8, 48, 128) == 0          # input channels: multiples of 32
    assert L.sbmc_conv3x3_supported(1, 8, 8, 128, 64) == 0          # output channels: multiples of 128
    assert L.sbmc_conv3x3_weights_bytes(48, 128) == 0
    assert L.sbmc_conv3x3_wgrad_supported(1, 8, 8, 64, 128) == 0
    assert L.sbmc_conv3x3_wgrad_supported(1, 8, 8, 256, 128) == 1
    assert L.sbmc_conv3x3_nhwc_f32(None, None, None, None, 1, 8, 8, 128, 128, None, None) == -1


@pytest.mark.parametrize("activation", ["tanh", "leaky_relu"])
def test_unet_is_the_same_function_with_and_without_the_kernel(monkeypatch, activation):
    """One U-net (the model's own module) forward + backward, every convolution on MIOpen vs on the kernel: both
    against float64.  With a smooth activation every tensor must be as close to float64 as the library's (2x).
    With the model's leaky ReLU the gradients of ANY fp32 implementation jump where a pre-activation within
    rounding of zero changes sign (tools' chain experiment: 1e-3 for either implementation at larger sizes), so
    only the output and a loose bound on the gradients are compared there."""
    from sbmc_amd import modules as ops
    dev = _dev()
    th.manual_seed(3)
    net = ops.Autoencoder(128, 128, num_levels=3, increase_factor=2.0, num_convs=3, width=128, ksize=3,
                          output_type="linear", activation=activation, pooling="max").to(dev)
    for m in net.modules():
        if isinstance(m, ops.ConvChain):
            m.fuse_bias_act = True
    x = th.randn(1, 128, 48, 80, device=dev)
    gy = th.randn(1, 128, 48, 80, device=dev)
    used = []
    real = funcs.Conv3x3NHWC._prepare
    monkeypatch.setattr(funcs.Conv3x3NHWC, "_prepare", staticmethod(lambda *a: (used.append(1), real(*a))[1]))

    def run(flag, dtype=th.float32):
        monkeypatch.setenv("SBMC_CONV3X3", flag)
        n = net if dtype == th.float32 else net.double()
        xi = x.to(dtype).requires_grad_(True)
        ps = [p for p in n.parameters()]
        y = n(xi)
        gs = th.autograd.grad(y, [xi] + ps, gy.to(dtype))
        out = [y.detach().double()] + [g_.double() for g_ in gs]
        if dtype != th.float32:
            n.float()
        return out

    ours = run("1")
    assert len(used) == 30, len(used)                   # weights prepared for 15 convolutions forward, 15 data gradients
    lib = run("0")
    assert len(used) == 30
    ref = run("0", th.float64)
    for i, (a, b, r) in enumerate(zip(ours, lib, ref)):
        ea, eb = _err(a, r), _err(b, r)
        if activation == "tanh" or i == 0:
            assert ea <= max(2.0 * eb, 2e-6), (i, ea, eb)
        else:
            assert ea <= max(2.0 * eb, 5e-5), (i, ea, eb)


def test_amax_tags_replace_the_absmax_pass_and_expire(monkeypatch):
    """The bias / activation passes leave max |.| of what they write on the tensor object; the convolution that
    reads it next must use it (no absmax pass of activations at all) -- forward AND backward, where the tag has to
    survive the autograd engine -- and a tag must die with any in-place change."""
    from sbmc_amd import modules as ops
    dev = _dev()
    th.manual_seed(4)
    net = ops.Autoencoder(128, 128, num_levels=3, increase_factor=2.0, num_convs=3, width=128, ksize=3,
                          output_type="leaky_relu", pooling="max").to(dev)
    for m in net.modules():
        if isinstance(m, ops.ConvChain):
            m.fuse_bias_act = True
    calls = []
    real = funcs.Conv3x3NHWC._absmax
    monkeypatch.setattr(funcs.Conv3x3NHWC, "_absmax", staticmethod(lambda t: (calls.append(tuple(t.shape)), real(t))[1]))
    x = th.randn(1, 128, 32, 48, device=dev, requires_grad=True)
    y = net(x)
    n_fwd = len(calls)
    y.backward(th.randn_like(y))
    n_bwd = len(calls) - n_fwd
    assert n_fwd == 0, calls[:n_fwd]                    # (the U-net's input: from its entry transpose)
    assert n_bwd == 0, calls[n_fwd:]
    # tags are bounds that hold: compare with a run that ignores them
    monkeypatch.setenv("SBMC_AMAX_TAGS", "0")
    y2 = net(x)
    assert _err(y2, y.detach().double()) < 2e-6
    monkeypatch.delenv("SBMC_AMAX_TAGS")
    # expiry
    t = _cl(th.randn(1, 128, 8, 8, device=dev))
    funcs.tag_amax(t, real(t))
    assert funcs.known_amax(t) is not None
    t.mul_(2.0)
    assert funcs.known_amax(t) is None


@pytest.mark.parametrize("act,slope", [(0, 0.0), (1, 0.0), (2, 0.01)])
def test_fused_epilogue_equals_the_two_passes(act, slope):
    """Bias + activation (+ sign bits, + largest magnitude) in the convolution's epilogue: the same arithmetic in the
    same order as the convolution followed by the bias / activation pass -- equal to the bit, forward and backward."""
    dev = _dev()
    g = th.Generator(device="cpu").manual_seed(11 + act)
    x = _cl((th.randn(2, 128, 21, 37, generator=g)).to(dev))
    wt = (th.randn(256, 128, 3, 3, generator=g) * 0.03).to(dev)
    bias = th.randn(256, generator=g).to(dev)
    gy = _cl(th.randn(2, 256, 21, 37, generator=g).to(dev))

    def two_passes():
        xs, ws, bs = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        y = funcs.BiasActNHWC.apply(funcs.Conv3x3NHWC.apply(xs, ws), bs, act, slope)
        return (y.detach().clone(),) + th.autograd.grad(y, (xs, ws, bs), gy)

    def fused():
        xs, ws, bs = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        y, amax = funcs.Conv3x3BiasActNHWC.apply(xs, ws, bs, act, slope)
        assert amax.view(th.float32).item() == y.abs().max().item()
        return (y.detach().clone(),) + th.autograd.grad(y, (xs, ws, bs), gy)

    for i, (a, b) in enumerate(zip(two_passes(), fused())):
        if i == 3:
            # the bias gradient: the same per-workgroup partial sums, added up by torch's `sum` in the two-pass form and
            # by the weight gradient's reduction launch in the fused one (round 4) -- two fixed orders of one fp32 sum
            assert (a - b).abs().max().item() <= 2e-6 * a.abs().max().item()
        else:
            assert th.equal(a, b)


@pytest.mark.parametrize("shape", [(1, 128, 48, 80), (2, 64, 31, 44), (1, 128, 720, 1280)])
def test_every_pass_that_leaves_a_scale_leaves_the_largest_magnitude(shape):
    """A word that is too small would overflow the f16 planes of the convolution that trusts it."""
    dev = _dev()
    x = th.randn(*shape, device=dev) * 3.0
    out, amax = funcs.ToChannelsLast.apply(x, True)
    assert th.equal(out, _cl(x)) and amax.view(th.float32).item() == x.abs().max().item()
    bias = th.randn(shape[1], device=dev)
    for act, slope in ((0, 0.0), (2, 0.01)):
        y = _cl(x).clone().requires_grad_(True)
        z, am = funcs.BiasActNHWC.apply(y * 1.0, bias, act, slope, True)
        assert am.view(th.float32).item() == z.abs().max().item()
        g = _cl(th.randn(*shape, device=dev))
        (gz,) = th.autograd.grad(z, y, g)
    # the adjoint's word rides on the gradient it returns (checked where it is consumed: the tag test above)


@pytest.mark.parametrize("shape", [(1, 128, 128, 94, 1280), (1, 256, 256, 49, 640), (1, 512, 512, 26, 320), (2, 768, 256, 33, 70),
                                   (1, 128, 128, 16, 16), (1, 512, 128, 5, 7)])
def test_stream_k_equals_whole_tiles(shape, monkeypatch):
    """Stream-K (round 4): a launch whose tile count is no multiple of the CU count cuts its (tile, chunk) units into equal
    ranges over all compute units; split tiles are completed by the fix-up launch.  Same values as the whole-tile walk
    up to the order of ONE addition per split tile, the same to the bit from run to run; shapes: the slabs of a rank of
    8 at the three U-net levels (480 / 320 / 160 tiles), more chunks than tiles, a single tile over 4 and 16 chunks."""
    dev = _dev()
    b, cin, cout, h, w = shape
    g = th.Generator(device="cpu").manual_seed(5 + cin + h)
    x = _cl(th.randn(b, cin, h, w, generator=g).to(dev))
    wt = (th.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    bias = th.randn(cout, generator=g).to(dev)
    ys = {}
    for mode in ("1", "0", "1"):
        monkeypatch.setenv("SBMC_CONV3X3_STREAMK", mode)
        y, amax = funcs.Conv3x3BiasActNHWC.apply(x, wt, bias, 2, 0.01)
        assert amax.view(th.float32).item() == y.abs().max().item()
        if mode in ys:
            assert th.equal(ys[mode], y)                    # reproducible to the bit
        ys[mode] = y.clone()
    scale = ys["0"].abs().max().item()
    assert (ys["1"] - ys["0"]).abs().max().item() <= 2e-6 * scale
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), bias.double(), padding=1), 0.01)
    assert _err(ys["1"], ref) <= 1e-5


# ---- ABI 7: the data gradient with the producing layer's activation adjoint in its epilogue (csrc ADJ) ----

def _two_layers(x, w1, b1, w2, b2, gy, act, slope, chained):
    """y2 = act(conv(act(conv(x, w1) + b1), w2) + b2) through Conv3x3BiasActNHWC, with (chained) or without the link between
    the layers; returns y2 and every gradient."""
    leaves = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    xs, w1s, b1s, w2s, b2s = leaves
    y1, a1 = funcs.Conv3x3BiasActNHWC.apply(xs, w1s, b1s, act, slope, None, chained)
    funcs.tag_amax(y1, a1)
    link = funcs.Conv3x3BiasActNHWC.adj_link_for(y1)
    assert (link is not None) == chained
    y2, _ = funcs.Conv3x3BiasActNHWC.apply(y1, w2s, b2s, act, slope, link, False)
    return (y2.detach().clone(),) + th.autograd.grad(y2, leaves, gy) + (y1.detach().clone(),)


@pytest.mark.parametrize("b,c0,c1,c2,h,w", [
    (1, 128, 128, 128, 37, 53),       # ragged: the general epilogue on the right edge, rows beyond the image below
    (1, 128, 256, 128, 48, 64),       # lean epilogue everywhere, two channel tiles of the adjoint's output
    (2, 384, 128, 256, 24, 40),
    (1, 256, 512, 512, 26, 320),      # a stream-K launch (160 tiles): the fix-up kernel's epilogue and its rows of sums
    (1, 128, 128, 128, 94, 1280),     # 480 tiles: two rounds of whole tiles
    (1, 128, 128, 128, 1, 1),
])
@pytest.mark.parametrize("act,slope", [(2, 0.01), (1, 0.0)])
def test_chained_layers_equal_the_unchained_and_float64(b, c0, c1, c2, h, w, act, slope):
    """The second layer's data gradient applies the first one's activation adjoint, sums its bias gradient and leaves the
    gradient's magnitude word: same values as the pass of its own (one product per element in either form; the bias sums
    in another fixed order), and both against float64."""
    dev = _dev()
    g = th.Generator(device="cpu").manual_seed(c0 + c1 + h * w + act)
    x = _cl((th.randn(b, c0, h, w, generator=g) * 2.0).to(dev))
    w1 = (th.randn(c1, c0, 3, 3, generator=g) * (2.0 / (9 * c0)) ** 0.5).to(dev)
    w2 = (th.randn(c2, c1, 3, 3, generator=g) * (2.0 / (9 * c1)) ** 0.5).to(dev)
    b1, b2 = th.randn(c1, generator=g).to(dev) * 0.5, th.randn(c2, generator=g).to(dev) * 0.5
    gy = _cl(th.randn(b, c2, h, w, generator=g).to(dev))
    launches = []
    funcs.enable_kernel_timing(launches)
    try:
        plain = _two_layers(x, w1, b1, w2, b2, gy, act, slope, False)
        n_plain = sum(n.startswith("conv3x3_bwd_data_adj") for n, _, _ in launches)
        chained = _two_layers(x, w1, b1, w2, b2, gy, act, slope, True)
        n_chained = sum(n.startswith("conv3x3_bwd_data_adj") for n, _, _ in launches)
        again = _two_layers(x, w1, b1, w2, b2, gy, act, slope, True)
    finally:
        funcs.enable_kernel_timing(None)
    assert n_plain == 0 and n_chained == 1                       # the fused epilogue ran, once
    for a, c in zip(chained, again):
        assert th.equal(a, c)                                     # the same to the bit from run to run
    names = ("y", "gx", "gw1", "gb1", "gw2", "gb2")
    for name, a, c in zip(names, plain, chained):
        scale = a.abs().max().item()
        # gz1 is the same product in both forms; what follows it differs by fp32 rounding of other sums' order only where
        # a sum's order differs: gb1 (partial sums per workgroup instead of per chunk of pixels)
        tol = 1e-5 if name == "gb1" else 0.0
        assert (a - c).abs().max().item() <= tol * scale, name
    # float64 with the activations' branches as the kernels took them: a pre-activation within fp32 rounding of zero (one or
    # two of a million here) may fall on the other side in float64 -- the function's own discontinuity, worth a whole
    # element of the gradient, and nothing the comparison is about
    xd, w1d, b1d, w2d, b2d = (t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2))
    m1 = th.where(chained[6] > 0, 1.0, slope).double()
    m2 = th.where(chained[0] > 0, 1.0, slope).double()
    yd = (F.conv2d(F.conv2d(xd, w1d, b1d, padding=1) * m1, w2d, b2d, padding=1)) * m2
    ref = (yd.detach(),) + th.autograd.grad(yd, (xd, w1d, b1d, w2d, b2d), gy.double())
    for name, a, r in zip(names, chained, ref):
        assert _err(a, r) <= 1e-5, name


def test_adj_entry_point_through_the_abi_and_its_words():
    """sbmc_conv3x3_adj_nhwc_f32 by itself: gz against float64, the bias sums, the magnitude word; argument errors."""
    dev = _dev()
    L = _lib.lib()
    b, cin, cout, h, w = 1, 256, 128, 20, 48          # gy has cin channels; the producing layer cout
    g = th.Generator(device="cpu").manual_seed(77)
    gy = _cl(th.randn(b, cin, h, w, generator=g).to(dev))
    wt = (th.randn(cin, cout, 3, 3, generator=g) * 0.02).to(dev)          # the consumer's weight [its cout = cin here][cout]
    z = _cl(th.randn(b, cout, h, w, generator=g).to(dev))                 # the producer's pre-activation
    bits = (z.permute(0, 2, 3, 1).reshape(-1, 32) > 0).to(th.int64)
    words = (bits << th.arange(32, device=dev)).sum(1)
    signs = th.where(words >= 2 ** 31, words - 2 ** 32, words).to(th.int32).contiguous()
    gmax = funcs.Conv3x3NHWC._absmax(gy)
    wp = funcs.Conv3x3NHWC._prepare(wt, True)
    gz = th.empty((b, cout, h, w), device=dev).contiguous(memory_format=th.channels_last)
    rows = L.sbmc_conv3x3_adj_partial_rows()
    partial = th.zeros(rows, cout, device=dev)
    amax = th.zeros(1, dtype=th.int32, device=dev)
    st = _lib.current_stream(dev)
    args = lambda signs_ptr, co: (_lib.ptr(gy), _lib.ptr(gmax), _lib.ptr(wp), signs_ptr, 0.01, _lib.ptr(gz), _lib.ptr(partial),
                                  _lib.ptr(amax), b, h, w, cin, co, None, st)
    assert L.sbmc_conv3x3_adj_supported(b, h, w, cin, cout) == 1 and L.sbmc_conv3x3_adj_supported(b, h, w, cin, 1024) == 0
    assert L.sbmc_conv3x3_adj_nhwc_f32(*args(None, cout)) == -1
    assert L.sbmc_conv3x3_adj_nhwc_f32(*args(_lib.ptr(signs), cout)) == 0
    gxd = th.nn.grad.conv2d_input((b, cout, h, w), wt.double(), gy.double(), padding=1)
    ref = gxd * th.where(z > 0, 1.0, 0.01).double()
    assert _err(gz, ref) <= 1e-5
    assert _err(partial.sum(0), ref.sum((0, 2, 3))) <= 1e-5
    assert amax.view(th.float32).item() == gz.abs().max().item()


def test_unet_chains_with_and_without_the_fused_adjoint(monkeypatch):
    """One U-net of the model (5 chains of 3 convolutions): with the links (the default) the second and third convolution
    of every chain apply the adjoint of the one before them -- 10 data gradients in the ADJ form, 10 `bias_act_nhwc_bwd` passes
    less -- and every gradient is what the passes of their own give: to the bit, except the 10 bias gradients whose partial
    sums are added up in another order."""
    from sbmc_amd import modules as ops
    dev = _dev()
    th.manual_seed(5)
    net = ops.Autoencoder(128, 128, num_levels=3, increase_factor=2.0, num_convs=3, width=128, ksize=3,
                          output_type="leaky_relu", pooling="max").to(dev)
    for m in net.modules():
        if isinstance(m, ops.ConvChain):
            m.fuse_bias_act = True
    x = th.randn(1, 128, 48, 80, device=dev)
    gy = th.randn(1, 128, 48, 80, device=dev)
    names = [n for n, _ in net.named_parameters()]

    def run(flag):
        monkeypatch.setenv("SBMC_CONV3X3_ADJ", flag)
        launches = []
        funcs.enable_kernel_timing(launches)
        try:
            xi = x.clone().requires_grad_(True)
            y = net(xi)
            gs = th.autograd.grad(y, [xi] + list(net.parameters()), gy)
        finally:
            funcs.enable_kernel_timing(None)
        return [y.detach()] + list(gs), sum(n.startswith("conv3x3_bwd_data_adj") for n, _, _ in launches)

    with_links, n1 = run("1")
    without, n0 = run("0")
    assert (n1, n0) == (10, 0), (n1, n0)
    differing = 0
    for name, a, b in zip(["y", "gx"] + names, with_links, without):
        if th.equal(a, b):
            continue
        assert name.endswith("bias"), name
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item(), name
        differing += 1
    assert differing <= 14            # (+ two pooling + skip nodes and two upsamplings that apply the adjoint of the chain before them)


@pytest.mark.parametrize("b,c0,c1,h,w", [(1, 128, 128, 36, 52), (2, 128, 256, 16, 32), (1, 256, 512, 6, 10)])
def test_pool_skip_applies_the_adjoint_of_the_chain_before_it(b, c0, c1, h, w):
    """A U-net level's pooling + skip node behind a chain's last convolution: with the link its backward pass is also
    that layer's `bias_act_nhwc_bwd` pass (sbmc_maxpool2_nhwc_bwd_add_adj_f32) -- same gradients to the bit, the bias gradient
    up to the order of its partial sums."""
    dev = _dev()
    g = th.Generator(device="cpu").manual_seed(b + c1 + h)
    x = _cl(th.randn(b, c0, h, w, generator=g).to(dev))
    wt = (th.randn(c1, c0, 3, 3, generator=g) * (2.0 / (9 * c0)) ** 0.5).to(dev)
    bias = (th.randn(c1, generator=g) * 0.3).to(dev)
    gp = _cl(th.randn(b, c1, h // 2, w // 2, generator=g).to(dev))
    gs = _cl(th.randn(b, c1, h, w, generator=g).to(dev))

    def run(linked):
        leaves = [t.clone().requires_grad_(True) for t in (x, wt, bias)]
        y, amax = funcs.Conv3x3BiasActNHWC.apply(leaves[0], leaves[1], leaves[2], 2, 0.01, None, linked)
        funcs.tag_amax(y, amax)
        link = funcs.Conv3x3BiasActNHWC.adj_link_for(y)
        assert (link is not None) == linked
        pooled, skip = funcs.PoolSkip.apply(y, link)
        out = th.autograd.grad((pooled, skip), leaves, (gp, gs))
        assert link is None or (link.taken and link.done is None)
        return (pooled.detach().clone(),) + out

    plain, linked = run(False), run(True)
    for name, a, c in zip(("pooled", "gx", "gw", "gb"), plain, linked):
        if name == "gb":
            assert (a - c).abs().max().item() <= 1e-5 * a.abs().max().item()
        else:
            assert th.equal(a, c), name


@pytest.mark.parametrize("b,c0,cu,cl,h,w", [(1, 128, 256, 128, 18, 26), (2, 256, 512, 256, 6, 10), (1, 128, 128, 128, 1, 1)])
def test_upsampling_applies_the_adjoint_of_the_chain_before_it(b, c0, cu, cl, h, w):
    """The bilinear x2 + concatenation behind a coarser level's last convolution: with the link the coarse map's gradient
    comes out of the upsampling's backward pass with that layer's activation adjoint applied
    (sbmc_upsample2x_cat_nhwc_bwd_adj_f32) -- same gradients to the bit, the bias gradient up to its partial sums' order."""
    dev = _dev()
    g = th.Generator(device="cpu").manual_seed(b + cu + h)
    x = _cl(th.randn(b, c0, h, w, generator=g).to(dev))
    wt = (th.randn(cu, c0, 3, 3, generator=g) * (2.0 / (9 * c0)) ** 0.5).to(dev)
    bias = (th.randn(cu, generator=g) * 0.3).to(dev)
    left = _cl(th.randn(b, cl, 2 * h, 2 * w, generator=g).to(dev))
    gout = _cl(th.randn(b, cu + cl, 2 * h, 2 * w, generator=g).to(dev))

    def run(linked):
        leaves = [t.clone().requires_grad_(True) for t in (x, wt, bias, left)]
        y, amax = funcs.Conv3x3BiasActNHWC.apply(leaves[0], leaves[1], leaves[2], 2, 0.01, None, linked)
        funcs.tag_amax(y, amax)
        link = funcs.Conv3x3BiasActNHWC.adj_link_for(y)
        cat = funcs.UpsampleCatNHWC.apply(y, leaves[3], 0, 0, link)
        out = th.autograd.grad(cat, leaves, gout)
        assert link is None or (link.taken and link.done is None)
        return (cat.detach().clone(),) + out

    plain, linked = run(False), run(True)
    for name, a, c in zip(("cat", "gx", "gw", "gb", "gleft"), plain, linked):
        if name == "gb":
            assert (a - c).abs().max().item() <= 1e-5 * a.abs().max().item()
        else:
            assert th.equal(a, c), name
