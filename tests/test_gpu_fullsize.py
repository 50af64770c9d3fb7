"""GPU, BASELINE.json sizes (1280x720, k=21): properties that do not need the CPU oracle at
full size, plus a full-width band checked against the oracle."""
import pytest
import torch as th

from helpers import close_sum, close, run_progressive

pytestmark = pytest.mark.gpu

H, W, K = 720, 1280, 21


def _inputs(spp, seed, h=H, w=W):
    g = th.Generator().manual_seed(seed)
    rad = [th.empty(1, 3, h, w).exponential_(1.0, generator=g) for _ in range(spp)]
    ker = [th.randn(1, K * K, h, w, generator=g) for _ in range(spp)]
    return rad, ker


def test_full_width_band_against_oracle(oracle):
    """Full 1280-px rows (every strip of a row: both borders and the interior fast path),
    48 rows, 2 spp -- the largest case the oracle finishes in seconds."""
    from sbmc_amd import modules
    rad, ker = _inputs(2, 21, h=48)
    grads = [th.randn(1, 3, 48, W), th.zeros(1, 1, 48, W), th.zeros(1, 1, 48, W)]
    ref_out, ref_dd, ref_dk = run_progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=True),
        rad, ker, grads, "cpu")
    out, dd, dk = run_progressive(modules.ProgressiveKernelApply(splat=True), rad, ker, grads, "cuda")
    for a, b in zip(out, ref_out):
        close(a, b)
    for s in range(2):
        close(dd[s], ref_dd[s]); close(dk[s], ref_dk[s])


def test_fullsize_fused_equals_composed_ops():
    """At 1280x720 the fused kernels agree with the composition of the boundary-level HIP
    operators (Scatter2Gather + KernelWeighting + torch), forward and backward."""
    from sbmc_amd import modules
    rad, ker = _inputs(2, 22)
    grads = [th.randn(1, 3, H, W), th.randn(1, 1, H, W), th.randn(1, 1, H, W)]
    a_out, a_dd, a_dk = run_progressive(modules.ProgressiveKernelApply(splat=True, fused=True), rad, ker, grads, "cuda")
    b_out, b_dd, b_dk = run_progressive(modules.ProgressiveKernelApply(splat=True, fused=False), rad, ker, grads, "cuda")
    for a, b in zip(a_out, b_out):
        close(a, b)
    for s in range(2):
        close(a_dd[s], b_dd[s]); close(a_dk[s], b_dk[s])


def test_fullsize_properties():
    from sbmc_amd import modules
    upd = modules.ProgressiveKernelApply(splat=True)
    rad, ker = _inputs(3, 23)
    rad = [r.cuda() for r in rad]
    ker = [k.cuda() for k in ker]

    def run(order, shift=0.0, const=None):
        sr = sw = mw = None
        for i in order:
            r = rad[i] if const is None else th.full_like(rad[i], const)
            sr, sw, mw = upd(r, ker[i] + shift, sr, sw, mw)
        return sr, sw, mw

    sr, sw, mw = run([0, 1, 2])
    out = sr / (sw + 1e-8)
    # (1) the normalised result does not depend on the order in which samples are splatted
    sr2, sw2, mw2 = run([2, 0, 1])
    close(sr2 / (sw2 + 1e-8), out)
    close(mw2, mw, rtol=0)
    # (2) softmax shift invariance: adding a constant to every logit moves max_w, nothing else
    #     (interior only: out-of-image taps carry logit 0 and do not shift)
    p = (K - 1) // 2
    sr3, sw3, mw3 = run([0, 1, 2], shift=1.5)
    close((sr3 / (sw3 + 1e-8))[..., p:-p, p:-p], out[..., p:-p, p:-p])
    close(mw3[..., p:-p, p:-p], mw[..., p:-p, p:-p] + 1.5, rtol=1e-6)
    # (3) partition of unity: constant radiance c is reproduced exactly in the interior
    src, swc, _ = run([0, 1, 2], const=2.5)
    close((src / (swc + 1e-8))[..., p:-p, p:-p], th.full((1, 3, H - 2 * p, W - 2 * p), 2.5))
    # (4) sum_w >= 1 everywhere (the arg-max tap contributes exp(0)), and everything is finite
    assert (sw >= 1.0 - 1e-6).all() and th.isfinite(sr).all() and th.isfinite(sw).all()


def test_fullsize_scatter2gather_involution_and_kw_linearity():
    from sbmc_amd import functions as F
    g = th.Generator().manual_seed(24)
    x = th.randn(1, K, K, H, W, generator=g).cuda()
    y = F.Scatter2Gather.apply(x)
    mask = F.Scatter2Gather.apply(F.Scatter2Gather.apply(th.ones_like(x)))
    assert th.equal(F.Scatter2Gather.apply(y), x * mask)
    del mask
    d1 = th.randn(1, 3, H, W, generator=g).cuda()
    d2 = th.randn(1, 3, H, W, generator=g).cuda()
    o1, s1 = F.KernelWeighting.apply(d1, y)
    o2, s2 = F.KernelWeighting.apply(d2, y)
    o12, s12 = F.KernelWeighting.apply(d1 + 2 * d2, y)
    # (linearity between THREE fp32 evaluations, each within 1e-5 of the exact operator: o12's error + o1's + twice o2's)
    close(o12, o1 + 2 * o2, rtol=4 * 1e-5)
    assert th.equal(s1, s12)
    close(s1, y.double().sum((1, 2)), rtol=1e-5)


def test_4k_frame_indexing():
    """BASELINE configs[3] size (3840x2160, k=21): 441 * H * W = 3.66e9 elements per sample, past
    2^31 -- every offset in the kernels must be 64-bit (or a per-row rebased 32-bit buffer
    offset).  Fused forward+backward vs the composed boundary-level operators on the full
    frame, plus a full-width band vs the oracle at the bottom of the frame (largest offsets)."""
    from sbmc_amd import modules
    h, w = 2160, 3840
    g = th.Generator(device="cuda").manual_seed(31)
    rad = th.rand(1, 3, h, w, device="cuda", generator=g)
    ker = th.randn(1, K * K, h, w, device="cuda", generator=g)
    grads = [th.randn(1, 3, h, w, device="cuda", generator=g), th.randn(1, 1, h, w, device="cuda", generator=g),
             th.zeros(1, 1, h, w, device="cuda")]
    a_out, a_dd, a_dk = run_progressive(modules.ProgressiveKernelApply(splat=True, fused=True), [rad], [ker], grads, "cuda")
    b_out, b_dd, b_dk = run_progressive(modules.ProgressiveKernelApply(splat=True, fused=False), [rad], [ker], grads, "cuda")
    for a, b, n in zip(a_out, b_out, ("sum_r", "sum_w", "max_w")):
        err = (a - b).abs().max().item()
        assert err <= 1e-5 * b.abs().max().item(), (n, err)
    for a, b, n in ((a_dd[0], b_dd[0], "d_data"), (a_dk[0], b_dk[0], "d_kernels")):
        err = (a - b).abs().max().item()
        assert err <= 1e-5 * b.abs().max().item() + 1e-12, (n, err)


def test_4k_bottom_band_against_oracle(oracle):
    """The last 32 rows of a 4K frame computed inside the full frame == the oracle on the band
    extended by the kernel radius (interior rows of the band only)."""
    from sbmc_amd import modules
    h, w, band, p = 2160, 3840, 32, (K - 1) // 2
    g = th.Generator(device="cuda").manual_seed(32)
    rad = th.rand(1, 3, h, w, device="cuda", generator=g)
    ker = th.randn(1, K * K, h, w, device="cuda", generator=g)
    sr, sw, mw = modules.ProgressiveKernelApply(splat=True)(rad, ker, None, None, None)
    y0 = h - band - p
    o_sr, o_sw, o_mw = oracle.progressive_kernel_apply(
        rad[..., y0:, :].cpu().contiguous(), ker[..., y0:, :].cpu().contiguous(), None, None, None, splat=True)
    # rows >= p of the band see the same neighbourhood as in the full frame (bottom border included)
    close(sr[..., y0 + p:, :], o_sr[..., p:, :])
    close(sw[..., y0 + p:, :], o_sw[..., p:, :])
    close(mw[..., y0 + p:, :], o_mw[..., p:, :])


def test_4k_pointwise_layer_indexing():
    """The fused 1x1 layer at 3840x2160 (32-bit byte offsets up to 4.2 GB per batch element):
    sampled pixels of the output and of the input gradient against fp64 matmuls, the weight /
    bias gradients against sums over a pixel subset that carries all of the upstream gradient."""
    from sbmc_amd import functions as F
    hw = 3840 * 2160
    th.manual_seed(77)
    x = th.randn(1, 128, hw, device="cuda").requires_grad_()
    w = (th.randn(128, 128, device="cuda") / 128 ** 0.5).requires_grad_()
    b = th.randn(128, device="cuda").requires_grad_()
    assert F.pointwise_supported(x, 128)
    y = F.PointwiseLayer.apply(x, w, b, None, 1, 1, 0.0)
    # pixels at both ends, around every 2^k boundary that could break the offset arithmetic, and random ones
    idx = th.cat([th.arange(0, 260), th.arange(hw - 260, hw), th.arange(2 ** 22 - 130, 2 ** 22 + 130),
                  th.randint(0, hw, (4096,))]).unique().cuda()
    xs = x.detach()[0][:, idx].double()
    pre = w.detach().double() @ xs + b.detach().double().view(-1, 1)
    close(y.detach()[0][:, idx], pre.clamp(min=0).float(), rtol=1e-5)
    # upstream gradient only on the sampled pixels (and away from the kink)
    g = th.zeros_like(y)
    gs = th.randn(128, idx.numel(), device="cuda") * (pre.abs() > 1e-4).float()
    g[0][:, idx] = gs
    y.backward(g)
    gz = gs.double() * (pre > 0).double()
    close(x.grad[0][:, idx], (w.detach().double().t() @ gz).float(), rtol=1e-5)
    assert x.grad.abs().sum().item() == pytest.approx(x.grad[0][:, idx].abs().sum().item(), rel=1e-6)
    close_sum(w.grad, gz @ xs.t(), gz.abs() @ xs.abs().t(), what="gw")
    close_sum(b.grad, gz.sum(1), gz.abs().sum(1), what="gbias")


def test_fullsize_regressor_output_layer_backward_in_one_pass():
    """BASELINE configs[2] size: the 441-channel layer's backward over 8 x 1280 x 720 sample-pixels (13 GB of logit gradient)
    through pw_wide_bwd2_kernel, behind the real chain splat backward -> bound word -> one-pass kernel.  No oracle at
    this size; properties instead:
      * gx against the library's fp32 product w^T gz (two fp32 evaluations of a 441-term sum): 1e-5 of the tensor's scale;
      * gw and gbias -- sums over 7.4 M terms -- against float64 sums of the same products on the device;
      * LINEARITY: the gradients for 3 gz are 3 x the gradients for gz to the last bit but one (a power-of-two-free
        factor changes every rounding: 4e-7 relative), and the words scale with it;
      * the splat's bound word really bounds the logit gradient it was made for."""
    from sbmc_amd import _lib, functions as F
    L = _lib.lib()
    dev = th.device("cuda")
    g = th.Generator(device="cuda").manual_seed(5)
    B, cin, cout, hw = 8, 128, 441, H * W
    # a real logit gradient: the splat's backward of random upstream gradients (its magnitudes span many binades)
    rad = th.rand(1, B, 3, H, W, device=dev, generator=g)
    ker = (th.randn(1, B, cout, H, W, device=dev, generator=g) * 2).requires_grad_()
    sr, sw, mw = F.SplatAll.apply(rad, ker)
    gz5, = th.autograd.grad([sr, sw], [ker], [th.randn(sr.shape, device=dev, generator=g), th.randn(sw.shape, device=dev, generator=g)])
    word = F.known_amax(gz5)
    assert word is not None
    true_max = gz5.abs().max().item()
    assert true_max <= word.view(th.float32).item() <= 256.0 * true_max
    del sr, sw, mw, ker, rad
    gz = gz5.view(B, cout, hw)
    x = th.randn(B, cin, hw, device=dev, generator=g).relu_()
    w = th.randn(cout, cin, device=dev, generator=g) / cin ** 0.5
    xword = x.abs().max().reshape(1).view(th.int32).clone()

    def run(gzt, gword):
        groups = L.sbmc_pointwise_gw_wide_groups(B, hw)
        gwp = th.empty(groups, cout, cin, device=dev)
        gbp = th.empty(groups, cout, device=dev)
        gx = th.empty(B, cin, hw, device=dev)
        ws = th.empty(L.sbmc_pointwise_wide_bwd_ws_bytes(), dtype=th.uint8, device=dev)
        gxmax = th.zeros(1, dtype=th.int32, device=dev)
        _lib.check(L.sbmc_pointwise_wide_bwd_f32(_lib.ptr(gzt), _lib.ptr(x), _lib.ptr(w), _lib.ptr(gx), _lib.ptr(gwp), _lib.ptr(gbp),
                                                 _lib.ptr(ws), _lib.ptr(gword), _lib.ptr(xword), _lib.ptr(gxmax), B, cin, cout, hw,
                                                 _lib.current_stream(dev)), "wide_bwd")
        return gx, gwp.double().sum(0), gbp.double().sum(0), gxmax
    gx, gw, gb, gxmax = run(gz, word)
    assert gxmax.item() == gx.abs().max().reshape(1).view(th.int32).item()
    lib = th.bmm(w.t().unsqueeze(0).expand(B, -1, -1), gz)
    assert (gx - lib).abs().max().item() <= 1e-5 * lib.abs().max().item()
    del lib
    gw64 = th.zeros(cout, cin, dtype=th.float64, device=dev)
    gb64 = th.zeros(cout, dtype=th.float64, device=dev)
    for b in range(B):                                   # (float64 on the device, image by image: 13 GB at a time would not fit twice)
        gzd = gz[b].double()
        gw64 += gzd @ x[b].double().t()
        gb64 += gzd.sum(1)
        del gzd
    assert (gw - gw64).abs().max().item() <= 1e-5 * gw64.abs().max().item()
    assert (gb - gb64).abs().max().item() <= 1e-5 * gb64.abs().max().item()
    gz3 = gz * 3.0
    word3 = (word.view(th.float32) * 3.0).view(th.int32)
    gx3, gw3, gb3, _ = run(gz3, word3)
    assert (gx3 - 3.0 * gx).abs().max().item() <= 4e-6 * gx3.abs().max().item()
    assert (gw3 - 3.0 * gw).abs().max().item() <= 4e-6 * gw3.abs().max().item()
    assert (gb3 - 3.0 * gb).abs().max().item() <= 4e-6 * gb3.abs().max().item()
