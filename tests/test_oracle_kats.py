"""The oracle against the reference's own known-answer tests (no GPU).

Each test restates one analytic check of /root/reference/tests (cited), with the
oracle's autograd wrappers standing where `sbmc.functions` stands in the reference.
(In the authoring container the reference's *unmodified* test files are also run on
top of the oracle: `python -m oracle.pin_against_reference`, 17/17 pass.)
"""
import numpy as np
import pytest
import torch as th
from torch.autograd import gradcheck


def test_kernel_weighting_forward_delta(oracle):
    """reference tests/test_functions.py:43-70"""
    bs, c, h, w, ksize = 4, 5, 16, 16, 5
    data = th.zeros(bs, c, h, w)
    idx, y, x = 1, h // 2, w // 2
    data[idx, 0, y, x], data[idx, 1, y, x], data[idx, 2, y, x] = 1.4, 2.4, 3.4
    for dy in range(-(ksize // 2), ksize // 2 + 1):
        for dx in range(-(ksize // 2), ksize // 2 + 1):
            weights = th.zeros(bs, ksize, ksize, h, w)
            weights[idx, ksize // 2 + dy, ksize // 2 + dx, y - dy, x - dx] = 0.5
            o, s = oracle.KernelWeighting.apply(data, weights)
            for ch, val in enumerate((1.4, 2.4, 3.4)):
                assert o[idx, ch, y - dy, x - dx].item() == pytest.approx(val * 0.5, abs=1e-7)
            assert s[idx, y - dy, x - dx].item() == pytest.approx(0.5, abs=1e-7)
            assert o.abs().sum().item() == pytest.approx(0.5 * (1.4 + 2.4 + 3.4), rel=1e-6)


@pytest.mark.parametrize("ksize", [3, 5, 7])
def test_kernel_weighting_backward_delta(oracle, ksize):
    """reference tests/test_functions.py:72-103"""
    bs, chans, h, w = 3, 5, 16, 16
    x, y = w // 2, h // 2
    for b in range(bs):
        for c in (0, chans - 1):
            data = th.full((bs, chans, h, w), 7.0, requires_grad=True)
            weights = th.ones(bs, ksize, ksize, h, w, requires_grad=True)
            o, s = oracle.KernelWeighting.apply(data, weights)
            o_grad = th.zeros_like(o)
            o_grad[b, c, y, x] = 1.1
            o.backward(o_grad)
            g = data.grad.clone()
            for dy in range(-(ksize // 2), ksize // 2 + 1):
                for dx in range(-(ksize // 2), ksize // 2 + 1):
                    assert g[b, c, y + dy, x + dx].item() == pytest.approx(1.1, abs=1e-6)
                    g[b, c, y + dy, x + dx] = 0.0
            assert g.abs().max().item() == 0.0
            assert weights.grad[b, ksize - 1, ksize - 1, y, x].item() == pytest.approx(7.7, abs=1e-3)


def test_kernel_weighting_gradcheck(oracle):
    """reference tests/test_functions.py:105-144 (same eps / tolerances)"""
    th.manual_seed(0)
    bs, c, h, w, ksize = 2, 3, 16, 16, 3
    data = 2 * th.randn(bs, c, h, w)
    weights = th.randn(bs, ksize, ksize, h, w)
    assert gradcheck(oracle.KernelWeighting.apply, (data.clone().requires_grad_(), weights),
                     eps=1e-4, atol=5e-2, rtol=5e-4, check_forward_ad=False)
    assert gradcheck(oracle.KernelWeighting.apply, (data, weights.clone().requires_grad_()),
                     eps=1e-4, atol=5e-2, rtol=5e-4)


@pytest.mark.parametrize("ksize", [3, 5, 7, 9])
def test_scatter2gather_delta(oracle, ksize):
    """reference tests/test_functions.py:164-185 (one batch index, all positions / taps)"""
    bs, h, w = 2, 32, 32
    idx = 1
    for y in range(h // 2 - ksize // 2, h // 2 + ksize // 2 + 1, max(1, ksize // 2)):
        for x in range(w // 2 - ksize // 2, w // 2 + ksize // 2 + 1, max(1, ksize // 2)):
            for ky in range(ksize):
                for kx in range(ksize):
                    scatter = th.zeros(bs, ksize, ksize, h, w)
                    scatter[idx, ky, kx, y, x] = 0.5
                    gather = oracle.Scatter2Gather.apply(scatter)
                    dy, dx = ky - ksize // 2, kx - ksize // 2
                    assert gather[idx, ksize - 1 - ky, ksize - 1 - kx, y + dy, x + dx].item() == 0.5
                    assert gather.sum().item() == 0.5


@pytest.mark.slow
def test_scatter2gather_gradcheck_and_involution(oracle):
    """reference tests/test_functions.py:187-208; for odd k the op is an involution"""
    th.manual_seed(0)
    weights = th.randn(2, 3, 3, 32, 32, requires_grad=True)
    assert gradcheck(oracle.Scatter2Gather.apply, (weights,), eps=1e-4, atol=5e-2, rtol=5e-4)
    x = th.randn(1, 5, 5, 9, 11)
    y = oracle.Scatter2Gather.apply(oracle.Scatter2Gather.apply(x))
    # twice = identity wherever the partner pixel is inside the image
    mask = oracle.Scatter2Gather.apply(oracle.Scatter2Gather.apply(th.ones_like(x)))
    assert th.equal(y, x * mask)


@pytest.mark.parametrize("splat", [True, False])
def test_kernel_apply_delta(oracle, splat):
    """reference tests/test_modules.py:63-99"""
    bs, c, h, w, k = 4, 5, 16, 16, 3
    data = th.zeros(bs, c, h, w)
    weights = th.zeros(bs, k * k, h, w)
    y, x, val = h // 2, w // 2, 1.43
    data[0, 0, y, x] = val
    weights[0, :, y, x] = 1.0
    output, sum_w = oracle.kernel_apply(data, weights, softmax=False, splat=splat)
    assert output[0, 0, y, x].item() == pytest.approx(val, abs=1e-4)
    if splat:
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                assert output[0, 0, y + dy, x + dx].item() == pytest.approx(val, abs=1e-4)
                assert sum_w[0, 0, y + dy, x + dx].item() == pytest.approx(1, abs=1e-4)
    else:
        assert sum_w[0, 0, y, x].item() == pytest.approx(k * k, abs=1e-4)


@pytest.mark.parametrize("splat", [True, False])
def test_progressive_kernel_apply_init(oracle, splat):
    """reference tests/test_modules.py:102-140"""
    bs, c, h, w, k = 4, 5, 16, 16, 3
    data = th.zeros(bs, c, h, w)
    weights = th.zeros(bs, k * k, h, w)
    y, x, val = h // 2, w // 2, 1.43
    data[0, 0, y, x] = val
    weights[0, :, y, x] = 1.0
    output, sum_w, max_w = oracle.progressive_kernel_apply(data, weights, None, None, None, splat=splat)
    assert output[0, 0, y, x].item() == pytest.approx(val, abs=1e-4)
    if splat:
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                assert output[0, 0, y + dy, x + dx].item() == pytest.approx(val, abs=1e-4)
    else:
        assert sum_w[0, 0, y, x].item() == pytest.approx(k * k, abs=1e-4)
    with pytest.raises(RuntimeError):
        oracle.progressive_kernel_apply(data, weights, None, sum_w, None, splat=splat)


def test_oracle_matches_committed_golden_ops(oracle):
    """The committed fixtures were produced by the reference's functions.py running on the
    oracle; this guards the oracle (and the fixtures) against silent drift."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops.npz"))
    for tag in ("a", "b", "c"):
        data = th.from_numpy(g[tag + ".data"]).requires_grad_()
        wts = th.from_numpy(g[tag + ".weights"]).requires_grad_()
        o, s = oracle.KernelWeighting.apply(data, wts)
        th.autograd.backward([o, s], [th.from_numpy(g[tag + ".d_output"]), th.from_numpy(g[tag + ".d_sum_w"])])
        assert th.equal(o.detach(), th.from_numpy(g[tag + ".output"]))
        assert th.equal(s.detach(), th.from_numpy(g[tag + ".sum_w"]))
        assert th.equal(data.grad, th.from_numpy(g[tag + ".d_data"]))
        assert th.equal(wts.grad, th.from_numpy(g[tag + ".d_weights"]))
        x = th.from_numpy(g[tag + ".s2g_in"])
        assert th.equal(oracle.Scatter2Gather.apply(x), th.from_numpy(g[tag + ".s2g_out"]))


def test_float64_evaluation_of_the_oracle_agrees_with_pure_torch(oracle):
    """The oracle's double instantiation (the tests' yardstick for cancellation-limited gradients) vs an
    independent pure-torch float64 restatement, and the float32 oracle within fp32 rounding of both."""
    from helpers import progressive_fp64, progressive_fp64_torch, run_progressive
    th.manual_seed(3)
    for splat, k in ((True, 5), (False, 3), (True, 7)):
        d = [th.rand(1, 3, 9, 13) for _ in range(3)]
        kk = [th.randn(1, k * k, 9, 13) * 2 for _ in range(3)]
        g = [th.randn(1, 3, 9, 13), th.randn(1, 1, 9, 13), th.randn(1, 1, 9, 13)]
        a, da, ka = progressive_fp64(d, kk, g, splat=splat)
        b, db, kb = progressive_fp64_torch(d, kk, g, splat=splat)
        for x, y in zip(list(a) + da + ka, list(b) + db + kb):
            assert (x - y).abs().max().item() <= 1e-12 * max(1.0, y.abs().max().item())
        o, do, ko = run_progressive(lambda *args: oracle.progressive_kernel_apply(*args, splat=splat), d, kk, g, "cpu")
        for x, y in zip(list(o) + do + ko, list(a) + da + ka):
            assert (x.double() - y).abs().max().item() <= 1e-5 * max(1.0, y.abs().max().item())
