"""Randomised parity sweep of the fused 1x1-convolution layer (csrc/pointwise.hip) against an
fp64 torch restatement (conv1x1 + context term + bias + activation), forward and backward:
random batch / sample count / channel counts / pixel counts / context modes / activations.

    python tools/fuzz_pointwise.py [--seconds 120] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import close  # noqa: E402
from sbmc_amd import functions as F  # noqa: E402


def close_sum(a, ref, rtol, atol):
    err = (a.double() - ref).abs().max().item()
    bound = rtol * ref.abs().max().item() + atol
    assert err <= bound, "max err %.3e > %.3e (scale %.3e)" % (err, bound, ref.abs().max().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    t0, n, fused_bwd, halves, half_trains, scaled, scaled_bwd, wide_fused = time.time(), 0, 0, 0, 0, 0, 0, 0
    calls = []
    F.enable_kernel_timing(calls)
    while time.time() - t0 < args.seconds:
        S = int(rng.choice([1, 1, 2, 3, 8]))
        B = S * int(rng.choice([1, 1, 2, 3]))
        cin = int(rng.choice([1, 3, 31, 32, 33, 64, 93, 96, 100, 127, 128]))
        cout = int(rng.choice([1, 5, 32, 33, 100, 128, 128, 129, 200, 441]))
        hw = 4 * int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 100, 1000, rng.randint(1, 3000)]))
        tm = int(rng.choice([0, 0, 1, 2]))
        act = int(rng.choice([0, 1, 2]))
        needx = bool(rng.randint(2))
        slope = float(rng.choice([0.01, 0.2])) if act == 2 else 0.0
        tag = "B%d S%d cin%d cout%d hw%d t%d act%d dx%d" % (B, S, cin, cout, hw, tm, act, needx)
        th.manual_seed(int(rng.randint(1 << 30)))
        # round 5: half of the cases carry MAGNITUDE WORDS (as tensors inside a chain do): the two-plane kernels run -- and,
        # for a wide linear layer with a data gradient, the one-pass backward -- at random magnitudes, with words that are
        # the exact maxima or bounds up to 50 x too large
        words = bool(rng.randint(2))
        xmag = 10.0 ** rng.uniform(-4, 3) if words else 1.0
        gmag = 10.0 ** rng.uniform(-6, 2) if words else 1.0
        tag += " words%d" % words
        x0 = th.randn(B, cin, hw, device="cuda") * xmag
        w0 = th.randn(cout, cin, device="cuda") / max(cin, 1) ** 0.5 / xmag
        b0 = th.randn(cout, device="cuda")
        t0_ = None if tm == 0 else (th.randn(B // S, cout, device="cuda") if tm == 1
                                    else th.randn(B // S, cout, hw, device="cuda"))
        try:
            assert F.pointwise_supported(x0, cout)

            def leaves(dt):
                return [None if v is None else v.to(dt).requires_grad_(needx or i > 0)
                        for i, v in enumerate((x0, w0, b0, t0_))]
            x, w, b, t = leaves(th.float64)
            pre = th.matmul(w, x) + b.view(1, -1, 1)
            if tm == 1:
                pre = pre + t.repeat_interleave(S, 0).unsqueeze(-1)
            elif tm == 2:
                pre = pre + t.repeat_interleave(S, 0)
            ref = pre if act == 0 else th.nn.functional.leaky_relu(pre, slope)
            g = th.randn(B, cout, hw, device="cuda") * (pre.detach().abs() > 1e-4).float() * gmag
            ref.backward(g.double())
            x2, w2, b2, t2 = leaves(th.float32)
            del calls[:]
            if words:
                def word(v):
                    return (v.detach().abs().max() * float(rng.choice([1.0, 1.0, 3.0, 50.0]))).reshape(1).view(th.int32).clone()
                F.tag_amax(x2, word(x2))
                F.tag_amax(g, word(g))
            out = F.PointwiseLayer.apply(x2, w2, b2, t2, S, act, slope)
            if words:
                yw = F.known_amax(out)
                assert yw is not None and yw.item() == out.detach().abs().max().reshape(1).view(th.int32).item(), "word of y"
            out.backward(g)
            if words:
                scaled += 1
                names = [c[0] for c in calls]
                if any(nm.startswith("pointwise_wide_bwd") for nm in names):
                    wide_fused += 1
                if cout <= 128:
                    scaled_bwd += 1
                if needx and (cout <= 128 or any(nm.startswith("pointwise_wide_bwd") for nm in names)) and not (
                        cout <= 128 and tm == 2 and False):
                    gw_ = F.known_amax(x2.grad)
                    if gw_ is not None:
                        assert gw_.item() == x2.grad.abs().max().reshape(1).view(th.int32).item(), "word of gx"
            close(out, ref.float(), rtol=1e-5)
            if needx:
                close(x2.grad, x.grad.float(), rtol=1e-5)
            # weight / bias / per-image context gradients are sums over up to B*hw terms of size
            # ~|g| |x|: compare them relative to that sum's natural scale sqrt(#terms) as well, or a
            # [1, 1] gradient that happens to cancel to ~0 fails on fp32 summation noise alone
            red = (B * hw) ** 0.5
            close_sum(w2.grad, w.grad, 3e-5, 1e-6 * red * gmag * xmag)
            close_sum(b2.grad, b.grad, 3e-5, 1e-6 * red * gmag)
            if tm == 1:
                close_sum(t2.grad, t.grad, 3e-5, 1e-6 * (S * hw) ** 0.5 * gmag)
            elif tm == 2:
                close(t2.grad, t.grad.float(), rtol=1e-5)
            if n % 3 == 0 and not words:            # half-storage forward (inference) on the same case
                xin = x0.half() if n % 2 else x0
                refh = F.PointwiseLayer.apply(xin.float(), w0, b0, t0_, S, act, slope)
                with th.no_grad(), th.autocast("cuda", dtype=th.float16):
                    assert F.pointwise_half_supported(xin, cout)
                    yh = F.pointwise_half(xin, w0, b0, t0_, S, act, slope)
                errh = (yh.float() - refh).abs().max().item()
                assert errh <= 1e-3 * refh.abs().max().item() + 1e-3, "half forward: %.3e" % errh
                halves += 1
            if n % 3 == 1 and cout <= 128 and not words:      # half-storage training (autocast): forward + backward on half tensors
                half_training_case(x0, w0, b0, t0_, S, tm, act, slope, x_half=bool(n % 2))
                half_trains += 1
        except Exception:
            print("FAILED case:", tag, flush=True)
            raise
        n += 1
        fused_bwd += int(cout <= 128)
    F.enable_kernel_timing(None)
    print("fuzz ok: %d random layers (%d with the fused backward, %d half-storage forwards, %d half-storage "
          "forward+backward; %d with magnitude words = the two-plane kernels, %d of them through the fused two-plane "
          "backward, %d through the wide layer's one-pass backward) in %.0f s" % (
              n, fused_bwd, halves, half_trains, scaled, scaled_bwd, wide_fused, time.time() - t0))


def half_training_case(x0, w0, b0, t0, S, tm, act, slope, x_half):
    """PointwiseLayer(..., half=True) forward + backward vs fp32 torch on exactly the values the kernels see
    (tests/test_gpu_ops.py::test_pointwise_half_training): an all-half layer runs on the f16 matrix pipe in both
    directions (weights and gz rounded to half, exact products, fp32 sums)."""
    x0, w0, b0 = x0.detach(), w0.detach(), b0.detach()       # (main's fp32 leaves are these very tensors)
    t0 = None if t0 is None else t0.detach()
    B, cin, hw = x0.shape
    cout = w0.shape[0]
    x = x0.half() if x_half else x0
    w, b = w0.clone().requires_grad_(), b0.clone().requires_grad_()
    t = None if t0 is None else t0.clone().requires_grad_()
    xg = x.clone().requires_grad_()
    y = F.PointwiseLayer.apply(xg, w, b, t, S, act, slope, True)
    gy = th.randn(B, cout, hw, device="cuda").half()
    y.backward(gy)
    xr = x.float()
    wq = w0.half().float() if x_half else w0
    pre = th.einsum("oc,bcp->bop", wq, xr) + b0.view(1, -1, 1)
    if tm == 1:
        pre = pre + t0.repeat_interleave(S, 0).unsqueeze(-1)
    elif tm == 2:
        pre = pre + t0.repeat_interleave(S, 0)
    yr = pre if act == 0 else th.where(pre > 0, pre, pre * slope)
    assert (y.float() - yr).abs().max().item() <= 2.0 ** -10 * yr.abs().max().item() + 1e-3, "half y"
    g = gy.float()
    gz = g if act == 0 else th.where(y > 0, g, g * slope)
    gzq = gz.half().float() if x_half else gz
    red = (B * hw) ** 0.5
    # (the sums' yardstick in FLOAT64: torch's fp32 einsum of 1e4 terms is itself 3e-4 off -- ten times further from float64 than the
    # kernel's fp32 accumulators are, tools/dev/half_gw_check.py -- and used to fail this check on the kernel's behalf)
    close_sum(w.grad, th.einsum("bop,bcp->oc", gzq.double(), xr.double()), 3e-5, 1e-6 * red)
    close_sum(b.grad, gz.double().sum((0, 2)), 3e-5, 1e-6 * red)
    gxr = th.einsum("oc,bop->bcp", wq, gzq)
    tol = 2.0 ** -10 if x_half else 1e-5
    assert (xg.grad.float() - gxr).abs().max().item() <= tol * gxr.abs().max().item() + 1e-6, "half gx"
    if tm == 1:
        close_sum(t.grad, gz.double().view(B // S, S, cout, hw).sum((1, 3)), 3e-5, 1e-6 * (S * hw) ** 0.5)
    elif tm == 2:
        close(t.grad, gz.view(B // S, S, cout, hw).sum(1), rtol=1e-5)


if __name__ == "__main__":
    main()
