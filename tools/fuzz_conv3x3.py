"""Randomised parity sweep of the split-precision 3 x 3 convolution (csrc/conv3x3.hip) against float64.

Random shapes (ragged heights / widths, batches, every channel combination the kernels take up to 512), random
magnitudes of activations, weights and gradients over ~12 decades, forward + both gradients through the autograd
function, with and without the fused bias / activation epilogue.  The yardstick is torch's own fp32 convolution on
the same inputs: a case passes if every tensor is no further from float64 than three times the library's error (with
a floor of 2e-7 of the tensor's largest value); the summary prints the worst ratio and the worst error.

    python tools/fuzz_conv3x3.py [--cases 300] [--seed 0]
"""
import argparse
import os
import sys

import torch as th
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbmc_amd import functions as funcs  # noqa: E402


def cl(t):
    return t.contiguous(memory_format=th.channels_last)


def err(a, r):
    return (a.double() - r).abs().max().item() / max(r.abs().max().item(), 1e-300)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    dev = th.device("cuda")
    g = th.Generator(device="cpu").manual_seed(args.seed)

    def ri(lo, hi):
        return int(th.randint(lo, hi + 1, (1,), generator=g).item())

    def mag():
        return 10.0 ** (th.rand(1, generator=g).item() * 12.0 - 8.0)

    worst = {"y": 0.0, "gx": 0.0, "gw": 0.0, "gb": 0.0}
    worst_abs = dict(worst)
    fails = 0
    for case in range(args.cases):
        cin = 128 * ri(1, 4) if ri(0, 3) else 32 * ri(1, 6)
        cout = 128 * [1, 2, 4][ri(0, 2)]       # (what the bias / activation pass behind the convolution takes as well)
        if cin % 128:                       # (the weight gradient kernel wants multiples of 128 both ways;
            cin = 128 * max(1, cin // 128)  #  the function falls back to the library otherwise: not this sweep's subject)
        b, h, w = ri(1, 2), ri(1, 70), ri(2, 90)
        act, slope = [(0, 0.0), (1, 0.0), (2, 0.01)][ri(0, 2)]
        fused = bool(ri(0, 1))
        x = (th.randn(b, cin, h, w, generator=g) * mag()).to(dev)
        wt = (th.randn(cout, cin, 3, 3, generator=g) * mag()).to(dev)
        bias = (th.randn(cout, generator=g) * x.abs().max().item() * wt.abs().max().item() * 10).to(dev)
        gy = (th.randn(b, cout, h, w, generator=g) * mag()).to(dev)
        if act:                             # keep pre-activations away from zero: a sign flip is not a rounding error
            bias = bias.abs() * 50 + 1e-30

        def reference(dt, layout):
            xs, ws, bs = (layout(t.to(dt)).clone().requires_grad_(True) for t in (x, wt, bias.view(1, -1, 1, 1)))
            z = F.conv2d(xs, ws, padding=1) + bs
            y = z if act == 0 else (F.relu(z) if act == 1 else F.leaky_relu(z, slope))
            return (y.detach(),) + th.autograd.grad(y, (xs, ws, bs), layout(gy.to(dt)))

        ref = reference(th.float64, lambda t: t)
        lib = reference(th.float32, lambda t: t)
        xs, ws, bs = cl(x).clone().requires_grad_(True), wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        if fused:
            y, _ = funcs.Conv3x3BiasActNHWC.apply(xs, ws, bs, act, slope)
        else:
            y = funcs.BiasActNHWC.apply(funcs.Conv3x3NHWC.apply(xs, ws), bs, act, slope)
        ours = (y.detach(),) + th.autograd.grad(y, (xs, ws, bs), cl(gy))
        bad = []
        for name, a, l, r in zip(("y", "gx", "gw", "gb"), ours, lib, ref):
            ea, el = err(a.reshape(r.shape), r), err(l, r)
            ratio = ea / max(el, 1e-7)
            worst[name] = max(worst[name], ratio)
            worst_abs[name] = max(worst_abs[name], ea)
            if ea > max(3.0 * el, 2e-7):
                bad.append("%s %.2e (library %.2e)" % (name, ea, el))
        if bad:
            fails += 1
            print("case %d FAILS: b %d %d->%d %dx%d act %d fused %d: %s" % (case, b, cin, cout, h, w, act, fused, ", ".join(bad)))
    print("%d cases, %d failures; worst error / max(library error, 1e-7): %s; worst error of ours: %s" % (
        args.cases, fails, ", ".join("%s %.2f" % kv for kv in worst.items()),
        ", ".join("%s %.2e" % kv for kv in worst_abs.items())))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
