"""Randomised sweep of the activation adjoint that rides in another node's backward pass (csrc ADJ, ABI 7): a chain's
layer followed by (a) another 3 x 3 layer, (b) the pooling + skip node, (c) the bilinear x2 + concatenation -- each with and
without the `_AdjLink`.  The linked form must give the unlinked form's gradients TO THE BIT (the same products, the same
rounding), except the producing layer's bias gradient, whose partial sums are added up in another fixed order (1e-5 of its
largest value).  Random shapes (ragged tiles, batches, channel counts up to 512: whole-tile and stream-K launches), random
magnitudes over ~8 decades, LeakyReLU / ReLU.

    python tools/fuzz_conv3x3_chain.py [--cases 300] [--seed 0]
"""
import argparse
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbmc_amd import functions as funcs  # noqa: E402


def cl(t):
    return t.contiguous(memory_format=th.channels_last)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    dev = th.device("cuda")
    g = th.Generator(device="cpu").manual_seed(args.seed)
    ri = lambda lo, hi: int(th.randint(lo, hi + 1, (1,), generator=g).item())
    mag = lambda: 10.0 ** (th.rand(1, generator=g).item() * 8.0 - 5.0)
    fails, kinds, worst_gb = 0, [0, 0, 0], 0.0
    for case in range(args.cases):
        kind = ri(0, 2)
        kinds[kind] += 1
        c0, c1, c2 = (128 * [1, 2, 3, 4][ri(0, 3)] for _ in range(3))
        c1 = min(c1, 512) if c1 != 384 else 256
        c2 = c2 if c2 != 384 else 256            # (output channels: what sbmc_bias_act_nhwc_supported takes, as Conv3x3BiasActNHWC.supported asks)
        b = ri(1, 2)
        h, w = (ri(1, 70), ri(2, 90)) if kind == 0 else ((2 * ri(1, 30), 2 * ri(1, 40)) if kind == 1 else (ri(1, 30), ri(1, 40)))
        act, slope = [(1, 0.0), (2, 0.01), (2, 0.2)][ri(0, 2)]
        x = cl((th.randn(b, c0, h, w, generator=g) * mag()).to(dev))
        w1 = (th.randn(c1, c0, 3, 3, generator=g) * mag()).to(dev)
        b1 = (th.randn(c1, generator=g) * x.abs().max().item() * w1.abs().max().item() * 3).to(dev)
        w2 = (th.randn(c2, c1, 3, 3, generator=g) * mag()).to(dev)
        b2 = th.randn(c2, generator=g).to(dev)
        gm = mag()
        if kind == 0:
            gouts = (cl((th.randn(b, c2, h, w, generator=g) * gm).to(dev)),)
        elif kind == 1:
            gouts = (cl((th.randn(b, c1, h // 2, w // 2, generator=g) * gm).to(dev)), cl((th.randn(b, c1, h, w, generator=g) * gm).to(dev)))
        else:
            left = cl(th.randn(b, 128, 2 * h, 2 * w, generator=g).to(dev))
            gouts = (cl((th.randn(b, c1 + 128, 2 * h, 2 * w, generator=g) * gm).to(dev)),)

        def run(linked):
            leaves = [t.clone().requires_grad_(True) for t in (x, w1, b1)]
            y1, a1 = funcs.Conv3x3BiasActNHWC.apply(leaves[0], leaves[1], leaves[2], act, slope, None, linked)
            funcs.tag_amax(y1, a1)
            link = funcs.Conv3x3BiasActNHWC.adj_link_for(y1)
            if kind == 0:
                extra = [t.clone().requires_grad_(True) for t in (w2, b2)]
                outs = (funcs.Conv3x3BiasActNHWC.apply(y1, extra[0], extra[1], act, slope, link, False)[0],)
            elif kind == 1:
                extra = []
                outs = funcs.PoolSkip.apply(y1, link)
            else:
                extra = [left.clone().requires_grad_(True)]
                outs = (funcs.UpsampleCatNHWC.apply(y1, extra[0], 0, 0, link),)
            grads = th.autograd.grad(outs, leaves + extra, gouts)
            if linked and not (link is not None and link.taken and link.done is None):
                raise RuntimeError("the link was not used")
            return [o.detach().clone() for o in outs] + list(grads), len(outs)

        try:
            (plain, n), (linked, _) = run(False), run(True)
        except Exception as e:                                      # noqa: BLE001
            fails += 1
            print("case %d RAISES (kind %d, b %d, %d->%d->%d, %dx%d, act %d): %s" % (case, kind, b, c0, c1, c2, h, w, act, e))
            continue
        bad = []
        for i, (a, c) in enumerate(zip(plain, linked)):
            if i == n + 2:                                           # the producing layer's bias gradient
                e = (a - c).abs().max().item() / max(a.abs().max().item(), 1e-300)
                worst_gb = max(worst_gb, e)
                if e > 1e-5:
                    bad.append("gb1 %.2e" % e)
            elif not th.equal(a, c):
                bad.append("tensor %d differs by %.2e of its scale" % (i, (a - c).abs().max().item() / max(a.abs().max().item(), 1e-300)))
        if bad:
            fails += 1
            print("case %d FAILS (kind %d, b %d, %d->%d->%d, %dx%d, act %d slope %g): %s" % (case, kind, b, c0, c1, c2, h, w, act, slope, ", ".join(bad)))
    print("%d cases (%d conv -> conv, %d conv -> pool + skip, %d conv -> upsample + cat), %d failures; worst bias-gradient "
          "difference %.2e of its scale" % (args.cases, kinds[0], kinds[1], kinds[2], fails, worst_gb))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
