"""Randomised sweep of the row-slab form of the splat (csrc/splat_fused.hip, `sbmc_splat_slab_*`): a frame cut
into random row slabs, every slab splatted on its own and the overhang rows merged as `dist.merge_overhang` does
across ranks, against the whole-frame `functions.SplatAll` on the same GPU (itself pinned to the oracle by
tests/ and tools/fuzz_gpu.py) -- and, for the small cases, against the CPU oracle's chain of
`progressive_kernel_apply` directly.  Values and gradients of the normalised output, 1e-5.

    python tools/fuzz_slab.py [--seconds 200] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_slab import close, sharded_state  # noqa: E402
from helpers import no_worse_than  # noqa: E402
from oracle import sbmc_oracle as orc  # noqa: E402
from sbmc_amd import functions as F  # noqa: E402


DK_RTOL = 1e-5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    t0, n, vs_oracle, halves = time.time(), 0, 0, 0
    # largest d_kernels error vs float64 in units of the bound max(1e-5 of scale, 2 x the oracle's own fp32 error,
    # 1e-5 of the cancelling terms -- see below): whole-frame GPU, sharded GPU; and how many cases needed more than 1e-5
    worst = [0.0, 0.0]
    beyond = 0
    while time.time() - t0 < args.seconds:
        k = int(rng.choice([3, 5, 7, 9, 13, 21, 21, 21]))
        p = (k - 1) // 2
        nslab = int(rng.choice([2, 2, 3, 4]))
        rows = [int(rng.randint(p, p + 40)) for _ in range(nslab)]        # every slab at least p rows
        H, W = sum(rows), int(rng.choice([1, 2, 63, 64, 65, 130, 200, rng.randint(1, 400)]))
        S = int(rng.choice([1, 2, 3, 5]))
        half = k == 21 and bool(rng.randint(3) == 0)
        bounds, y = [], 0
        for r in rows:
            bounds.append((y, y + r))
            y += r
        tag = "k%d H%d W%d S%d bounds %s half %d" % (k, H, W, S, bounds, half)
        th.manual_seed(int(rng.randint(1 << 30)))
        rad = th.empty(1, S, 3, H, W).exponential_(1.0)
        kern = th.randn(1, S, k * k, H, W) * float(rng.choice([1.0, 3.0]))
        if half:
            kern = kern.half()
        d_out = th.randn(1, 3, H, W)
        try:
            r1, k1 = rad.cuda().requires_grad_(), kern.cuda().requires_grad_()
            if not F.splat_slab_supported(r1[..., :rows[0], :].contiguous(), k1[..., :rows[0], :].contiguous(), 0, p):
                continue
            sr, sw, _ = F.SplatAll.apply(r1, k1)
            o1 = sr / (sw + 1e-8)
            o1.backward(d_out.cuda())
            r2, k2 = rad.cuda().requires_grad_(), kern.cuda().requires_grad_()
            sr, sw, _ = sharded_state(r2, k2, bounds, p)
            o2 = sr / (sw + 1e-8)
            o2.backward(d_out.cuda())
            # d_kernels, GPU slabs vs GPU whole frame: the one element per destination that receives the routed
            # gradient of the running max is a cancellation residual in fp32 whose value depends on the order of the
            # merges (up to 2.5e-4 between two correct fp32 codes) -- the small cases below hold
            # it to the float64 evaluation instead; half logit gradients: one half rounding each
            rt = 2e-3 if half else 2e-4       # (GPU vs GPU, any size; the small cases go to the float64 evaluation)
            close(o2, o1, what="normalised output")
            close(r2.grad, r1.grad, what="d_radiance")
            close(k2.grad.float(), k1.grad.float(), rtol=rt, what="d_kernels")
            if H * W * S * k * k < 3e6 and not half:         # small enough for the CPU oracle
                ro, ko = rad.clone().requires_grad_(), kern.clone().requires_grad_()
                st = (None, None, None)
                for s in range(S):
                    st = orc.progressive_kernel_apply(ro[:, s], ko[:, s], *st, splat=True)
                oo = st[0] / (st[1] + 1e-8)
                oo.backward(d_out)
                close(o2, oo, what="normalised output vs oracle")
                close(r2.grad, ro.grad, what="d_radiance vs oracle")
                # d_kernels against the float64 evaluation of the same chain: 1e-5, or no worse than twice the
                # oracle's own fp32 error on the routed arg-max elements (tests/helpers.no_worse_than)
                r64, k64 = rad.double().requires_grad_(), kern.double().requires_grad_()
                st = (None, None, None)
                for s in range(S):
                    st = orc.progressive_kernel_apply(r64[:, s], k64[:, s], *st, splat=True)
                (st[0] / (st[1] + 1e-8)).backward(d_out.double())
                # d_kernels[s, tap, source] = e (dW[q] + sum_c dR[q, c] D[c]) at its destination q (+ the routed
                # gradient of the running max, analytically zero with this objective).  With out = sum_r / sum_w,
                # dR = g / sum_w and dW = -g . out / sum_w: the bracket is g . (D - out) / sum_w, the difference of two
                # terms of size |g| |D| / sum_w that nearly cancel wherever one tap dominates the softmax (out ~ D).
                # An fp32 evaluation is off by rounding relative to the TERMS, not to their difference: what the
                # forward's sum_r / sum_w carry (each within 1e-5, typically 1e-6) is multiplied by |dW| + |dR| |D|.
                # (The failing case of an earlier version of this tool: a 2-pixel-wide frame, D = 8, g = 2.5: value
                # 0.517, oracle off by 1.1e-5, GPU by 5.3e-5 = 2.6e-6 of the two terms.)  So: 1e-5 of the tensor's
                # scale, or twice the oracle's own error, or 1e-5 of the largest |dW| + |dR| . |D| of the frame.
                sr64, sw64 = st[0].detach(), st[1].detach()
                g64 = d_out.double()
                d_r = g64.abs() / (sw64 + 1e-8)
                d_w = ((g64 * sr64).sum(1, keepdim=True) / (sw64 + 1e-8) ** 2).abs()
                terms = (d_w + d_r.sum(1, keepdim=True) * rad.double().abs().max()).max().item()
                floor = DK_RTOL * terms
                scale = k64.grad.abs().max().item()
                eo = (ko.grad.double() - k64.grad).abs().max().item()
                bound = max(DK_RTOL * scale, 2.0 * eo, floor)
                e1 = (k1.grad.detach().cpu().double() - k64.grad).abs().max().item()
                e2 = (k2.grad.detach().cpu().double() - k64.grad).abs().max().item()
                if not (e1 <= bound and e2 <= bound):
                    d = (k1.grad.detach().cpu().double() - k64.grad).abs()
                    at = [int(v) for v in th.unravel_index(d.argmax(), d.shape)]
                    _, sidx, tap, yy, xx = at
                    dy, dx = tap // k, tap % k
                    Y, X = yy + dy - p, xx + dx - p                      # the destination this splat tap lands on
                    print("worst element: sample %d tap (%d, %d) source (%d, %d) -> destination (%d, %d); gpu %.9g oracle "
                          "%.9g float64 %.9g" % (sidx, dy, dx, yy, xx, Y, X, k1.grad[tuple(at)].item(),
                                                 ko.grad[tuple(at)].item(), k64.grad[tuple(at)].item()), flush=True)
                    srt = d.flatten().sort(descending=True)[0][:8]
                    print("largest errors:", ["%.2e" % v for v in srt.tolist()], "median", "%.2e" % d.median().item(), flush=True)
                assert e1 <= bound and e2 <= bound, "d_kernels vs float64: whole-frame %.3e, sharded %.3e > max(1e-5 * scale " \
                    "= %.3e, 2 x oracle %.3e, 1e-5 of the cancelling terms %.3e)" % (e1, e2, DK_RTOL * scale, 2.0 * eo, floor)
                worst[0] = max(worst[0], e1 / bound)
                worst[1] = max(worst[1], e2 / bound)
                beyond += int(max(e1, e2) > DK_RTOL * scale)
                vs_oracle += 1
            halves += int(half)
        except Exception:
            print("FAILED case:", tag, flush=True)
            raise
        n += 1
    print("fuzz ok: %d random sharded frames (%d also against the CPU oracle and float64, %d with half logits) in %.0f s; "
          "d_kernels vs float64: %d of those cases beyond 1e-5 of the tensor's scale, all within max(2 x the oracle's own "
          "fp32 error, 1e-5 of the terms that cancel in g . (D - out) / sum_w); largest error in units of the bound: "
          "whole-frame GPU %.2f, sharded GPU %.2f" % (n, vs_oracle, halves, time.time() - t0, beyond, worst[0], worst[1]))


if __name__ == "__main__":
    main()
