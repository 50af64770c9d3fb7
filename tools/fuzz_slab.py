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


DK_RTOL = 5e-5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    t0, n, vs_oracle, halves = time.time(), 0, 0, 0
    worst = [0.0, 0.0]     # largest d_kernels error vs float64 in units of the oracle's own: whole-frame GPU, sharded GPU
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
            rt = 2e-3 if half else 2e-4
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
                # (DK_RTOL, as tools/fuzz_gpu.py: with this objective the routed element is analytically zero --
                # d(out)/d(max) = 0 -- so its fp32 value is pure rounding noise of 441-term sums in every
                # implementation.  The summary line reports the largest error seen for the whole-frame and the
                # sharded GPU path in units of the oracle's own: the two turn out equal, sharding adds none.)
                no_worse_than(k2.grad, ko.grad, k64.grad, rtol=DK_RTOL, slack=4.0, what="d_kernels vs float64")
                eo = (ko.grad.double() - k64.grad).abs().max().item() + 1e-30
                worst[0] = max(worst[0], (k1.grad.detach().cpu().double() - k64.grad).abs().max().item() / eo)
                worst[1] = max(worst[1], (k2.grad.detach().cpu().double() - k64.grad).abs().max().item() / eo)
                vs_oracle += 1
            halves += int(half)
        except Exception:
            print("FAILED case:", tag, flush=True)
            raise
        n += 1
    print("fuzz ok: %d random sharded frames (%d also against the CPU oracle, %d with half logits) in %.0f s; largest "
          "d_kernels error vs float64 in units of the oracle's own fp32 error: whole-frame GPU %.2f, sharded GPU %.2f" % (
              n, vs_oracle, halves, time.time() - t0, worst[0], worst[1]))


if __name__ == "__main__":
    main()
