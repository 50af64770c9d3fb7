"""Randomised parity sweep of the device kernels against the CPU oracle (run on the GPU box):
random batch / channels / frame sizes / kernel sizes / sample counts, forward and backward, per
sample (ProgressiveKernelApply) and all samples per launch (SplatAll), plus the boundary ops.

    python tools/fuzz_gpu.py [--seconds 240] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import close, close_or_yardstick, progressive_fp64, run_progressive  # noqa: E402
from oracle import sbmc_oracle as orc  # noqa: E402
from sbmc_amd import functions as F, modules  # noqa: E402


# d_kernels gets, at each destination's arg-max tap, the routed gradient of the running max
# dM - (dR.sum_r + dW*sum_w): a difference of 441-term fp32 sums, so that one element inherits
# their rounding (1.5e-5 of max|d_kernels| was observed on a 1x4 frame with k=21, where nearly
# every tap is the zero-filled border).  Everything is held to 1e-5 of the oracle; where a d_kernels
# tensor misses that, it must be no further from the float64 evaluation of the chain than twice the
# oracle's own fp32 result is (helpers.close_or_yardstick).
DK_RTOL = 1e-5
YARDSTICK = [0]


def dk_close(got, ref32, s, datas, kerns, grads, splat):
    def truth():
        _, _, dk64 = progressive_fp64(datas, kerns, grads, splat=splat)
        return dk64[s]
    if close_or_yardstick(got, ref32, truth, rtol=DK_RTOL, slack=2.0, what="d_kernels[%d]" % s) is not None:
        YARDSTICK[0] += 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    t0, n = time.time(), 0
    stats = {}
    while time.time() - t0 < args.seconds:
        k = int(rng.choice([1, 3, 5, 7, 9, 21, 21, 21]))
        bs = int(rng.choice([1, 1, 2, 3]))
        c = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 8]))
        h = int(rng.randint(1, 40))
        w = int(rng.choice([rng.randint(1, 30), rng.randint(60, 70), rng.randint(120, 200), rng.randint(250, 330)]))
        spp = int(rng.randint(1, 4))
        scale = float(rng.choice([0.1, 1.0, 3.0, 10.0]))
        th.manual_seed(int(rng.randint(1 << 30)))
        datas = [th.randn(bs, c, h, w) for _ in range(spp)]
        kerns = [th.randn(bs, k * k, h, w) * scale for _ in range(spp)]
        grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]
        tag = "k%d c%d" % (k, c)
        try:
            ref_out, ref_dd, ref_dk = run_progressive(
                lambda d, kk, a, b, m: orc.progressive_kernel_apply(d, kk, a, b, m, splat=True),
                datas, kerns, grads, "cpu")
            out, dd, dk = run_progressive(modules.ProgressiveKernelApply(splat=True), datas, kerns, grads, "cuda")
            for a, b in zip(out, ref_out):
                close(a, b)
            for s in range(spp):
                close(dd[s], ref_dd[s]); dk_close(dk[s], ref_dk[s], s, datas, kerns, grads, True)
            # gather-kernel (splat=False) update, fused where the strip kernels apply
            g_ref, g_dd, g_dk = run_progressive(
                lambda d, kk, a, b, m: orc.progressive_kernel_apply(d, kk, a, b, m, splat=False),
                datas, kerns, grads, "cpu")
            g_out, gdd, gdk = run_progressive(modules.ProgressiveKernelApply(splat=False), datas, kerns, grads, "cuda")
            for a, b in zip(g_out, g_ref):
                close(a, b)
            for s in range(spp):
                close(gdd[s], g_dd[s]); dk_close(gdk[s], g_dk[s], s, datas, kerns, grads, False)
            dg = th.stack(datas, 1).cuda().requires_grad_()
            kg = th.stack(kerns, 1).cuda().requires_grad_()
            if F.splat_all_supported(dg, kg):
                res = F.SplatAll.apply(dg, kg)
                th.autograd.backward(res, [g.cuda() for g in grads])
                for a, b in zip(res, ref_out):
                    close(a, b)
                for s in range(spp):
                    close(dg.grad[:, s], ref_dd[s]); dk_close(kg.grad[:, s], ref_dk[s], s, datas, kerns, grads, True)
                tag += " all"
            # boundary ops on the first sample
            x5 = kerns[0].view(bs, k, k, h, w)
            assert th.equal(F.Scatter2Gather.apply(x5.cuda()).cpu(), orc.Scatter2Gather.apply(x5))
            d0 = datas[0].clone().requires_grad_()
            w0 = x5.clone().requires_grad_()
            o_ref, s_ref = orc.KernelWeighting.apply(d0, w0)
            th.autograd.backward([o_ref, s_ref], [grads[0], grads[1][:, 0]])
            d1 = datas[0].cuda().requires_grad_()
            w1 = x5.cuda().requires_grad_()
            o, s_ = F.KernelWeighting.apply(d1, w1)
            th.autograd.backward([o, s_], [grads[0].cuda(), grads[1][:, 0].cuda()])
            close(o, o_ref); close(s_, s_ref); close(d1.grad, d0.grad); close(w1.grad, w0.grad)
        except Exception:
            print("FAILED case: bs=%d c=%d h=%d w=%d k=%d spp=%d scale=%g" % (bs, c, h, w, k, spp, scale))
            raise
        stats[tag] = stats.get(tag, 0) + 1
        n += 1
    print("fuzz ok: %d random cases in %.0f s (everything within 1e-5 of the oracle, except %d d_kernels tensors held to "
          "the float64 yardstick instead: no worse than 2x the oracle's own error)" % (n, time.time() - t0, YARDSTICK[0]))
    print(" ".join("%s:%d" % kv for kv in sorted(stats.items())))


if __name__ == "__main__":
    main()
