#!/usr/bin/env python
"""Micro-benchmark of the boundary-level operators and the fused splat update (counterpart of the
reference's scripts/profile/kernel_weighting.py:29-59 and scatter2gather.py:29-57, which only
print a profiler table).  Prints one JSON line per (operator, shape) with the time, the
algorithmic HBM bytes and the fraction of the 8 TB/s roofline.

    python scripts/bench_ops.py [--reps 20]
"""
import argparse
import json
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbmc_amd import functions as F, modules  # noqa: E402

PEAK = 8000.0


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = th.device("cuda")
    shapes = [("profile-script shape (bs4,c3,128x128,k21)", 4, 3, 128, 128, 21),
              ("720p (bs1,c3,1280x720,k21)", 1, 3, 720, 1280, 21)]
    for label, bs, c, h, w, k in shapes:
        px = bs * h * w
        data = th.rand(bs, c, h, w, device=dev)
        wts = th.randn(bs, k, k, h, w, device=dev)
        go = th.randn(bs, c, h, w, device=dev)
        gs = th.randn(bs, h, w, device=dev)
        out, sw = th.empty_like(data), th.empty(bs, h, w, device=dev)
        dd, dw, s2g = th.empty_like(data), th.empty_like(wts), th.empty_like(wts)
        from sbmc_amd import halide_ops as ops
        rows = [
            ("scatter2gather", lambda: ops.scatter2gather_cuda_float32(wts, s2g), 8 * k * k),
            ("kernel_weighting fwd", lambda: ops.kernel_weighting_cuda_float32(data, wts, out, sw),
             4 * k * k + 4 * (2 * c + 1)),
            ("kernel_weighting bwd", lambda: ops.kernel_weighting_grad_cuda_float32(
                data, wts, sw, go, gs, dd, dw), 8 * k * k + 4 * (4 * c + 1)),
        ]
        upd = modules.ProgressiveKernelApply(splat=True)
        kern = wts.view(bs, k * k, h, w)
        st = upd(data, kern, None, None, None)
        st = tuple(t.detach() for t in st)
        rows.append(("fused splat update fwd (with running state)",
                     lambda: upd(data, kern, *st), 4 * k * k + 4 * c + 8 * (c + 2)))
        kg = kern.clone().requires_grad_()
        dg = data.clone().requires_grad_()

        def fused_fb():
            kg.grad = None
            dg.grad = None
            r = upd(dg, kg, *st)
            th.autograd.backward(r, [go, gs.unsqueeze(1), gs.unsqueeze(1)])
        rows.append(("fused splat update fwd+bwd", fused_fb, 12 * k * k + 116))
        gupd = modules.ProgressiveKernelApply(splat=False)
        gst = tuple(t.detach() for t in gupd(data, kern, None, None, None))

        def gather_fb():
            kg.grad = None
            dg.grad = None
            r = gupd(dg, kg, *gst)
            th.autograd.backward(r, [go, gs.unsqueeze(1), gs.unsqueeze(1)])
        rows.append(("fused gather update fwd+bwd (splat=False)", gather_fb, 16 * k * k + 116))
        for name, fn, bpp in rows:
            ms = timeit(fn, args.reps)
            gbps = px * bpp / (ms * 1e-3) / 1e9
            print(json.dumps({"op": name, "shape": label, "ms": round(ms, 4),
                              "alg_bytes": px * bpp, "GBps": round(gbps, 1),
                              "roofline_frac": round(gbps / PEAK, 4)}))


if __name__ == "__main__":
    main()
