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

PEAK = 8000.0          # GB/s, HBM3E spec
MFMA_PEAK = 157.3      # TFLOP/s, dense fp32 MFMA


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
        del data, wts, go, gs, out, sw, dd, dw, s2g, kern, kg, dg, st, gst
        th.cuda.empty_cache()
    cnn_glue(dev, args.reps)


def cnn_glue(dev, reps):
    """The kernels around the kernel-predicting CNN at 1280x720, 8 samples: fused 1x1 layers (fp32
    MFMA; flop = 2*cin*cout per output pixel and product), bias/activation passes and the U-net's
    upsample + concat (HBM-bound)."""
    hw, B = 1280 * 720, 8
    label = "720p, 8 samples"

    def emit(name, fn, flop=None, nbytes=None):
        ms = timeit(fn, reps)
        rec = {"op": name, "shape": label, "ms": round(ms, 4)}
        if flop is not None:
            rec.update(TFLOPs=round(flop / (ms * 1e-3) / 1e12, 1),
                       frac_of_fp32_mfma_peak=round(flop / (ms * 1e-3) / 1e12 / MFMA_PEAK, 4))
        if nbytes is not None:
            rec.update(alg_bytes=nbytes, GBps=round(nbytes / (ms * 1e-3) / 1e9, 1),
                       roofline_frac=round(nbytes / (ms * 1e-3) / 1e9 / PEAK, 4))
        print(json.dumps(rec))

    for cin, cout, act in ((128, 128, 1), (128, 441, 0)):
        x = th.randn(B, cin, hw, device=dev, requires_grad=True)
        w = (th.randn(cout, cin, device=dev) / cin ** 0.5).requires_grad_()
        bias = th.randn(cout, device=dev, requires_grad=True)
        emit("pointwise layer %d->%d fwd (GEMM + bias + act)" % (cin, cout),
             lambda: F.PointwiseLayer.apply(x.detach(), w.detach(), bias.detach(), None, 1, act, 0.0),
             flop=2.0 * cin * cout * B * hw, nbytes=4.0 * B * hw * (cin + cout))
        y = F.PointwiseLayer.apply(x, w, bias, None, 1, act, 0.0)
        g = th.randn_like(y)
        emit("pointwise layer %d->%d bwd (gx + gw + gbias)" % (cin, cout),
             lambda: th.autograd.grad(y, [x, w, bias], g, retain_graph=True),
             flop=4.0 * cin * cout * B * hw, nbytes=4.0 * B * hw * (2 * cin + (2 if act else 1) * cout))
        del x, y, g
        th.cuda.empty_cache()
    y0 = th.randn(1, 128, 720, 1280, device=dev)
    bias = th.randn(128, device=dev, requires_grad=True)
    emit("bias_act fwd [1,128,720,1280] (in place)", lambda: F.BiasAct.apply(y0, bias.detach(), 1, 0.0),
         nbytes=8.0 * y0.numel())
    yy = F.BiasAct.apply(y0.clone().requires_grad_() * 1.0, bias, 1, 0.0)
    g = th.randn_like(yy)
    emit("bias_act bwd [1,128,720,1280]", lambda: th.autograd.grad(yy, [bias], g, retain_graph=True),
         nbytes=12.0 * y0.numel())
    coarse = th.randn(1, 256, 360, 640, device=dev, requires_grad=True)
    left = th.randn(1, 128, 720, 1280, device=dev, requires_grad=True)
    emit("upsample x2 + concat fwd (256 + 128 channels at 720p)",
         lambda: F.UpsampleCat.apply(coarse.detach(), left.detach()),
         nbytes=4.0 * (coarse.numel() + 2 * left.numel() + 4 * coarse.numel()))
    out = F.UpsampleCat.apply(coarse, left)
    g = th.randn_like(out)
    emit("upsample x2 + concat bwd (gather adjoint)",
         lambda: th.autograd.grad(out, [coarse], g, retain_graph=True),
         nbytes=4.0 * (4 * coarse.numel() + coarse.numel()))


if __name__ == "__main__":
    main()
