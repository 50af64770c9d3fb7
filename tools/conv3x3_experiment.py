"""The split-precision 3 x 3 convolution against MIOpen's fp32 solver: values (vs float64) and time.

    python tools/conv3x3_experiment.py [--shapes 720p|all] [--reps 10]
"""
import argparse
import ctypes
import os
import sys

import torch as th
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sbmc_amd  # noqa: E402,F401  (installs the MIOpen find records)
from sbmc_amd import _lib  # noqa: E402


def prepare(w, flip=False):
    lib = _lib.lib()
    cout, cin = w.shape[:2]
    if flip:
        cout, cin = cin, cout
    nbytes = lib.sbmc_conv3x3_weights_bytes(cin, cout)
    assert nbytes, (cin, cout)
    wp = th.empty(nbytes, dtype=th.uint8, device=w.device)
    s = w.stride()
    s_co, s_ci = (s[1], s[0]) if flip else (s[0], s[1])
    _lib.check(lib.sbmc_conv3x3_prepare_weights_f32(_lib.ptr(w), s_co, s_ci, s[2], s[3], w.numel(), cin, cout,
                                                     1 if flip else 0, _lib.ptr(wp), _lib.current_stream(th.device("cuda"))), "prepare")
    return wp


def conv(x_nhwc, wp, cout, xmax=None):
    lib = _lib.lib()
    n, h, w, cin = x_nhwc.shape
    if xmax is None:
        xmax = th.empty(1, dtype=th.int32, device=x_nhwc.device)
        _lib.check(lib.sbmc_conv3x3_absmax_f32(_lib.ptr(x_nhwc), x_nhwc.numel(), _lib.ptr(xmax), _lib.current_stream(th.device("cuda"))), "absmax")
    y = th.empty(n, h, w, cout, dtype=th.float32, device=x_nhwc.device)
    _lib.check(lib.sbmc_conv3x3_nhwc_f32(_lib.ptr(x_nhwc), _lib.ptr(xmax), _lib.ptr(wp), _lib.ptr(y), n, h, w, cin, cout,
                                         None, _lib.current_stream(th.device("cuda"))), "conv3x3")
    return y, xmax


def timed(fn, reps):
    for _ in range(2):
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
    ap.add_argument("--shapes", default="720p")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--power", action="store_true", help="also time the forward kernel on all-zero and constant operands")
    args = ap.parse_args()
    dev = th.device("cuda")
    th.manual_seed(1)
    # small: values against float64 (odd sizes: edge tiles)
    for (h, w, cin, cout) in [(37, 53, 32, 128), (16, 16, 64, 128), (40, 70, 128, 256)]:
        x = th.randn(1, cin, h, w, device=dev) * 3.0
        wt = th.randn(cout, cin, 3, 3, device=dev) * 0.05
        ref = F.conv2d(x.double(), wt.double(), padding=1)
        xn = x.permute(0, 2, 3, 1).contiguous()
        y, _ = conv(xn, prepare(wt), cout)
        y = y.permute(0, 3, 1, 2)
        mi = F.conv2d(x.contiguous(memory_format=th.channels_last), wt.contiguous(memory_format=th.channels_last), padding=1)
        scale = ref.abs().max().item()
        print("values %3dx%3d %3d->%3d: ours %.3e  MIOpen fp32 %.3e  (max abs error / max |ref|)" % (
            h, w, cin, cout, (y.double() - ref).abs().max().item() / scale, (mi.double() - ref).abs().max().item() / scale))
        # adjoint weights: gx = conv(gy, flipped transposed w)
        gy = th.randn(1, cout, h, w, device=dev)
        gref = th.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), padding=1)
        if cin % 128 == 0 and cout % 32 == 0:
            gx, _ = conv(gy.permute(0, 2, 3, 1).contiguous(), prepare(wt, flip=True), cin)
            print("   adjoint: %.3e" % ((gx.permute(0, 3, 1, 2).double() - gref).abs().max().item() / gref.abs().max().item()))
    shapes = [(720, 1280, 128, 128)]
    if args.shapes == "all":
        shapes += [(720, 1280, 384, 128), (360, 640, 128, 256), (360, 640, 256, 256), (360, 640, 768, 256),
                   (180, 320, 256, 512), (180, 320, 512, 512)]
    for (h, w, cin, cout) in shapes:
        x = th.randn(1, cin, h, w, device=dev).contiguous(memory_format=th.channels_last)
        wt = (th.randn(cout, cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=th.channels_last)
        xn = x.permute(0, 2, 3, 1)
        assert xn.is_contiguous()
        wp = prepare(wt)
        y, xmax = conv(xn, wp, cout)
        mi = F.conv2d(x, wt, padding=1)
        err = (y.permute(0, 3, 1, 2) - mi).abs().max().item() / mi.abs().max().item()
        t_mi = timed(lambda: F.conv2d(x, wt, padding=1), args.reps)
        t_us = timed(lambda: conv(xn, wp, cout, xmax), args.reps)
        if args.power:
            # the same launch on data that toggles fewer bits (the scale word stays: same instructions, same addresses): what
            # the part's power limit has to do with the time.  One value in the tensor keeps the scale, everything else is zero
            # / a constant.
            for name, fill in (("zeros", 0.0), ("constant 1", 1.0)):
                xz = th.full_like(xn, fill)
                xz.view(-1)[0] = xn.abs().max()
                wz = prepare(th.full_like(wt, fill * 0.05 if fill else 0.0).index_put_((th.tensor(0), th.tensor(0), th.tensor(0), th.tensor(0)), wt.abs().max()))
                tz = timed(lambda: conv(xz, wz, cout, xmax), args.reps)
                print("      %-10s operands: %.3f ms (random: %.3f)" % (name, tz, t_us))
        t_all = timed(lambda: conv(xn, prepare(wt), cout), args.reps)
        gf = 2.0 * h * w * cin * cout * 9 / 1e9
        # weight gradient: ours (C ABI) against MIOpen's
        lib = _lib.lib()
        gy = th.randn(1, cout, h, w, device=dev).contiguous(memory_format=th.channels_last)
        if lib.sbmc_conv3x3_wgrad_supported(1, h, w, cin, cout):
            gmax = th.empty(1, dtype=th.int32, device=dev)
            st = _lib.current_stream(dev)
            _lib.check(lib.sbmc_conv3x3_absmax_f32(_lib.ptr(gy), gy.numel(), _lib.ptr(gmax), st), "absmax")
            gw = th.empty((cout, cin, 3, 3), device=dev).contiguous(memory_format=th.channels_last)
            scratch = th.empty(lib.sbmc_conv3x3_wgrad_scratch_bytes(1, h, w, cin, cout), dtype=th.uint8, device=dev)
            sw = gw.stride()

            def ours_w():
                _lib.check(lib.sbmc_conv3x3_wgrad_f32(_lib.ptr(gy), _lib.ptr(gmax), _lib.ptr(xn), _lib.ptr(xmax), _lib.ptr(gw),
                                                      sw[0], sw[1], sw[2], sw[3], _lib.ptr(scratch), 1, h, w, cin, cout, st), "wgrad")

            def lib_w():
                return th.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                        [False, True, False])[1]
            ours_w()
            ref_w = lib_w()
            werr = (gw - ref_w).abs().max().item() / ref_w.abs().max().item()
            print("      weight gradient: MIOpen %.3f ms  ours %.3f ms  diff %.2e" % (timed(lib_w, args.reps), timed(ours_w, args.reps), werr))
            if args.power:
                keep_gy, keep_x = gy.clone(), xn.clone()
                for name, fill in (("zeros", 0.0), ("constant 1", 1.0)):
                    gy.fill_(fill); xn.fill_(fill)
                    gy[0, 0, 0, 0] = keep_gy.abs().max(); xn[0, 0, 0, 0] = keep_x.abs().max()
                    print("      weight gradient, %-10s operands: %.3f ms" % (name, timed(ours_w, args.reps)))
                gy.copy_(keep_gy); xn.copy_(keep_x)
        print("%4dx%4d %3d->%3d: MIOpen %.3f ms (%.0f TFLOP/s)  ours %.3f ms (%.0f TFLOP/s fp32-equivalent, %.0f f16)  "
              "with absmax+prepare %.3f ms   diff %.2e" % (h, w, cin, cout, t_mi, gf / t_mi, t_us, gf / t_us,
                                                          3 * gf / t_us, t_all, err))


if __name__ == "__main__":
    main()
