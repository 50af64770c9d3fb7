"""Does the row stride of the planar activations hold the 1x1 kernels' memory stream down?  The same 128 -> 128
forward over the same number of pixels, as few large images (row stride 3.7 MB: a 64-pixel tile touches 128 + 128
rows in as many different 2 MB pages) and as many small images (row stride 64 KB: the tile's rows lie within 8 MB).
    gpurun -- python tools/pw_stride_experiment.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import _lib

dev = th.device("cuda")
L = _lib.lib()


def run(b, hw, cin=128, cout=128, n=10):
    x = th.randn(b, cin, hw, device=dev)
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    bias = th.randn(cout, device=dev)
    y = th.empty(b, cout, hw, device=dev)

    def f():
        rc = L.sbmc_pointwise_fwd_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), None, _lib.ptr(y), b, 1, cin, cout, hw,
                                      0, 1, 0.0, _lib.current_stream(dev))
        _lib.check(rc, "fwd")
    for _ in range(3):
        f()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    th.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    gb = 4.0 * b * hw * (cin + cout) / 1e9
    print("B %4d x hw %7d (row stride %8.1f KB): %.3f ms, %.2f TB/s" % (b, hw, hw * 4 / 1024, ms, gb / ms), flush=True)


total = 8 * 1280 * 720
for hw in (1280 * 720, 230400, 57600, 16384, 4096, 1024):
    run(total // hw, hw)
