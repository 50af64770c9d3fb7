import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch as th
from sbmc_amd import functions as funcs
dev = th.device("cuda")
x = th.randn(8, 128, 1280 * 720, device=dev); funcs.ensure_amax(x)
w = th.randn(441, 128, device=dev) / 128 ** 0.5; b = th.randn(441, device=dev)
def t(n=6):
    for _ in range(2): funcs.PointwiseLayer.apply(x, w, b, None, 1, 0, 0.0)
    th.cuda.synchronize(); a, c = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): funcs.PointwiseLayer.apply(x, w, b, None, 1, 0, 0.0)
    c.record(); th.cuda.synchronize(); return a.elapsed_time(c) / n
with th.no_grad():
    for knob in ("1", "0"):
        os.environ["SBMC_PW_WIDE_FWD"] = knob
        ms = t(); print("SBMC_PW_WIDE_FWD=%s: %.3f ms  %.2f TB/s" % (knob, ms, 4.0 * 8 * 921600 * (128 + 441) / ms / 1e9))
