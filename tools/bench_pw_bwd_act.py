"""Dev tool: what the y (activation mask) stream costs the fused 1x1 backward: act = 0 (no y read) vs act = 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import functions as funcs
dev = th.device("cuda")
hw = 1280 * 720
def timeit(fn, n=10):
    for _ in range(3): fn()
    th.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for act in (0, 1, 0, 1):
    x = th.randn(8, 128, hw, device=dev, requires_grad=True)
    w = (th.randn(128, 128, device=dev) / 128 ** 0.5).requires_grad_()
    bias = th.randn(128, device=dev, requires_grad=True)
    y = funcs.PointwiseLayer.apply(x, w, bias, None, 1, act, 0.0)
    g = th.randn_like(y)
    tb = timeit(lambda: th.autograd.grad(y, [w, bias, x], g, retain_graph=True))
    print("bwd 128x128 act %d: %.3f ms" % (act, tb), flush=True)
    del x, y, g
