"""The 8-sample splat BACKWARD reads logit plane (s, t) and writes gradient plane (s, t): two streams a fixed distance apart.
Does that DISTANCE decide the launch time?  One 44 GB allocation made first thing; the logits at its start, the gradient at
chosen distances behind them (th.empty_like patched for that one tensor).     python tools/placement_experiment2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import functions as F

dev = th.device("cuda")
H, W, S, K = 720, 1280, 8, 21
n = S * K * K * H * W
pool = th.empty(44 << 30, dtype=th.uint8, device=dev)
log = pool[:4 * n].view(th.float32).view(1, S, K * K, H, W)
log.normal_()
log.requires_grad_()
rad = th.rand(1, S, 3, H, W, device=dev).requires_grad_()
dout = th.rand(1, 3, H, W, device=dev)
orig = th.empty_like
MB, GB = 1 << 20, 1 << 30
base = (4 * n + 2 * MB - 1) // (2 * MB) * (2 * MB)
for name, diff in [("adjacent (size rounded to 2 MB)", base), ("+ 4 KB", base + 4096), ("+ 64 KB", base + 65536), ("+ 1 MB", base + MB),
                   ("+ 2 MB", base + 2 * MB), ("+ 16 MB", base + 16 * MB), ("13.5 GB", 13 * GB + GB // 2), ("14 GB", 14 * GB),
                   ("16 GB", 16 * GB), ("16 GB + 4 KB", 16 * GB + 4096), ("16 GB + 64 KB", 16 * GB + 65536), ("16 GB + 1 MB", 16 * GB + MB),
                   ("16 GB + 32 MB", 16 * GB + 32 * MB), ("20 GB", 20 * GB), ("24 GB", 24 * GB), ("28 GB", 28 * GB), ("30 GB - 2 MB", 30 * GB - 2 * MB)]:
    grad = pool[diff:diff + 4 * n].view(th.float32).view(1, S, K * K, H, W)

    def patched(t, *a, **k):
        if t.shape == log.shape and t.dtype == th.float32:
            hits.append(1)
            return grad
        return orig(t, *a, **k)
    th.empty_like = patched
    store, hits = [], []
    for i in range(7):
        rad.grad = None; log.grad = None
        if i == 2:
            F.enable_kernel_timing(store)
        sr, sw, _ = F.SplatAll.apply(rad, log)
        (sr / (sw + 1e-8)).backward(dout)
    th.cuda.synchronize()
    F.enable_kernel_timing(None)
    th.empty_like = orig
    assert len(hits) == 7, hits          # (the backward's d_kernels; autograd then copies it into the leaf's .grad, outside the timed call)
    t = [a.elapsed_time(b) for nme, a, b in store if nme == "splat_update_bwd_all"]
    print("gradient %-32s behind the logits (%14d B, mod 4 GB = %10d): backward %.3f ms" % (name, diff, diff % (4 * GB), sum(t) / len(t)), flush=True)
