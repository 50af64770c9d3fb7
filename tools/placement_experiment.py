"""Does WHERE a 13 GB logit tensor's pages land decide the 8-sample splat kernels' launch time?  (tools/roofline_variance.sh found
4.4 / 5.0 / 5.4 ms from run to run, every launch of a run alike.)  One process: the same SplatAll forward + backward on tensors
allocated (1) first thing, (2) after 2 GB holes were punched into 100 GB, (3) after 60 GB of 2 MB blocks were allocated and every
other one freed, (4) again on the tensors of (1).     python tools/placement_experiment.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import functions as F

dev = th.device("cuda")
H, W, S, K = 720, 1280, 8, 21


def make():
    th.cuda.empty_cache()
    rad = th.rand(1, S, 3, H, W, device=dev).requires_grad_()
    log = th.randn(1, S, K * K, H, W, device=dev).requires_grad_()
    return rad, log


def run(tag, rad, log, reps=8):
    dout = th.rand(1, 3, H, W, device=dev)
    store = []
    for i in range(reps + 2):
        rad.grad = None; log.grad = None
        if i == 2:
            F.enable_kernel_timing(store)
        sr, sw, _ = F.SplatAll.apply(rad, log)
        (sr / (sw + 1e-8)).backward(dout)
    th.cuda.synchronize()
    F.enable_kernel_timing(None)
    per = {}
    for n, a, b in store:
        per.setdefault(n, []).append(a.elapsed_time(b))
    print("%-52s fwd %.3f ms  bwd %.3f ms   (logits @ %x, gradient @ %x)" % (
        tag, sum(per["splat_update_fwd_all"]) / reps, sum(per["splat_update_bwd_all"]) / reps, log.data_ptr(), log.grad.data_ptr()), flush=True)


first = make()
run("1. allocated first thing", *first)
run("1b. the same tensors, gradient reallocated", *first)
junk = [th.empty(2 << 30, dtype=th.uint8, device=dev) for _ in range(50)]
del junk[::2]
th.cuda.empty_cache()
second = make()
run("2. after 2 GB holes in 100 GB", *second)
del junk, second
th.cuda.empty_cache()
t0 = time.time()
small = [th.empty(2 << 20, dtype=th.uint8, device=dev) for _ in range(30000)]
del small[::2]
th.cuda.empty_cache()
print("   (60 GB of 2 MB blocks, every other one freed: %.1f s)" % (time.time() - t0), flush=True)
third = make()
run("3. after 2 MB holes in 60 GB", *third)
run("4. the tensors of 1. again", *first)
del small, third
th.cuda.empty_cache()
fourth = make()
run("5. fresh tensors after everything was freed", *fourth)
