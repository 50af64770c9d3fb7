"""Is it the SMALL tensors?  The same SplatAll forward + backward with every tensor of the process carved out of ONE device allocation
(a bump allocator plugged into torch: tools/dev/arena_alloc.cpp; ARENA_CONTIGUOUS=1: physically contiguous) against torch's caching
allocator.     python tools/placement_experiment4.py [arena]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
if len(sys.argv) > 1 and sys.argv[1] == "arena":
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "arena_alloc.so")
    th.cuda.memory.change_current_allocator(th.cuda.memory.CUDAPluggableAllocator(so, "arena_malloc", "arena_free"))
from sbmc_amd import functions as F

dev = th.device("cuda")
H, W, S, K = 720, 1280, 8, 21
rad = th.rand(1, S, 3, H, W, device=dev).requires_grad_()
log = th.randn(1, S, K * K, H, W, device=dev).requires_grad_()
dout = th.rand(1, 3, H, W, device=dev)
store = []
reps = 4
for i in range(reps + 2):
    rad.grad = None; log.grad = None
    if i == 2:
        F.enable_kernel_timing(store)
    sr, sw, _ = F.SplatAll.apply(rad, log)
    (sr / (sw + 1e-8)).backward(dout)
th.cuda.synchronize()
per = {}
for n, a, b in store:
    per.setdefault(n, []).append(a.elapsed_time(b))
print("%-28s fwd %s   bwd %s" % (" ".join(sys.argv[1:]) + " contiguous=" + os.environ.get("ARENA_CONTIGUOUS", "-") if len(sys.argv) > 1 else "caching allocator",
                                 " ".join("%.3f" % t for t in per["splat_update_fwd_all"]), " ".join("%.3f" % t for t in per["splat_update_bwd_all"])), flush=True)
