"""Why are some regions of device memory slow for the 8-sample splat backward (tools/placement_experiment2.py: the gradient written
13-33 GB into a 44 GB allocation: 5.35 ms, 30 GB in: 4.44)?  (a) a plain linear fill of the same regions -- a stream that misses
no TLB -- and (b) the same backward with logits and gradient in PHYSICALLY CONTIGUOUS allocations
(hipExtMallocWithFlags(hipDeviceMallocContiguous)).     python tools/placement_experiment3.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import functions as F

dev = th.device("cuda")
H, W, S, K = 720, 1280, 8, 21
n = S * K * K * H * W
GB = 1 << 30
pool = th.empty(44 << 30, dtype=th.uint8, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]


class Raw(object):
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def contiguous(count):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), count * 4, 0x4)
    if rc != 0:
        print("hipExtMallocWithFlags(contiguous, %.1f GB) failed: %d" % (count * 4 / GB, rc), flush=True)
        return None
    return th.as_tensor(Raw(p.value, count), device=dev)


def timed_fill(t):
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    t.fill_(1.0); th.cuda.synchronize()
    a.record(); t.fill_(2.0); b.record(); th.cuda.synchronize()
    return a.elapsed_time(b)


for off in (0, 13, 16, 20, 24, 28, 30):
    reg = pool[off * GB: off * GB + 4 * n].view(th.float32)
    print("linear fill of 13 GB at %2d GB of the pool: %.3f ms" % (off, timed_fill(reg)), flush=True)

rad = th.rand(1, S, 3, H, W, device=dev).requires_grad_()
dout = th.rand(1, 3, H, W, device=dev)
orig = th.empty_like


def backward_ms(log, grad):
    log.requires_grad_()
    hits = []

    def patched(t, *a, **k):
        if t.shape == log.shape and t.dtype == th.float32:
            hits.append(1)
            return grad
        return orig(t, *a, **k)
    th.empty_like = patched
    store = []
    for i in range(7):
        rad.grad = None; log.grad = None
        if i == 2:
            F.enable_kernel_timing(store)
        sr, sw, _ = F.SplatAll.apply(rad, log)
        (sr / (sw + 1e-8)).backward(dout)
    th.cuda.synchronize()
    F.enable_kernel_timing(None)
    th.empty_like = orig
    assert len(hits) == 7
    f = [a.elapsed_time(b) for nme, a, b in store if nme == "splat_update_fwd_all"]
    t = [a.elapsed_time(b) for nme, a, b in store if nme == "splat_update_bwd_all"]
    return sum(f) / len(f), sum(t) / len(t)


shape = (1, S, K * K, H, W)
pl = pool[:4 * n].view(th.float32).view(shape); pl.normal_()
for off in (14, 30):
    pg = pool[off * GB: off * GB + 4 * n].view(th.float32).view(shape)
    print("pool: logits at 0, gradient at %d GB: forward %.3f ms, backward %.3f ms" % ((off,) + backward_ms(pl.detach(), pg)), flush=True)
cl, cg = contiguous(n), contiguous(n)
if cl is not None and cg is not None:
    cl = cl.view(shape); cl.normal_()
    print("contiguous allocations (logits @ %x, gradient @ %x): forward %.3f ms, backward %.3f ms" % (
        (cl.data_ptr(), cg.data_ptr()) + backward_ms(cl, cg.view(shape))), flush=True)
    print("pool logits, contiguous gradient: forward %.3f ms, backward %.3f ms" % backward_ms(pl.detach(), cg.view(shape)), flush=True)
    pg = pool[14 * GB: 14 * GB + 4 * n].view(th.float32).view(shape)
    print("contiguous logits, pool gradient at 14 GB: forward %.3f ms, backward %.3f ms" % backward_ms(cl.detach(), pg), flush=True)
