"""Is the mode of the 8-sample forward a property of TIME rather than of addresses?  Fixed tensors, 300 launches back to back, every
launch's time; a pause; again -- and the same with the logits at another offset of the allocation.     python tools/placement_experiment6.py"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import _lib

dev = th.device("cuda")
H, W, S, K = 720, 1280, 8, 21
hw = H * W
GB = 1 << 30
pool = th.empty(60 * GB, dtype=th.uint8, device=dev)
L = _lib.lib()
carve = lambda off, count, dtype=th.float32: pool[off:off + 4 * count].view(dtype)
rad = carve(56 * GB, S * 3 * hw); rad.uniform_()
outs = [carve(57 * GB + i * (1 << 28), c) for i, c in enumerate((S * 3 * hw, S * hw, S * hw, S * hw))]
atap = carve(59 * GB, S * hw, th.int32)
stream = _lib.current_stream(dev)


def series(log, n):
    ev = [th.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        ev[i].record()
        rc = L.sbmc_splat_update_fwd_f32(_lib.ptr(rad), _lib.ptr(log), None, None, None, _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]),
                                         _lib.ptr(outs[3]), _lib.ptr(atap), S, 3, H, W, K, stream)
        assert rc == 0
    ev[n].record()
    th.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]


def show(tag, t):
    fast = sum(1 for x in t if x < 2.45)
    print("%-34s %3d launches: min %.3f max %.3f, %3d below 2.45 ms; first 12: %s ... last 6: %s" % (
        tag, len(t), min(t), max(t), fast, " ".join("%.2f" % x for x in t[:12]), " ".join("%.2f" % x for x in t[-6:])), flush=True)


for off in (0, 14, 28, 42):
    log = carve(off * GB, S * K * K * hw); log.normal_()
    th.cuda.synchronize()
    show("logits at %2d GB, after a sync" % off, series(log, 300))
    time.sleep(3.0)
    show("logits at %2d GB, after 3 s idle" % off, series(log, 300))
try:
    print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout[-900:])
except Exception as e:       # noqa: BLE001
    print("rocm-smi:", e)
