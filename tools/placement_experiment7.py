"""SHOW what differs between a fast and a slow placement of the logits.  The 8-sample forward strip kernel on
logits at 0 / 14 / 28 / 42 GB of one 60 GB allocation (placement_experiment6: the mode is a property of the region), 6 launches
each -- plain for the times, and under `rocprofv3 --pmc` for the per-launch counters (tools/placement_pmc.sh).
    python tools/placement_experiment7.py"""
import os, sys
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
N = 6
for off in (0, 14, 28, 42):
    log = carve(off * GB, S * K * K * hw); log.normal_()
    th.cuda.synchronize()
    ev = [th.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    for i in range(N):
        ev[i].record()
        assert L.sbmc_splat_update_fwd_f32(_lib.ptr(rad), _lib.ptr(log), None, None, None, _lib.ptr(outs[0]), _lib.ptr(outs[1]),
                                           _lib.ptr(outs[2]), _lib.ptr(outs[3]), _lib.ptr(atap), S, 3, H, W, K, stream) == 0
    ev[N].record()
    th.cuda.synchronize()
    print("logits at %2d GB: %s ms" % (off, " ".join("%.3f" % ev[i].elapsed_time(ev[i + 1]) for i in range(N))), flush=True)
