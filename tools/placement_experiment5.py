"""The 8-sample splat FORWARD through the C ABI with every tensor at a chosen offset of ONE allocation: logits at 0, radiance behind
them, the seven output planes per sample (part_r x3, part_w, part_m, kmax, atap) in one block whose START is swept.  Which
offsets are slow?     python tools/placement_experiment5.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import _lib

dev = th.device("cuda")
H, W, S, K = 720, 1280, 8, 21
hw = H * W
MB, GB = 1 << 20, 1 << 30
pool = th.empty(40 * GB, dtype=th.uint8, device=dev)
L = _lib.lib()


def carve(off, count, dtype=th.float32):
    return pool[off:off + 4 * count].view(dtype)


log = carve(0, S * K * K * hw); log.normal_()
rad_off = 14 * GB
rad = carve(rad_off, S * 3 * hw); rad.uniform_()
stream = _lib.current_stream(dev)


def fwd_ms(out_off, gap=0, reps=4):
    """outputs from out_off on: part_r, part_w, part_m, kmax, atap, `gap` bytes between them"""
    o = out_off
    ts = []
    for cnt in (S * 3 * hw, S * hw, S * hw, S * hw, S * hw):
        ts.append(o)
        o += 4 * cnt + gap
    part_r, part_w, part_m, kmax = (carve(ts[i], c) for i, c in enumerate((S * 3 * hw, S * hw, S * hw, S * hw)))
    atap = carve(ts[4], S * hw, th.int32)
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    tot = 0.0
    for i in range(reps + 1):
        a.record()
        rc = L.sbmc_splat_update_fwd_f32(_lib.ptr(rad), _lib.ptr(log), None, None, None, _lib.ptr(part_r), _lib.ptr(part_w), _lib.ptr(part_m),
                                         _lib.ptr(kmax), _lib.ptr(atap), S, 3, H, W, K, stream)
        b.record()
        assert rc == 0, rc
        th.cuda.synchronize()
        if i:
            tot += a.elapsed_time(b)
    return tot / reps


base = 16 * GB
print("outputs' block start swept (logits at 0, radiance at 14 GB):")
for d in (0, 256, 1024, 4096, 16384, 65536, 262144, MB, 2 * MB, 3 * MB, 4 * MB, 8 * MB, 16 * MB, 64 * MB, 256 * MB, GB, 2 * GB, 4 * GB, 8 * GB):
    print("  16 GB + %11d B: %.3f ms" % (d, fwd_ms(base + d)), flush=True)
print("gap between the five output tensors swept (block at 16 GB):")
for g in (0, 256, 4096, 65536, MB, 2 * MB + 4096, 16 * MB):
    print("  gap %9d B: %.3f ms" % (g, fwd_ms(base, g)), flush=True)
print("repeat of the first: %.3f ms" % fwd_ms(base), flush=True)
