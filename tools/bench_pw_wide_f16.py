"""Dev tool: the all-half 441-channel forward, wide kernel (one workgroup walks the row tiles) vs one workgroup per row tile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import functions as F
dev = th.device("cuda"); hw = 1280 * 720
for B in (8, 32):
    th.manual_seed(0)
    x = th.randn(B, 128, hw, device=dev).half(); w = th.randn(441, 128, device=dev) / 11.3; b = th.randn(441, device=dev)
    res = {}
    for knob in ("1", "0"):
        os.environ["SBMC_HIP_PW_FWD_WIDE"] = knob
        for _ in range(3):
            y = F.pointwise_half(x, w, b, None, 1, 0, 0.0)
        th.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            y = F.pointwise_half(x, w, b, None, 1, 0, 0.0)
        th.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
        res[knob] = y.clone()
        gb = B * (128 + 441) * hw * 2 / 1e9
        print("B%d 128->441 half, wide=%s: %.3f ms  %.0f GB/s" % (B, knob, ms, gb / ms * 1e3), flush=True)
    print("   equal to the bit:", th.equal(res["1"], res["0"]))
    del x, y, res; th.cuda.empty_cache()
