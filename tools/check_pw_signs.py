"""Dev tool: the sign words pw_fwd_s_kernel leaves ([B, Cout, ceil(hw / 32)], bit px % 32) against the output itself."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import _lib
dev = th.device("cuda")
L = _lib.lib()
for (b, s, cin, cout, hw, t_mode) in [(2, 1, 128, 128, 1024, 0), (3, 3, 32, 32, 720, 2), (4, 2, 93, 128, 516, 1), (3, 1, 32, 25, 100, 0)]:
    th.manual_seed(1)
    x = th.randn(b, cin, hw, device=dev)
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    bias = th.randn(cout, device=dev)
    t = None
    if t_mode == 1:
        t = th.randn(b // s, cout, device=dev)
    elif t_mode == 2:
        t = th.randn(b // s, cout, hw, device=dev)
    y = th.empty(b, cout, hw, device=dev)
    wpr = (hw + 31) // 32
    signs = th.full((b, cout, wpr), 0x55555555, dtype=th.int32, device=dev)
    rc = L.sbmc_pointwise_fwd_signs_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(t) if t is not None else None,
                                        _lib.ptr(y), _lib.ptr(signs), b, s, cin, cout, hw, t_mode, 2, 0.01, _lib.current_stream(dev))
    _lib.check(rc, "fwd_signs")
    bits = (signs.unsqueeze(-1) >> th.arange(32, device=dev, dtype=th.int32)) & 1
    bits = bits.reshape(b, cout, wpr * 32)[..., :hw].bool()
    want = y > 0
    bad = (bits != want)
    print("b%d s%d cin%d cout%d hw%d t%d: %d / %d sign bits differ" % (b, s, cin, cout, hw, t_mode, int(bad.sum()), bad.numel()))
    if bad.any():
        idx = bad.nonzero()[:8].tolist()
        print("   first:", idx)
        rows = sorted(set(i[1] for i in bad.nonzero().tolist()))
        print("   rows:", rows[:40])
