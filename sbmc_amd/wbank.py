"""Weight bank: what a step derives from its weight-normalised convolution weights, for many layers per launch.

The reference wraps every convolution of its ConvChains in torch's (old-style) weight norm (sbmc/modules.py:85-94,
178-188): ``w = g v / ||v||`` with the norm taken over each output channel's cin x kh x kw block.  Done layer by layer
that is one ``_weight_norm`` kernel per layer in the forward and one in the backward -- and, for the 3 x 3 layers
that run on csrc/conv3x3.hip, an absmax pass and a preparation pass (the two f16 planes in the kernel's stage
order) per direction: up to 8 launches for each of the 57 layers of ``Multisteps``.  None of that work shrinks with
the slab when a frame is sharded over several GPUs, so it bounds the strong scaling (tools/rank_cost.py).

A ``WeightBank`` does it for up to 24 layers per launch (include/sbmc_hip.h ``sbmc_wbank_*``): two launches in
the forward (rows: norm, w and a row's largest magnitude; prepare: both orientations of every 3 x 3 layer), one in
the backward (the weight norm's adjoint for every layer at once, run when the last layer's gradient has arrived).
Same arithmetic as ``torch._weight_norm`` up to the order of the sum of squares.

No fallback: a bank is only built over layers it can take (``WeightBank.takes``); everything else keeps
``torch._weight_norm``.
"""
import ctypes
import os

import torch as th

from . import _lib
from .utils import knob


def _conv3x3_enabled():
    return knob("SBMC_CONV3X3") != 0


class _BankFn(th.autograd.Function):
    """(v_0, g_0, v_1, g_1, ...) -> (w_0, w_1, ...), views of one flat buffer."""

    @staticmethod
    def forward(ctx, bank, *vg):
        L = _lib.lib()
        dev = vg[0].device
        n = len(vg) // 2
        shapes = [vg[2 * i].shape for i in range(n)]
        numel = [vg[2 * i].numel() for i in range(n)]
        flat = th.empty(sum(numel), dtype=th.float32, device=dev)
        norms = th.empty(2 * sum(s[0] for s in shapes), dtype=th.float32, device=dev)
        # prepared weights of the 3 x 3 layers csrc/conv3x3.hip takes: [forward | adjoint] per layer, 16-byte aligned
        prep = []
        total = 0
        for s in shapes:
            nb = 0
            if bank.prepare and len(s) == 4 and s[2] == 3 and s[3] == 3:
                a, b = L.sbmc_conv3x3_weights_bytes(s[1], s[0]), L.sbmc_conv3x3_weights_bytes(s[0], s[1])
                if a and b:
                    nb = (a, b)
            prep.append((total, nb))
            if nb:
                total += nb[0] + nb[1]
        planes = th.empty(total, dtype=th.uint8, device=dev) if total else None
        entries = (_lib.WBankEntry * n)()
        ws, wps = [], []
        off = noff = 0
        for i in range(n):
            v, g = vg[2 * i], vg[2 * i + 1]
            s = shapes[i]
            e = entries[i]
            e.v, e.g = v.data_ptr(), g.data_ptr()
            e.w = flat.data_ptr() + 4 * off
            e.norm = norms.data_ptr() + 4 * noff
            e.cout, e.cin, e.kh, e.kw = s[0], s[1], s[2], s[3]
            ws.append(flat[off:off + numel[i]].view(s))
            p0, nb = prep[i]
            if nb:
                fwd, bwd = planes[p0:p0 + nb[0]], planes[p0 + nb[0]:p0 + nb[0] + nb[1]]
                e.wp_fwd, e.wp_bwd = fwd.data_ptr(), bwd.data_ptr()
                wps.append((fwd, bwd))
            else:
                e.wp_fwd = e.wp_bwd = None
                wps.append(None)
            off += numel[i]
            noff += 2 * s[0]
        with th.cuda.device(dev):
            for i0 in range(0, n, _lib.WBANK_MAX):
                cnt = min(_lib.WBANK_MAX, n - i0)
                first = ctypes.c_void_p(ctypes.addressof(entries) + i0 * ctypes.sizeof(_lib.WBankEntry))
                _lib.check(L.sbmc_wbank_forward_f32(first, cnt, _lib.current_stream(dev)), "wbank_forward")
        bank._prepared = wps
        ctx.save_for_backward(norms, *vg)
        ctx.shapes = shapes
        return tuple(ws)

    @staticmethod
    def backward(ctx, *gws):
        L = _lib.lib()
        norms, vg = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        shapes = ctx.shapes
        n = len(shapes)
        dev = norms.device
        gflat = th.empty(sum(vg[2 * i].numel() for i in range(n)), dtype=th.float32, device=dev)
        ggflat = th.empty(sum(s[0] for s in shapes), dtype=th.float32, device=dev)
        entries = (_lib.WBankGrad * n)()
        keep, out = [], []
        off = noff = goff = 0
        for i in range(n):
            v, g = vg[2 * i], vg[2 * i + 1]
            s = shapes[i]
            gw = gws[i]
            e = entries[i]
            if gw is not None:
                if gw.dtype != th.float32:
                    gw = gw.float()
                keep.append(gw)
                e.gw = gw.data_ptr()
                e.s_co, e.s_ci, e.s_ky, e.s_kx = gw.stride()
            else:
                e.gw = None
            e.v, e.g = v.data_ptr(), g.data_ptr()
            e.norm = norms.data_ptr() + 4 * noff
            e.gv = gflat.data_ptr() + 4 * off
            e.gg = ggflat.data_ptr() + 4 * goff
            e.cout, e.cin, e.kh, e.kw = s[0], s[1], s[2], s[3]
            out.append(gflat[off:off + v.numel()].view(s))
            out.append(ggflat[goff:goff + s[0]].view(g.shape))
            off += v.numel()
            noff += 2 * s[0]
            goff += s[0]
        with th.cuda.device(dev):
            for i0 in range(0, n, _lib.WBANK_MAX):
                cnt = min(_lib.WBANK_MAX, n - i0)
                first = ctypes.c_void_p(ctypes.addressof(entries) + i0 * ctypes.sizeof(_lib.WBankGrad))
                _lib.check(L.sbmc_wbank_backward_f32(first, cnt, _lib.current_stream(dev)), "wbank_backward")
        return (None,) + tuple(out)


class WeightBank(object):
    """The weight-normalised convolutions `convs` (each with `weight_v`, `weight_g`) as one unit."""

    def __init__(self, convs, prepare=True):
        self.convs = list(convs)
        self.prepare = bool(prepare)
        self._prepared = None

    @staticmethod
    def takes(conv):
        """Old-style weight norm over dim 0 (what the reference's ConvChain registers), fp32 parameters on a GPU."""
        v, g = getattr(conv, "weight_v", None), getattr(conv, "weight_g", None)
        return (isinstance(conv, th.nn.Conv2d) and v is not None and g is not None and v.is_cuda
                and v.dtype == th.float32 and g.dtype == th.float32 and v.dim() == 4 and v.is_contiguous()
                and g.is_contiguous() and g.numel() == v.shape[0] and v.data_ptr() % 16 == 0)

    def weights(self):
        """[w of every layer]; where csrc/conv3x3.hip takes the layer, its prepared forms ride on the tensor
        (`w._sbmc_wp = (forward, adjoint)`: functions.Conv3x3NHWC then skips its own absmax + preparation)."""
        if not self.convs:
            return []
        self.prepare = self.prepare and _conv3x3_enabled()
        vg = []
        for c in self.convs:
            vg += [c.weight_v, c.weight_g]
        ws = _BankFn.apply(self, *vg)
        for w, wp in zip(ws, self._prepared):
            if wp is not None:
                w._sbmc_wp = wp
        self._prepared = None
        return list(ws)


class installed(object):
    """Context manager: the banks' weights made available to the modules' forward passes (`modules.conv_weight`
    finds them on the convolution objects) and removed again -- a stale weight must never outlive its step."""

    def __init__(self, banks):
        self.banks = banks
        self._convs = []

    def __enter__(self):
        for bank in self.banks:
            for conv, w in zip(bank.convs, bank.weights()):
                conv.__dict__["_sbmc_bank_w"] = w
                self._convs.append(conv)
        return self

    def __exit__(self, *exc):
        for conv in self._convs:
            conv.__dict__.pop("_sbmc_bank_w", None)
        self._convs = []
        return False
