"""Host side of the neighbour halo transport (csrc/halo.hip, include/sbmc_hip.h `sbmc_halo_*`).

One frame sharded over the GPUs of a node exchanges a few rows with each neighbouring rank ~150 times per
training step (sbmc_amd/dist.py).  Through `torch.distributed` P2P every such exchange costs ~50 us of launches,
stream hand-overs and host time however few bytes it moves; here it is two kernel launches: every rank owns a
mailbox in uncached device memory which its neighbours map through HIP IPC, `put` stores rows straight into the
neighbour's mailbox over xGMI and raises a flag there, `get` waits for the flag on the device and unpacks.
`torch.distributed` is used once, to hand the IPC handles round.

New functionality (the reference is single-process; closest analogue: scripts/denoise.py:54-93).
"""
import ctypes
import logging
import os
import socket
import time
import weakref

import torch as th
import torch.distributed as dist

from . import _lib
from .functions import _timed

LOG = logging.getLogger(__name__)

TICKS_PER_SECOND = 100 * 1000 * 1000        # wall_clock64() of gfx950
HANDLE_BYTES = 64


def rows_run(t, r0, r1, nhwc=False):
    """Rows [r0, r1) of `t` ([..., h, w] contiguous, or a 4-d channels-last tensor with nhwc=True) as a run of
    bytes: (address, chunks, chunk_bytes, pitch).  Planar: one chunk per leading index; channels-last: one per
    image (rows of one image are one block of [row, column, channel] memory)."""
    es = t.element_size()
    if nhwc:
        b, c, h, w = t.shape
        if t.stride() != (h * w * c, 1, w * c, c) and not (b == 1 and t.is_contiguous(memory_format=th.channels_last)):
            raise ValueError("not a dense channels-last tensor")
        return (t.data_ptr() + r0 * w * c * es, b, (r1 - r0) * w * c * es, h * w * c * es)
    if not t.is_contiguous():
        raise ValueError("not a contiguous tensor")
    h, w = t.shape[-2:]
    return (t.data_ptr() + r0 * w * es, t.numel() // max(h * w, 1), (r1 - r0) * w * es, h * w * es)


def _pieces(chunks, chunk_bytes, slot_bytes):
    """Splits a run into messages of at most slot_bytes: (first chunk, chunks, byte offset in a chunk, bytes)."""
    if chunks * chunk_bytes <= slot_bytes:
        return [(0, chunks, 0, chunk_bytes)]
    out = []
    if chunk_bytes <= slot_bytes:
        n = slot_bytes // chunk_bytes
        for c0 in range(0, chunks, n):
            out.append((c0, min(n, chunks - c0), 0, chunk_bytes))
        return out
    step = slot_bytes - slot_bytes % 16
    for c0 in range(chunks):
        for b0 in range(0, chunk_bytes, step):
            out.append((c0, 1, b0, min(step, chunk_bytes - b0)))
    return out


class HaloChannel(object):
    """This rank's mailbox plus the mappings of its neighbours' mailboxes.  All calls enqueue on the current
    stream of `device`; sequence numbers are counted here (both ends of a link issue the same exchanges in
    the same order, as with any matched send / receive pair)."""

    def __init__(self, device, slot_bytes, nslots=4, timeout_s=None):
        self.lib = _lib.lib()
        self.device = th.device(device)
        self.slot_bytes = int(slot_bytes + 15) // 16 * 16
        self.nslots = int(nslots)
        if timeout_s is None:
            # (a wait that runs into it poisons the mailbox: every later wait of the step returns at once, the host
            # raises at its next status read -- csrc/halo.hip wait_reached)
            timeout_s = float(os.environ.get("SBMC_HALO_TIMEOUT_S", "120"))
        self.timeout_ticks = int(timeout_s * TICKS_PER_SECOND)
        self.bytes = self.lib.sbmc_halo_bytes(self.slot_bytes, self.nslots)
        base = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(HANDLE_BYTES)
        with th.cuda.device(self.device):
            _lib.check(self.lib.sbmc_halo_alloc(self.bytes, ctypes.byref(base), handle), "sbmc_halo_alloc")
        self.box = base.value
        self.handle = handle.raw
        self.peer = [None, None]            # mapped mailboxes of the up / down neighbour
        self._opened = []
        self.send_seq = [0, 0]
        self.recv_seq = [0, 0]
        self.handshake_ms = None
        self._finalizer = weakref.finalize(self, HaloChannel._release, self.lib, self.box, self._opened)

    @staticmethod
    def _release(lib, box, opened):
        try:
            for p in opened:
                lib.sbmc_halo_close(ctypes.c_void_p(p))
            del opened[:]
            lib.sbmc_halo_free(ctypes.c_void_p(box))
        except Exception:       # interpreter shutdown: the driver reclaims everything anyway
            pass

    def close(self):
        th.cuda.synchronize(self.device)
        self._finalizer()

    # -- wiring -----------------------------------------------------------------------------------------
    def open_peer(self, direction, handle):
        base = ctypes.c_void_p()
        with th.cuda.device(self.device):
            _lib.check(self.lib.sbmc_halo_open(handle, ctypes.byref(base)), "sbmc_halo_open")
        self.peer[direction] = base.value
        self._opened.append(base.value)

    def loopback(self):
        """This rank as both of its own neighbours (cost measurements on one GPU, unit tests): what is sent up
        comes back as if from below and vice versa."""
        self.peer = [self.box, self.box]
        return self

    @classmethod
    def connect(cls, part, device, slot_bytes, nslots=4):
        """Collective over the partition's group: every rank creates its mailbox, the handles go round once,
        every rank maps its neighbours'.  Returns the channel, or None on EVERY rank if any rank could not
        (ranks on different hosts, IPC refused): the callers then stay on torch.distributed P2P."""
        ok, ch, err = True, None, ""
        try:
            ch = cls(device, slot_bytes, nslots)
        except Exception as e:      # noqa: BLE001 -- any failure means "no channel", agreed below
            ok, err = False, "%s: %s" % (type(e).__name__, e)
        mine = (socket.gethostname(), os.getpid(), ch.handle if ch else None, ok)
        everyone = [None] * part.world
        dist.all_gather_object(everyone, mine, group=part.group)
        ok = ok and all(e[3] and e[0] == mine[0] for e in everyone)
        if ok:
            try:
                for d, has, delta in ((0, part.has_up, -1), (1, part.has_down, +1)):
                    if has:
                        host, pid, handle, _ = everyone[part.rank + delta]
                        if pid == os.getpid():
                            raise RuntimeError("neighbour rank lives in this process")
                        ch.open_peer(d, handle)
            except Exception as e:  # noqa: BLE001
                ok, err = False, "%s: %s" % (type(e).__name__, e)
        if ok:
            # a handshake over every link before anything depends on it: a few bytes to each neighbour and back,
            # with a short time-out.  Stores into a mapped mailbox that the other side never sees (a platform
            # on which peer writes or the flags do not behave) show up here, as "no channel", not as a hang
            # in the first training step.
            try:
                ok = ch._handshake(part)
                if not ok:
                    err = "handshake over the mailboxes failed"
            except Exception as e:  # noqa: BLE001
                ok, err = False, "%s: %s" % (type(e).__name__, e)
        flags = [None] * part.world
        dist.all_gather_object(flags, (ok, err), group=part.group)
        if all(f[0] for f in flags):
            return ch
        reasons = sorted({f[1] for f in flags if f[1]}) or ["ranks on different hosts"]
        LOG.info("halo transport: staying on torch.distributed P2P (%s)", "; ".join(reasons))
        if ch is not None:
            ch._finalizer()
        return None

    def _handshake(self, part, timeout_s=10.0):
        """Every rank sends 256 bytes that name it to both neighbours and checks what arrives (all ranks call
        this together: the sequence numbers advance alike on both ends of every link)."""
        mine = th.full((1, 1, 1, 64), float(part.rank + 1), dtype=th.float32, device=self.device)
        got = th.zeros(2, 1, 1, 64, dtype=th.float32, device=self.device)
        keep, self.timeout_ticks = self.timeout_ticks, int(timeout_s * TICKS_PER_SECOND)
        try:
            run = rows_run(mine, 0, 1)
            th.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            self.put(run if part.has_up else None, run if part.has_down else None)
            self.get(rows_run(got[0:1], 0, 1) if part.has_up else None, rows_run(got[1:2], 0, 1) if part.has_down else None)
            th.cuda.synchronize(self.device)
            #: wall time of the first exchange over this rank's links (two launches + the neighbours' answers; the ranks
            #: do not enter it at the same instant, so this bounds the link latency from above)
            self.handshake_ms = (time.perf_counter() - t0) * 1e3
            err = ctypes.c_uint(0)
            with th.cuda.device(self.device):
                _lib.check(self.lib.sbmc_halo_status(self.box, ctypes.byref(err)), "sbmc_halo_status")
        finally:
            self.timeout_ticks = keep
        want_up = float(part.rank) if part.has_up else 0.0           # rank - 1 sends rank - 1 + 1
        want_down = float(part.rank + 2) if part.has_down else 0.0
        return (err.value == 0 and bool((got[0] == want_up).all().item()) and bool((got[1] == want_down).all().item()))

    # -- data path --------------------------------------------------------------------------------------
    def _stream(self):
        return _lib.current_stream(self.device)

    def put(self, up=None, down=None, amax=None):
        """Sends the runs `up` / `down` ((address, chunks, chunk_bytes, pitch), same shape; None: nothing) to
        the neighbours.  amax: the device word with the largest magnitude of the tensor the rows belong to
        (functions.tag_amax); it travels with the message and the receiver's `get(..., amax=word)` raises its own
        word to it."""
        if self.peer[0] is None:
            up = None
        if self.peer[1] is None:
            down = None
        ref = up or down
        if ref is None:
            return
        _, chunks, chunk_bytes, pitch = ref
        for c0, n, b0, nb in _pieces(chunks, chunk_bytes, self.slot_bytes):
            off = c0 * pitch + b0
            with _timed("halo_exchange", self.device):
                rc = self.lib.sbmc_halo_put(
                    self.box, self.peer[0] if up else None, self.peer[1] if down else None,
                    up[0] + off if up else None, down[0] + off if down else None,
                    n, nb, pitch, self.send_seq[0], self.send_seq[1], self.nslots, self.slot_bytes,
                    self.timeout_ticks, _lib.ptr(amax), self._stream())
            _lib.check(rc, "sbmc_halo_put")
            self.send_seq[0] += 1 if up else 0
            self.send_seq[1] += 1 if down else 0

    def get(self, up=None, down=None, add_up=None, add_down=None, add_elem=0, body=None, amax=None):
        """Receives into the runs `up` / `down`; with add_elem (4: float, 2: half) the result is
        add_* + received.  body = (dst_run, src_run): a plain copy in the same launch (src_run None: dst_run is
        filled with zeros).  amax: a device word that is raised to the words the senders attached (`put`)."""
        if self.peer[0] is None:
            up = None
        if self.peer[1] is None:
            down = None
        ref = up or down
        if ref is None:
            if body is not None:
                self._get(None, None, None, None, 0, (0, 0, 0, 0), (0, 1, 0, 1), body)
            return
        _, chunks, chunk_bytes, pitch = ref
        for i, piece in enumerate(_pieces(chunks, chunk_bytes, self.slot_bytes)):
            self._get(up, down, add_up, add_down, add_elem, ref, piece, body if i == 0 else None, amax)

    def _get(self, up, down, add_up, add_down, add_elem, ref, piece, body, amax=None):
        c0, n, b0, nb = piece
        pitch = ref[3]
        add_pitch = (add_up or add_down or (0, 0, 0, 0))[3]
        off, aoff = c0 * pitch + b0, c0 * add_pitch + b0
        bd = body[0] if body else (None, 0, 0, 0)
        bs = (body[1] if body else None) or (None, 0, 0, 0)
        with _timed("halo_exchange", self.device):
            rc = self.lib.sbmc_halo_get(
                self.box, self.peer[0] if up else None, self.peer[1] if down else None,
                up[0] + off if up else None, down[0] + off if down else None,
                add_up[0] + aoff if (add_elem and up) else None, add_down[0] + aoff if (add_elem and down) else None,
                add_elem, n, nb, pitch, add_pitch,
                bd[0], bs[0], bd[1], bd[2], bd[3], bs[3],
                self.recv_seq[0], self.recv_seq[1], self.nslots, self.slot_bytes, self.timeout_ticks,
                _lib.ptr(amax), self._stream())
        _lib.check(rc, "sbmc_halo_get")
        self.recv_seq[0] += 1 if up else 0
        self.recv_seq[1] += 1 if down else 0

    def merge_state_fwd(self, ext, p, top, bot):
        """csrc/halo.hip merge_fwd_kernel: ext [bs, c + 2, top + rows + bot, w] whose overhang rows have been
        `put` -> (own rows with the neighbours' overhangs merged in, what arrived from above, from below)."""
        bs, c2, hd, w = ext.shape
        rows = hd - top - bot
        out = ext.new_empty(bs, c2, rows, w)
        recv_up = ext.new_empty(bs, c2, p, w) if top else None
        recv_down = ext.new_empty(bs, c2, p, w) if bot else None
        with _timed("halo_exchange", self.device):
            rc = self.lib.sbmc_halo_merge_state_fwd_f32(
                self.box, self.peer[0] if top else None, self.peer[1] if bot else None, _lib.ptr(ext), _lib.ptr(out),
                _lib.ptr(recv_up), _lib.ptr(recv_down), bs, c2 - 2, rows, w, p, top, bot,
                self.recv_seq[0], self.recv_seq[1], self.nslots, self.slot_bytes, self.timeout_ticks, self._stream())
        _lib.check(rc, "sbmc_halo_merge_state_fwd_f32")
        self.recv_seq[0] += 1 if top else 0
        self.recv_seq[1] += 1 if bot else 0
        return out, recv_up, recv_down

    def merge_state_bwd(self, ext, recv_up, recv_down, gout, p, top, bot):
        bs, c2, hd, w = ext.shape
        rows = hd - top - bot
        gext = th.empty_like(ext)
        g_up = th.empty_like(recv_up) if top else None
        g_down = th.empty_like(recv_down) if bot else None
        rc = self.lib.sbmc_halo_merge_state_bwd_f32(
            _lib.ptr(ext), _lib.ptr(recv_up), _lib.ptr(recv_down), _lib.ptr(gout), _lib.ptr(gext),
            _lib.ptr(g_up), _lib.ptr(g_down), bs, c2 - 2, rows, w, p, top, bot, self._stream())
        _lib.check(rc, "sbmc_halo_merge_state_bwd_f32")
        return gext, g_up, g_down

    def check(self):
        """Raises if a device-side wait has timed out since the channel was created (synchronises)."""
        err = ctypes.c_uint(0)
        with th.cuda.device(self.device):
            _lib.check(self.lib.sbmc_halo_status(self.box, ctypes.byref(err)), "sbmc_halo_status")
        if err.value:
            what = {1: "a free slot towards the rank above", 2: "a free slot towards the rank below",
                    3: "rows from the rank above", 4: "rows from the rank below"}.get(err.value, "?")
            raise RuntimeError("halo transport: a kernel gave up waiting for %s (the neighbour is gone or the two "
                               "ranks disagree about the sequence of exchanges)" % what)
