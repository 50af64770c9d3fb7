"""One frame on several MI355X: H-slab sharding with halo exchange over RCCL / xGMI.

New functionality (the reference is single-process; its closest analogue is the overlapped
tiling of scripts/denoise.py:54-93, which recomputes a 256-px halo per tile).  Design
(SURVEY.md section 8e), one process per GPU, rank r owns the contiguous row slab r:

* per-sample 1x1 ConvChains (embedding, kernel regressor) and the mean over samples are
  pointwise in space: no communication;
* the U-net: every ConvChain of vertical reach R (three 3x3 convs -> R = 3) first receives
  R rows from each neighbouring rank (`halo_pad`), runs unchanged on the padded slab, and
  drops the R rows that saw the artificial zero padding; at the true image border nothing
  is padded or dropped, so the convolutions' own zero padding acts exactly as in the
  single-GPU model.  2x2 max-pools are local because slab boundaries are multiples of 4
  rows; the bilinear x2 upsample takes 1 halo row of the coarse map;
* the splat: the running state (sum_r, sum_w, max_w) is an associative log-sum-exp monoid
  (reference sbmc/modules.py:450-471), so every rank predicts kernels for, and splats, its OWN
  samples only -- into a destination slab extended by the kernel radius p towards its
  neighbours (`functions.SplatAll` in its row-slab form) -- then sends the p overhang rows of
  the state (c + 2 channels x p rows x W, 256 KB at 720p) to the neighbour that owns them, which
  merges them into its own rows with the reference's merge rule: one neighbour exchange per
  frame and direction, nothing recomputed.  (Slabs thinner than p rows, or gather kernels: the
  regressor's inputs are halo-padded by p instead and the halo's destination rows dropped.)
* training: the loss is the global mean (each rank contributes its rows); parameter
  gradients are packed into one persistent flat buffer and summed by ONE all-reduce.

`halo_pad` is a `torch.autograd.Function`: its backward sends the gradient of the halo
rows back to the rank that owns them, so autograd over the sharded graph equals autograd
over the full frame.  Neighbour traffic is batched `isend/irecv` pairs (the direct xGMI
link between adjacent GPUs); works unchanged on gloo (CPU tests) and nccl (= RCCL).  When
all ranks run a U-net channels-last, its halo rows travel in memory order and are received
in place, overlapped with the copy of the slab's own rows (`_HaloPad`).
"""
import torch as th
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib
from . import functions as funcs
from .utils import crop_like

__all__ = ["SlabPartition", "halo_pad", "sharded_autoencoder", "ShardedDenoiser", "HaloTimeout"]


class HaloTimeout(RuntimeError):
    """Raised by `ShardedDenoiser.train_step` on EVERY rank of the partition, at the same point of the step, when any
    rank's halo mailbox recorded a time-out during it (the flag rides in the last gradient bucket's all-reduce).  The
    step's results are void, nothing was applied; the ranks have already left the IPC transport together
    (`settle_transport`), so the caller may simply repeat the step -- it then runs over torch.distributed P2P."""


class SlabPartition(object):
    """Rows [y0, y1) of an H-row frame owned by `rank` out of `world` ranks.

    Slab boundaries are multiples of `align` rows (4: two 2x2 poolings in the U-net)."""

    def __init__(self, height, world, rank, align=4, group=None):
        self.height, self.world, self.rank, self.group, self.align = height, world, rank, group, align
        if world > 1 and height % align != 0:
            raise ValueError("sharded path needs the frame height to be a multiple of %d" % align)
        units = height // align if world > 1 else height
        unit = align if world > 1 else 1
        base, rem = divmod(units, world)
        u0 = rank * base + min(rank, rem)
        u1 = u0 + base + (1 if rank < rem else 0)
        self.y0, self.y1 = u0 * unit, u1 * unit
        self.has_up = rank > 0
        self.has_down = rank < world - 1
        #: halo.HaloChannel once ShardedDenoiser (or a test) has connected the ranks of one node through
        #: IPC mailboxes; None: neighbour rows travel through torch.distributed P2P
        self.channel = None
        self._agreed = {}

    @property
    def rows(self):
        return self.y1 - self.y0

    @property
    def min_rows(self):
        """Rows of the thinnest slab of the partition (the same number on every rank)."""
        if self.world == 1:
            return self.height
        return (self.height // self.align // self.world) * self.align

    def peer(self, delta):
        r = self.rank + delta
        return dist.get_global_rank(self.group, r) if self.group is not None else r


class _Nothing(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NOTHING = _Nothing()


def _exchange_into(part, to_up, to_down, into_up, into_down, between=None):
    """Sends `to_up` to rank-1 and `to_down` to rank+1, receives their counterparts into `into_up` /
    `into_down` (written in place; may be views of a larger buffer).  All four are contiguous tensors; a pair
    (what one rank sends down, what the next receives from above) has the same shape on both sides.
    `between`: called after the transfers have been started and before they are waited for -- work that needs
    neither (the copy of the slab's own rows) overlaps the transfer.

    RCCL ("nccl") moves device tensors directly over xGMI.  A backend without device support (gloo: the CPU
    tests, and the single-GPU two-process tests of the device kernels) gets the rows staged through host
    memory."""
    staged = to_up.is_cuda and dist.get_backend(part.group) != "nccl"
    ops, landed = [], []
    for has, src, dst, delta in ((part.has_up, to_up, into_up, -1), (part.has_down, to_down, into_down, +1)):
        if not has:
            continue
        if staged:
            src = src.cpu()
            tmp = th.empty_like(src)
            landed.append((dst, tmp))
        else:
            tmp = dst
        ops += [dist.P2POp(dist.isend, src, part.peer(delta), group=part.group),
                dist.P2POp(dist.irecv, tmp, part.peer(delta), group=part.group)]
    dev = to_up.device
    with funcs._timed("halo_exchange", dev) if to_up.is_cuda else _NOTHING:
        reqs = dist.batch_isend_irecv(ops) if ops else []
    if between is not None:
        between()
    with funcs._timed("halo_exchange", dev) if to_up.is_cuda else _NOTHING:
        for req in reqs:
            req.wait()
        for dst, tmp in landed:
            dst.copy_(tmp)


def _exchange(part, to_up, to_down):
    """Sends `to_up` to rank-1 and `to_down` to rank+1; returns (from_up, from_down) as new tensors in the
    default (planar) memory order, whatever the order of the arguments -- layout-agnostic."""
    to_up, to_down = to_up.contiguous(), to_down.contiguous()
    from_up = th.empty_like(to_up) if part.has_up else None
    from_down = th.empty_like(to_down) if part.has_down else None
    _exchange_into(part, to_up, to_down, from_up, from_down)
    return from_up, from_down


def _all_agree(flag, part, key):
    """Logical AND of `flag` over the ranks of the partition (cached per key ON the partition object, whose
    lifetime is the process group's: one tiny collective per new shape).  Used for decisions every rank takes
    by *measurement* but that both ends of an exchange must share."""
    key = tuple(key)
    if key not in part._agreed:
        t = th.tensor([1.0 if flag else 0.0])
        part._agreed[key] = bool(_all_reduce_min(t, part).item() > 0.5)
    return part._agreed[key]


def _all_reduce_min(t, part):
    if dist.get_backend(part.group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=part.group)
    return t.cpu()


def _all_reduce_sum(t, part):
    """Sum over ranks; device tensors go through host memory when the backend is not RCCL."""
    if dist.get_backend(part.group) == "nccl":
        if not t.is_cuda:
            t = t.cuda()
        dist.all_reduce(t, group=part.group)
        return t
    dev = t.device
    h = t.cpu()
    dist.all_reduce(h, group=part.group)
    return h.to(dev)


def _all_reduce_sum_start(t, part):
    """Starts the in-place sum of `t` over the ranks and returns what `wait()` must be called on before `t` is
    read (None: already done).  RCCL: asynchronous on its own stream, ordered after what the current stream
    has enqueued so far -- the rest of the backward overlaps it.  Other backends: done on return."""
    if part.world == 1:
        return None
    if t.is_cuda and dist.get_backend(part.group) == "nccl":
        return dist.all_reduce(t, group=part.group, async_op=True)
    summed = _all_reduce_sum(t, part)
    if summed.data_ptr() != t.data_ptr():
        t.copy_(summed)
    return None


def _rows_nhwc(t, r0, r1):
    """Rows [r0, r1) of a channels-last map of ONE image as a contiguous [1, rows, w, c] view (one block of
    memory: nothing is copied, and writes through the view land in `t`)."""
    return t[..., r0:r1, :].permute(0, 2, 3, 1)


class _HaloPad(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, part, nhwc_wire=False, amax=None):
        ctx.r, ctx.part, ctx.rows = r, part, x.shape[-2]
        h = x.shape[-2]
        if h < r:
            raise RuntimeError("slab of %d rows is thinner than the halo (%d)" % (h, r))
        # (decided from what every rank shares -- the agreed flag and the batch size -- never from this rank's
        # strides: both ends of an exchange must interpret the bytes alike)
        ctx.nhwc_wire = bool(nhwc_wire and x.dim() == 4 and x.shape[0] == 1)
        ctx.via_channel = part.channel is not None and x.is_cuda
        if ctx.via_channel:
            return _halo_pad_channel(ctx, x, r, part, bool(nhwc_wire and x.dim() == 4), amax)
        if ctx.nhwc_wire:
            x = x.contiguous(memory_format=th.channels_last)     # a no-op in the channels-last U-net
            # every rank runs this U-net channels-last (agreed, see sharded_autoencoder) on one image: a run of
            # rows is one block of [row, column, channel] memory.  The halo rows travel in that order and land
            # directly in the padded map; the slab's own rows are copied while they travel.
            top = r if part.has_up else 0
            bot = r if part.has_down else 0
            out = th.empty(x.shape[:-2] + (top + h + bot, x.shape[-1]), dtype=x.dtype, device=x.device,
                           memory_format=th.channels_last)
            _exchange_into(part, _rows_nhwc(x, 0, r), _rows_nhwc(x, h - r, h),
                           _rows_nhwc(out, 0, top), _rows_nhwc(out, top + h, top + h + bot),
                           between=lambda: out[..., top:top + h, :].copy_(x))
            if amax is not None:
                # the padded map's scale for the 3 x 3 kernels: this slab's word raised to the received rows' magnitudes
                # (two launches over a few rows instead of an absmax pass over the whole padded map)
                with th.cuda.device(x.device):
                    if top:
                        funcs.raise_amax(amax, _rows_nhwc(out, 0, top))
                    if bot:
                        funcs.raise_amax(amax, _rows_nhwc(out, top + h, top + h + bot))
            return out
        from_up, from_down = _exchange(part, x[..., :r, :], x[..., -r:, :])
        if funcs._is_channels_last(x):
            # keep the U-net's channels-last order (the halo rows travelled in planar order: layout-agnostic)
            top = r if from_up is not None else 0
            bot = r if from_down is not None else 0
            out = th.empty(x.shape[:-2] + (top + h + bot, x.shape[-1]), dtype=x.dtype, device=x.device,
                           memory_format=th.channels_last)
            out[..., top:top + h, :].copy_(x)
            if from_up is not None:
                out[..., :top, :].copy_(from_up)
            if from_down is not None:
                out[..., top + h:, :].copy_(from_down)
            return out
        pieces = [t for t in (from_up, x, from_down) if t is not None]
        return th.cat(pieces, -2) if len(pieces) > 1 else x.clone()

    @staticmethod
    def backward(ctx, g):
        r, part, h = ctx.r, ctx.part, ctx.rows
        top = r if part.has_up else 0
        # gradient of my halo rows goes back to their owners; theirs for my edge rows comes here
        if ctx.via_channel:
            return _halo_pad_channel_bwd(ctx, g, r, part, h), None, None, None, None
        if ctx.nhwc_wire:
            g = g.contiguous(memory_format=th.channels_last)     # the wire order both ends agreed on
            hp = g.shape[-2]
            box = {}
            back_up = th.empty_like(_rows_nhwc(g, 0, r)) if part.has_up else None
            back_down = th.empty_like(_rows_nhwc(g, 0, r)) if part.has_down else None

            def own_rows():
                box["gx"] = g[..., top:top + h, :].clone(memory_format=th.channels_last)
            _exchange_into(part, _rows_nhwc(g, 0, r), _rows_nhwc(g, hp - r, hp), back_up, back_down,
                           between=own_rows)
            gx = box["gx"]
            if back_up is not None:
                _rows_nhwc(gx, 0, r).add_(back_up)
            if back_down is not None:
                _rows_nhwc(gx, h - r, h).add_(back_down)
            return gx, None, None, None, None
        from_up, from_down = _exchange(part, g[..., :r, :], g[..., g.shape[-2] - r:, :])
        gx = g[..., top:top + h, :].clone()
        if from_up is not None:
            gx[..., :r, :] += from_up
        if from_down is not None:
            gx[..., h - r:, :] += from_down
        return gx, None, None, None, None


def _halo_pad_channel(ctx, x, r, part, nhwc, amax=None):
    """`_HaloPad.forward` through the IPC mailboxes (halo.HaloChannel): one `put` (my edge rows into the
    neighbours' mailboxes) and one `get` (their rows into the padded map + the copy of my own rows) -- two
    launches, nothing on the host.  nhwc: every rank holds this tensor channels-last (agreed).
    amax: the word of x's largest magnitude (functions.tag_amax).  It travels with the rows, and the `get` raises
    it to the neighbours' words: it then bounds the padded map, which the 3 x 3 convolution scales by (the
    absmax pass it would otherwise run over every padded map was 6.7 of a rank-of-8's 51.5 ms)."""
    from .halo import rows_run
    ch, h = part.channel, x.shape[-2]
    ctx.nhwc = nhwc
    fmt = th.channels_last if nhwc else th.contiguous_format
    x = x.contiguous(memory_format=fmt)
    top = r if part.has_up else 0
    bot = r if part.has_down else 0
    out = th.empty(x.shape[:-2] + (top + h + bot, x.shape[-1]), dtype=x.dtype, device=x.device, memory_format=fmt)
    ch.put(rows_run(x, 0, r, nhwc) if top else None, rows_run(x, h - r, h, nhwc) if bot else None, amax=amax)
    ch.get(rows_run(out, 0, top, nhwc) if top else None, rows_run(out, top + h, top + h + bot, nhwc) if bot else None,
           body=(rows_run(out, top, top + h, nhwc), rows_run(x, 0, h, nhwc)), amax=amax)
    return out


def _halo_pad_channel_bwd(ctx, g, r, part, h):
    """The adjoint: the gradient of my halo rows goes back to their owners, theirs for my edge rows is added
    to my own rows' gradient inside the `get`."""
    from .halo import rows_run
    ch, nhwc = part.channel, ctx.nhwc
    fmt = th.channels_last if nhwc else th.contiguous_format
    g = g.contiguous(memory_format=fmt)
    hp = g.shape[-2]
    top = r if part.has_up else 0
    bot = r if part.has_down else 0
    gx = th.empty(g.shape[:-2] + (h, g.shape[-1]), dtype=g.dtype, device=g.device, memory_format=fmt)
    es = g.element_size()
    if es not in (2, 4):
        raise RuntimeError("halo transport: float32 / float16 gradients only")
    ch.put(rows_run(g, 0, r, nhwc) if top else None, rows_run(g, hp - r, hp, nhwc) if bot else None)
    if nhwc and g.shape[0] == 1 and not (top and bot and h < 2 * r) and g._base is None:
        # (g._base is None: the gradient is a tensor of its own -- what a convolution's data-gradient kernel returns --,
        # never a view into memory another branch of the graph may still read; the kernels below write through raw
        # pointers, so a magnitude tag on it would go stale without its version counter noticing)
        g.__dict__.pop("_sbmc_amax", None)
        # one channels-last image: its own rows are one dense block of the incoming gradient -- the neighbours'
        # contributions are added to its edge rows in place and that block is handed on as a view (no copy of the
        # slab; the gradient of a padded map has no other consumer: it comes out of the first convolution's
        # data-gradient kernel)
        ch.get(up=rows_run(g, top, top + r, nhwc) if top else None, add_up=rows_run(g, top, top + r, nhwc) if top else None,
               down=rows_run(g, top + h - r, top + h, nhwc) if bot else None,
               add_down=rows_run(g, top + h - r, top + h, nhwc) if bot else None, add_elem=es)
        return g[..., top:top + h, :]
    if top and bot and h < 2 * r:
        # both neighbours reach the same rows: first the rows from above plus everything else, then the rows
        # from below are added in place
        ch.get(up=rows_run(gx, 0, r, nhwc), add_up=rows_run(g, top, top + r, nhwc), add_elem=es,
               body=(rows_run(gx, r, h, nhwc), rows_run(g, top + r, top + h, nhwc)) if h > r else None)
        ch.get(down=rows_run(gx, h - r, h, nhwc), add_down=rows_run(gx, h - r, h, nhwc), add_elem=es)
        return gx
    lo, hi = (r if top else 0), (h - r if bot else h)
    ch.get(up=rows_run(gx, 0, r, nhwc) if top else None, add_up=rows_run(g, top, top + r, nhwc) if top else None,
           down=rows_run(gx, h - r, h, nhwc) if bot else None,
           add_down=rows_run(g, top + h - r, top + h, nhwc) if bot else None, add_elem=es,
           body=(rows_run(gx, lo, hi, nhwc), rows_run(g, top + lo, top + hi, nhwc)) if hi > lo else None)
    return gx


def halo_pad(x, r, part, nhwc_wire=False):
    """[..., h, w] -> [..., (r if up) + h + (r if down), w] with the neighbours' edge rows.
    nhwc_wire: ALL ranks hold this tensor channels-last (an agreed fact, not a local guess): the rows then
    travel in memory order and land in place."""
    if r == 0 or part.world == 1:
        return x
    if nhwc_wire and x.dim() == 4 and funcs.wants_amax(x) and (
            part.channel is not None or (x.shape[0] == 1 and dist.get_backend(part.group) == "nccl")):
        # the scale of the 3 x 3 convolution that reads the padded map: this slab's word, raised in place to the
        # neighbours' -- by the exchange itself through the mailboxes, by two launches over the received rows through
        # torch.distributed -- (a larger bound stays a bound for everyone else who holds the word)
        amax = funcs.ensure_amax(x)
        return funcs.tag_amax(_HaloPad.apply(x, r, part, nhwc_wire, amax), amax)
    return _HaloPad.apply(x, r, part, nhwc_wire)


class _HaloRefresh(th.autograd.Function):
    """A padded map (top + h + bot rows whose outer rows hold stale values: the result of a padded convolution
    on a halo-padded slab) gets FRESH halo rows from the neighbours, in place: the rows of a slab that a chain
    of convolutions keeps halo-padded are never copied between its convolutions -- `_crop_halo` + `halo_pad`
    would copy the whole slab to renew a row or two.  IPC mailboxes only (the kernels write through raw
    pointers; torch sees a view of the same tensor).

    Backward, in place on the incoming gradient as well: the halo rows' gradient goes back to the rows' owners,
    theirs for this slab's edge rows is added to them, and the halo rows' own gradient becomes zero -- the
    producer of the padded map computes nothing from it (its stale rows are not part of the function)."""

    @staticmethod
    def forward(ctx, x, r, part, nhwc, amax=None):
        from .halo import rows_run
        ch = part.channel
        top = r if part.has_up else 0
        bot = r if part.has_down else 0
        hp = x.shape[-2]
        h = hp - top - bot
        ctx.r, ctx.part, ctx.nhwc, ctx.h = r, part, nhwc, h
        ch.put(rows_run(x, top, top + r, nhwc) if top else None, rows_run(x, top + h - r, top + h, nhwc) if bot else None,
               amax=amax)
        ch.get(rows_run(x, 0, top, nhwc) if top else None, rows_run(x, top + h, hp, nhwc) if bot else None, amax=amax)
        # a second handle on the same memory with the same strides: returning `x` itself would make autograd
        # re-view it (`x.view_as(x)`), which loses the channels-last strides of a one-image batch -- MIOpen then
        # transposes every input -- and marking it dirty would invalidate what its in-place producer saved
        return x.detach()

    @staticmethod
    def backward(ctx, g):
        from .halo import rows_run
        r, part, nhwc, h = ctx.r, ctx.part, ctx.nhwc, ctx.h
        ch = part.channel
        fmt = th.channels_last if nhwc else th.contiguous_format
        g = g.contiguous(memory_format=fmt)
        if g._base is not None:
            g = g.clone(memory_format=fmt)     # never write through a view into a gradient another branch may read
        g.__dict__.pop("_sbmc_amax", None)     # (raw writes below: a magnitude tag would go stale unnoticed)
        top = r if part.has_up else 0
        bot = r if part.has_down else 0
        hp = top + h + bot
        es = g.element_size()
        ch.put(rows_run(g, 0, top, nhwc) if top else None, rows_run(g, top + h, hp, nhwc) if bot else None)
        up = rows_run(g, top, top + r, nhwc) if top else None
        down = rows_run(g, top + h - r, top + h, nhwc) if bot else None
        # the halo rows' gradient has been sent: zero it in the same launch where the two halo runs are ONE
        # 2-d run (a single channels-last image with both neighbours), else with torch
        zero = None
        if nhwc and g.shape[0] == 1 and top and bot:
            a = rows_run(g, 0, top, nhwc)
            zero = ((a[0], 2, a[2], (top + h) * g.shape[-1] * g.shape[1] * es), None)
        if top and bot and h < 2 * r:                  # both neighbours reach the same rows: one after the other
            ch.get(up=up, add_up=up, add_elem=es, body=zero)
            ch.get(down=down, add_down=down, add_elem=es)
        else:
            ch.get(up=up, down=down, add_up=up, add_down=down, add_elem=es, body=zero)
        if zero is None:
            if top:
                g[..., :top, :].zero_()
            if bot:
                g[..., top + h:, :].zero_()
        return g, None, None, None, None


def halo_refresh(x, r, part, nhwc=False):
    """`x`: top + h + bot rows (top / bot = r where there is a neighbour), outer rows stale -> the same tensor
    with the neighbours' current edge rows there; None where this in-place form does not apply (no IPC
    channel, memory not dense in the agreed layout, half-precision planar...): the caller then crops and pads.
    Either form puts the same messages on the wire, so neighbouring ranks need not choose alike."""
    if part.channel is None or not x.is_cuda or x.dim() != 4 or x.element_size() not in (2, 4):
        return None
    top = r if part.has_up else 0
    bot = r if part.has_down else 0
    if x.shape[-2] - top - bot < r:
        return None
    if nhwc:
        b, c, h, w = x.shape
        if x.stride() != (h * w * c, 1, w * c, c):
            return None
    elif not x.is_contiguous():
        return None
    if nhwc and funcs.wants_amax(x):
        amax = funcs.ensure_amax(x)          # (x's word covers its stale outer rows too: whoever wrote x wrote them)
        return funcs.tag_amax(_HaloRefresh.apply(x, r, part, True, amax), amax)
    return _HaloRefresh.apply(x, r, part, bool(nhwc))


class _CropRows(th.autograd.Function):
    """y[..., top : h - bot, :] whose backward is ONE zero-padding pass (torch's slice backward fills a
    zero tensor and then copies the gradient into it: two passes over every U-net activation)."""

    @staticmethod
    def forward(ctx, y, top, bot):
        ctx.pad = (top, bot)
        return y[..., top:y.shape[-2] - bot, :]

    @staticmethod
    def backward(ctx, g):
        top, bot = ctx.pad
        return F.pad(g, (0, 0, top, bot)), None, None


def _crop_halo(y, r, part):
    top = r if part.has_up else 0
    bot = r if part.has_down else 0
    if top == 0 and bot == 0:
        return y
    out = _CropRows.apply(y, top, bot)
    known = funcs.known_amax(y)
    return out if known is None else funcs.tag_amax(out, known)     # (rows of y: its bound holds)


def _reach(chain):
    """Vertical receptive reach (rows per side) of a stride-1 ConvChain."""
    r = 0
    for m in chain.modules():
        if isinstance(m, th.nn.Conv2d):
            if m.stride[0] != 1:
                raise ValueError("sharded path expects stride-1 convolutions")
            r += m.kernel_size[0] // 2
    return r


#: Slabs thinner than this many rows (at the frame's resolution) exchange ONE halo row before every convolution
#: instead of three before every chain of three: a chain then convolves rows + 2 three times instead of rows + 6.
#: Measured at 8 ranks of a 720p frame (23 rows per rank at the U-net's coarsest level; find records for the
#: new heights are shipped): -4.4 ms of convolution work per step (80.8 -> 76.4 ms with the exchanges stubbed
#: out) against +5.5 ms for the 62 extra exchanges issued through torch.distributed (84.5 vs 83.4 ms with the
#: exchanges running over RCCL to the rank itself, tools/rank_cost.py --rccl-self): a loss while a neighbour
#: exchange costs ~50 us, so through torch.distributed P2P it stays OFF (0).  Through the IPC mailboxes
#: (halo.HaloChannel, two launches per exchange) it is a gain: on below PER_CONV_HALO_BELOW_CHANNEL rows.
#: None = that automatic choice; a number forces the threshold for both transports (tests).
PER_CONV_HALO_BELOW = None
PER_CONV_HALO_BELOW_CHANNEL = 128


def _per_conv_below(part):
    if PER_CONV_HALO_BELOW is not None:
        return PER_CONV_HALO_BELOW
    import os
    if os.environ.get("SBMC_PER_CONV_HALO_BELOW"):        # measurements (the same value on every rank)
        return int(os.environ["SBMC_PER_CONV_HALO_BELOW"])
    return PER_CONV_HALO_BELOW_CHANNEL if part.channel is not None else 0


def _per_conv(chain, part):
    return (hasattr(chain, "_run") and part.min_rows < _per_conv_below(part)
            and all(m.padding[0] == m.kernel_size[0] // 2 and m.stride[0] == 1
                    for m in chain.modules() if isinstance(m, th.nn.Conv2d)))


def _chain(chain, x, part, nhwc=False):
    if _per_conv(chain, part):          # (a property of the partition and the module: the same on every rank)
        return chain._run(list(chain.children()), x,
                          halo=(lambda t, r: halo_pad(t, r, part, nhwc), lambda t, r: _crop_halo(t, r, part),
                                lambda t, r: halo_refresh(t, r, part, nhwc)))
    r = _reach(chain)
    return _crop_halo(chain(halo_pad(x, r, part, nhwc)), r, part)


def _level(level, x, part, nhwc=False):
    left = _chain(level.left, x, part, nhwc)
    if level.is_last:
        return left
    if left.shape[-2] % 2:
        raise RuntimeError("sharded path needs an even number of rows at every U-net level")
    if funcs.PoolSkip.supported(left, level.downsample):
        known = funcs.known_amax(left)
        pooled, left = funcs.PoolSkip.apply(left)        # (down path + skip connection: one node, one gradient pass)
        if known is not None:
            funcs.tag_amax(left, known)
    else:
        pooled = level.downsample(left)
    if isinstance(level.downsample, (th.nn.MaxPool2d, th.nn.AvgPool2d)) and funcs.known_amax(left) is not None:
        funcs.tag_amax(pooled, funcs.known_amax(left))            # pooling grows no magnitude
    coarse = _level(level.next_level, pooled, part, nhwc)
    padded = halo_pad(coarse, 1, part, nhwc)
    top, bot = int(part.has_up), int(part.has_down)
    if funcs.upsample_cat_nhwc_supported(padded, left, top, bot):
        cat = funcs.UpsampleCatNHWC.apply(padded, left, top, bot)
        bound = funcs.bound_amax(funcs.known_amax(padded), funcs.known_amax(left))
        return _chain(level.right, cat if bound is None else funcs.tag_amax(cat, bound), part, nhwc)
    if funcs.upsample_cat_supported(padded, left, top, bot):
        # one pass, the upsampled tensor never exists (functions.UpsampleCat in its row-slab form)
        return _chain(level.right, funcs.UpsampleCat.apply(padded, left, top, bot), part, nhwc)
    up = F.interpolate(padded, size=(2 * padded.shape[-2], left.shape[-1]), mode="bilinear",
                       align_corners=False)
    up = _crop_halo(up, 2, part)
    return _chain(level.right, th.cat([up, left], 1), part, nhwc)


def sharded_autoencoder(autoencoder, x, part):
    """`modules.Autoencoder.forward` on a row slab (exact, see the module docstring).  Like the whole-frame
    U-net it runs channels-last when that measures faster for this slab's shape (modules.unet_channels_last:
    MIOpen's NHWC solvers, given find records for the shape; cropping rows of a channels-last map is free)."""
    if part.world == 1:
        return autoencoder(x)
    from . import modules as ops
    # every rank measures for itself (an edge rank convolves a different height) and the ranks then settle on
    # channels-last only if all of them would pick it: halo rows can then travel in [row, column, channel]
    # memory order and land in place (_HaloPad), and no rank is left with the layout that is slow for it
    reach = _reach(autoencoder.net.left)
    rows = x.shape[-2] + reach * (int(part.has_up) + int(part.has_down))     # what the first chain convolves
    mine = ops.unet_channels_last(autoencoder, x, rows=rows)
    grad = th.is_grad_enabled() and any(q.requires_grad for q in autoencoder.parameters())
    # (the key holds nothing rank-specific: every rank meets a new key at the same call)
    if _all_agree(mine, part, (part.world, part.height, x.shape[0], x.shape[1], x.shape[-1], x.dtype, grad,
                               th.is_autocast_enabled())):
        if funcs.ToChannelsLast.supported(x) and x.dtype == th.float32:
            xin, amax = funcs.ToChannelsLast.apply(x, True)       # (the first convolution's scale, found on the way)
            funcs.tag_amax(xin, amax)
        elif funcs.ToChannelsLast.supported(x):
            xin = funcs.ToChannelsLast.apply(x)                   # (half activations: no scales)
        else:
            xin = x.contiguous(memory_format=th.channels_last)
        y = _level(autoencoder.net, xin, part, nhwc=True)
        if autoencoder.keep_channels_last:
            return y
        return funcs.FromChannelsLast.apply(y) if funcs.FromChannelsLast.supported(y) else y.contiguous()
    return _level(autoencoder.net, x, part)


class _OverhangExchange(th.autograd.Function):
    """state [bs, c+2, top + rows + bot, w] (this rank's log-sum-exp partial on its extended slab)
    -> (own [bs, c+2, rows, w], from_up [bs, c+2, p, w], from_down [bs, c+2, p, w]): the overhang
    rows go to the neighbours that own them, theirs for this rank's edge rows come back (an empty
    tensor where there is no neighbour).  Pure communication (linear): backward returns the
    gradient of what was received to its sender."""

    @staticmethod
    def forward(ctx, state, p, part):
        top = p if part.has_up else 0
        bot = p if part.has_down else 0
        hd = state.shape[-2]
        ctx.part, ctx.p = part, p
        from_up, from_down = _exchange(part, state[..., :top, :], state[..., hd - bot:, :])
        none = state.new_zeros(state.shape[:-2] + (0, state.shape[-1]))
        return (state[..., top:hd - bot, :].contiguous(),
                none if from_up is None else from_up, none if from_down is None else from_down)

    @staticmethod
    def backward(ctx, g_own, g_up, g_down):
        part, p = ctx.part, ctx.p
        shape = g_own.shape[:-2] + (p, g_own.shape[-1])
        g_up = g_own.new_zeros(shape) if g_up is None else g_up
        g_down = g_own.new_zeros(shape) if g_down is None else g_down
        back_up, back_down = _exchange(part, g_up, g_down)
        pieces = [t for t in (back_up, g_own, back_down) if t is not None]
        return (th.cat(pieces, -2) if len(pieces) > 1 else g_own), None, None


class _MergeOverhangChannel(th.autograd.Function):
    """`merge_overhang` through the IPC mailboxes: the overhang rows are `put` into the neighbours' mailboxes,
    one kernel waits for theirs and merges them into this slab's edge rows (csrc/halo.hip, the reference's
    rule sbmc/modules.py:450-471, first the rows from above, then those from below, like `_merge_rows`); the
    backward runs the merge's adjoint locally and returns the gradient of what was received to its sender."""

    @staticmethod
    def forward(ctx, state, p, part):
        from .halo import rows_run
        ch = part.channel
        state = state.contiguous()
        top = p if part.has_up else 0
        bot = p if part.has_down else 0
        hd = state.shape[-2]
        ch.put(rows_run(state, 0, top) if top else None, rows_run(state, hd - bot, hd) if bot else None)
        out, recv_up, recv_down = ch.merge_state_fwd(state, p, top, bot)
        ctx.part, ctx.p = part, p
        ctx.save_for_backward(state, recv_up, recv_down)
        return out

    @staticmethod
    def backward(ctx, gout):
        from .halo import rows_run
        part, p = ctx.part, ctx.p
        ch = part.channel
        state, recv_up, recv_down = ctx.saved_tensors
        top = p if part.has_up else 0
        bot = p if part.has_down else 0
        hd = state.shape[-2]
        gext, g_up, g_down = ch.merge_state_bwd(state, recv_up, recv_down, gout.contiguous(), p, top, bot)
        ch.put(rows_run(g_up, 0, p) if top else None, rows_run(g_down, 0, p) if bot else None)
        ch.get(rows_run(gext, 0, top) if top else None, rows_run(gext, hd - bot, hd) if bot else None)
        return gext, None, None


def _merge_rows(state, other, r0, r1, c):
    """Rows [r0, r1) of `state` (+)= `other`, the merge of two running-softmax states (reference
    sbmc/modules.py:450-471): M = max(m1, m2), sum = sum1 * exp(m1 - M) + sum2 * exp(m2 - M).
    Channels: c of sum_r, sum_w, max_w."""
    a = state[..., r0:r1, :]
    m = th.max(a[:, c + 1:], other[:, c + 1:])
    sa, sb = th.exp(a[:, c + 1:] - m), th.exp(other[:, c + 1:] - m)
    merged = th.cat([a[:, :c + 1] * sa + other[:, :c + 1] * sb, m], 1)
    return th.cat([state[..., :r0, :], merged, state[..., r1:, :]], -2)


def merge_overhang(sum_r, sum_w, max_w, p, part):
    """The cross-rank step of the sharded splat.  In: this rank's partial state on
    its slab extended by p rows towards every neighbour; out: the complete state of its own rows."""
    c = sum_r.shape[1]
    if (part.channel is not None and sum_r.is_cuda and sum_r.dtype == th.float32 and c <= 8
            and sum_r.shape[0] * (c + 2) * p * sum_r.shape[-1] * 4 <= part.channel.slot_bytes):     # (one message per edge)
        # put + ONE merge kernel (csrc/halo.hip) instead of an exchange and 9 torch kernels per edge
        own = _MergeOverhangChannel.apply(th.cat([sum_r, sum_w, max_w], 1), p, part)
        return own[:, :c], own[:, c:c + 1], own[:, c + 1:]
    own, from_up, from_down = _OverhangExchange.apply(th.cat([sum_r, sum_w, max_w], 1), p, part)
    rows = own.shape[-2]
    if part.has_up:
        own = _merge_rows(own, from_up, 0, p, c)
    if part.has_down:
        own = _merge_rows(own, from_down, rows - p, rows, c)
    return own[:, :c], own[:, c:c + 1], own[:, c + 1:]


class ShardedDenoiser(object):
    """Runs a `Multisteps` model on this rank's slab of one frame.

    The batch holds the slab rows only: radiance [bs, spp, 3, rows, w], features
    [bs, spp, nf, rows, w], global_features [bs, ngf, 1, 1] (replicated) and, for training,
    target_image [bs, 3, rows, w].
    """

    def __init__(self, model, part, merge_state=None):
        self.model, self.part = model, part
        p = (model.ksize - 1) // 2
        possible = model.splat and part.world > 1 and part.min_rows >= p
        # merge_state=False forces the halo-recompute form of the splat (kept for the comparison)
        self.merge_state = possible if merge_state is None else (bool(merge_state) and possible)
        self._flat = None
        self._channel_tried = False
        self._hook_handles = []
        self.transport_note = None

    def _connect(self, device, bs, w):
        """Once, at the first frame (a collective: every rank gets here): the IPC mailboxes between
        neighbouring ranks (halo.HaloChannel) when all ranks sit on one node; SBMC_HALO_TRANSPORT=p2p keeps
        torch.distributed P2P (RCCL over xGMI; gloo: staged through the host)."""
        import os
        self._channel_tried = True
        part = self.part
        if part.world == 1 or part.channel is not None or device.type != "cuda":
            return
        if os.environ.get("SBMC_HALO_TRANSPORT", "ipc").lower() != "ipc":
            return
        from .halo import HaloChannel
        # the largest U-net message: 3 rows of a right branch's input (3 * width channels at full width; the
        # same number of bytes at every level); larger runs (the halo-recompute form of the splat) are split
        mb = 1 << 20
        overhang = bs * 5 * ((self.model.ksize - 1) // 2) * w * 4          # the splat state's rows: ONE message (merge kernel)
        slot = min(64 * mb, max(mb, -(-max(9 * self.model.width * w * 4 * bs, overhang) // mb) * mb))
        part.channel = HaloChannel.connect(part, device, slot)

    def check(self):
        """Raises if the halo transport has reported a time-out (synchronises the device)."""
        if self.part.channel is not None:
            self.part.channel.check()

    @property
    def transport(self):
        """How neighbour rows travel right now: "ipc" (mailboxes, csrc/halo.hip) or "p2p" (torch.distributed)."""
        return "ipc" if self.part.channel is not None else "p2p"

    def settle_transport(self):
        """COLLECTIVE (every rank of the partition calls it, e.g. after a step raised or at a checkpoint): has any
        rank's mailbox recorded a time-out?  Then every rank drops the IPC channel together and the frame goes on
        over torch.distributed P2P (RCCL) -- a slower step instead of a dead run.  Returns the transport in use;
        `transport_note` says why a fallback happened."""
        part = self.part
        bad, why = 0.0, ""
        if part.channel is not None:
            try:
                part.channel.check()
            except RuntimeError as e:
                bad, why = 1.0, str(e)
        if part.world > 1:
            flag = _all_reduce_sum(th.tensor([bad]), part)
            bad = float(flag.cpu().item())
        if bad > 0 and part.channel is not None:
            ch, part.channel = part.channel, None
            th.cuda.synchronize(ch.device)
            if part.world > 1:
                dist.barrier(group=part.group)         # nobody still stores into a mailbox that is about to go
            ch._finalizer()
            self.transport_note = "fell back from ipc to p2p: " + (why or "a neighbouring rank reported a time-out")
        return self.transport

    def close(self):
        """Removes the gradient hooks this runner registered on the model's parameters and frees its flat gradient
        buffer (a second runner on the same model would otherwise leave both sets of hooks firing)."""
        for h in getattr(self, "_hook_handles", []):
            h.remove()
        self._hook_handles = []
        self._flat = None

    def forward(self, batch):
        from . import wbank
        with wbank.installed(self.model.weight_banks(batch["radiance"])):
            return self._forward(batch)

    def _forward(self, batch):
        m, part = self.model, self.part
        radiance = batch["radiance"]
        features = batch["features"].to(radiance.device)
        gfeatures = batch["global_features"].to(radiance.device)
        if m.pixel:
            radiance = radiance.mean(1, keepdim=True)
            features = features.mean(1, keepdim=True)
        bs, spp, nf, h, w = features.shape
        if not self._channel_tried:
            self._connect(radiance.device, bs, w)
        context = gfeatures
        for step in range(m.nsteps):
            features, reduced = m._embed(getattr(m, "embedding_{:02d}".format(step)), features, context,
                                         want_mean=True)
            context = sharded_autoencoder(getattr(m, "propagation_{:02d}".format(step)), reduced, part)

        p = (m.ksize - 1) // 2
        if self.merge_state:
            # every rank splats its own samples; the p overhang rows of the state cross the link
            slab = (p if part.has_up else 0, p if part.has_down else 0, not part.has_up, not part.has_down)
            sum_r, sum_w, max_w = m._predict_and_splat(features, context, radiance.contiguous(), slab=slab)
            sum_r, sum_w, max_w = merge_overhang(sum_r, sum_w, max_w, p, part)
            output = sum_r / (sum_w + m.eps)
            # only the invalid border of the true image goes (reference models.py:215-216)
            top = 0 if part.has_up else p
            bot = 0 if part.has_down else p
            return {"radiance": output[..., top:output.shape[-2] - bot, p:-p]}
        features = halo_pad(features, p, part)
        context = halo_pad(context, p, part)
        radiance = halo_pad(radiance, p, part)
        sum_r, sum_w, max_w = m._predict_and_splat(features, context, radiance.contiguous())
        output = sum_r / (sum_w + m.eps)
        # p rows per side go in every case: the halo at an inner boundary, the invalid border
        # (reference models.py:215-216) at the true image border
        return {"radiance": output[..., p:-p, p:-p]}

    __call__ = forward

    def target_rows(self, target):
        """The rows (and columns) of this rank's target slab matching `forward`'s output."""
        p = (self.model.ksize - 1) // 2
        top = 0 if self.part.has_up else p
        bot = 0 if self.part.has_down else p
        return target[..., top:target.shape[-2] - bot, p:-p]

    #: bytes of gradients per all-reduce: the flat gradient buffer (139 MB for Multisteps(93, 3)) is summed in
    #: buckets of about this size, each as soon as the backward has produced its gradients
    BUCKET_BYTES = 48 << 20

    def _flat_grads(self):
        """One flat fp32 buffer: [halo status | a slot per parameter gradient | the loss].  It is cut into
        contiguous buckets at parameter boundaries (registration order: the backward fills the buffer from its
        end); a bucket's cross-rank sum starts when its last gradient arrives (post-accumulate hooks), so only
        the last bucket's all-reduce is not hidden behind the rest of the backward.  The loss rides with the bucket
        the backward completes first, the halo transport's time-out flag with the one it completes LAST (every
        exchange of the step is behind it by then): all ranks learn of a time-out together."""
        if self._flat is None:
            self._params = [q for q in self.model.parameters() if q.requires_grad]
            n = sum(q.numel() for q in self._params) + 1
            self._flat = th.zeros(n + 1, dtype=th.float32, device=self._params[0].device)
            self._views, off = [], 1
            self._buckets = []                       # [first param, last param + 1, first element, last element + 1]
            for i, q in enumerate(self._params):
                self._views.append(self._flat[off:off + q.numel()].view_as(q))
                if not self._buckets or (off - self._buckets[-1][2]) * 4 >= self.BUCKET_BYTES:
                    self._buckets.append([i, i, off, off])
                off += q.numel()
                self._buckets[-1][1], self._buckets[-1][3] = i + 1, off
            self._buckets[0][2] = 0                  # the halo status rides with the bucket the backward completes last
            self._buckets[-1][3] = n + 1             # the loss rides with the bucket the backward completes first
            self._bucket_of = {}
            for b, (i0, i1, _, _) in enumerate(self._buckets):
                for i in range(i0, i1):
                    self._bucket_of[i] = b
            self._in_step = False
            for i, q in enumerate(self._params):
                self._hook_handles.append(q.register_post_accumulate_grad_hook(lambda _q, i=i: self._grad_arrived(i)))
        return self._flat

    def _grad_arrived(self, i):
        if not self._in_step:
            return
        b = self._bucket_of[i]
        self._missing[b] -= 1
        if self._missing[b] == 0:
            self._reduce_bucket(b)

    def _reduce_bucket(self, b):
        """The bucket's gradients into the flat buffer (ONE multi-tensor copy; the backward runs with
        `.grad = None`, so autograd hands every gradient over without an accumulation pass), every `.grad`
        re-pointed at its slot (clipping and the optimizer then read the summed values in place), and the
        cross-rank sum of the bucket's slice started."""
        i0, i1, e0, e1 = self._buckets[b]
        params, views = self._params[i0:i1], self._views[i0:i1]
        have = [(v, q.grad) for q, v in zip(params, views) if q.grad is not None]
        if have:
            th._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for q, v in zip(params, views):
            if q.grad is None:
                v.zero_()
            q.grad = v
        self._missing[b] = -1
        if b == 0 and self.part.channel is not None and self._flat.is_cuda:
            ch = self.part.channel
            with th.cuda.device(ch.device):
                _lib.check(ch.lib.sbmc_halo_status_to(ch.box, _lib.ptr(self._flat), _lib.current_stream(ch.device)),
                           "sbmc_halo_status_to")
        work = _all_reduce_sum_start(self._flat[e0:e1], self.part)
        if work is not None:
            self._works.append(work)

    def train_step(self, optimizer, loss_fn, batch, clip=1000):
        """The reference training step (sbmc/interfaces.py:78-105) on the sharded frame.
        `loss_fn` must be a mean over pixels (all of sbmc_amd.losses are)."""
        part = self.part
        flat = self._flat_grads()
        for q in self._params:
            q.grad = None                              # == optimizer.zero_grad(set_to_none=True)
        self._missing = [i1 - i0 for i0, i1, _, _ in self._buckets]
        self._works = []
        out = self.forward(batch)["radiance"]
        tgt = self.target_rows(batch["target_image"])
        # this rank's share of the global mean: the frame's output size is known from the partition (the
        # model crops (ksize-1)/2 pixels on every side), no collective and no host synchronisation needed
        p = (self.model.ksize - 1) // 2
        total = out.shape[0] * out.shape[1] * (part.height - 2 * p) * out.shape[-1]
        loss = loss_fn(out, tgt) * (out.numel() / float(total))
        flat[-1] = loss.detach()
        self._in_step = True
        try:
            loss.backward()                            # (buckets are summed across ranks as they complete)
        finally:
            self._in_step = False
        for b in range(len(self._buckets) - 1, -1, -1):
            if self._missing[b] >= 0:                  # holds a parameter the loss does not depend on
                self._reduce_bucket(b)
        # (what of the all-reduces the backward did not hide: the stream waits here)
        with funcs._timed("grad_all_reduce_exposed", flat.device) if flat.is_cuda else _NOTHING:
            for work in self._works:
                work.wait()
        total = flat[-1]
        status, value = th.stack([flat[0], total]).tolist()     # the step's one host synchronisation
        if status != 0:
            # some rank's mailbox timed out during this step (its neighbour gone, or -- several ranks sharing ONE
            # device in a dry run -- starved): every rank is here with the same flag.  Leave the transport together.
            flat[0] = 0.0
            self.settle_transport()
            raise HaloTimeout("a halo mailbox wait timed out on %d rank(s) during this step; the ranks now exchange over "
                              "torch.distributed P2P -- repeat the step" % int(round(status)))
        if not (value == value and abs(value) != float("inf")):  # (reference guard, sbmc/interfaces.py:95-99)
            raise RuntimeError("non-finite loss")
        th.nn.utils.clip_grad_norm_(self.model.parameters(), clip)
        optimizer.step()
        return total.clone()
