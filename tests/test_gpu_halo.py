"""The IPC mailbox transport between neighbouring ranks (csrc/halo.hip, sbmc_amd/halo.py) on ONE MI355X:
in loop-back (a rank as its own two neighbours: every kernel, flag and sequence rule, no second process) and
between two processes that map each other's mailbox through HIP IPC."""
import os
import socket
import sys

import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Loop(object):
    """A partition whose rank is its own up and down neighbour."""
    world, rank, has_up, has_down = 3, 1, True, True

    def __init__(self, channel):
        self.channel = channel
        self._agreed = {}


def _loop(slot_bytes=1 << 20, nslots=4):
    from sbmc_amd.halo import HaloChannel
    return _Loop(HaloChannel("cuda:0", slot_bytes, nslots, timeout_s=5.0).loopback())


@pytest.mark.parametrize("shape,nhwc,r", [((1, 8, 12, 20), False, 3), ((2, 8, 12, 20), True, 3),
                                          ((1, 2, 3, 5, 7, 9), False, 2), ((1, 16, 6, 32), True, 1),
                                          ((1, 5, 6, 33), False, 3)])
@pytest.mark.parametrize("dtype", [th.float32, th.float16])
def test_halo_pad_loopback(shape, nhwc, r, dtype):
    """_HaloPad through the mailboxes, forward and backward, vs the closed form of the loop-back wiring:
    what goes down comes back from above.  Shapes: planar / channels-last, batch > 1, 6-d, odd widths
    (4- and 2-byte units instead of 16), 16 > nslots exchanges in a row (slot reuse + acks)."""
    from sbmc_amd import dist as sdist
    part = _loop()
    g = th.Generator().manual_seed(1)
    for it in range(9):
        x = th.randn(shape, generator=g).to("cuda", dtype)
        if nhwc:
            x = x.contiguous(memory_format=th.channels_last)
        x.requires_grad_()
        y = sdist.halo_pad(x, r, part, nhwc)
        want = th.cat([x[..., -r:, :], x, x[..., :r, :]], -2)
        assert th.equal(y, want.detach()), it
        gy = th.randn(y.shape, generator=g).to("cuda", dtype)
        y.backward(gy)
        h = shape[-2]
        gx = gy[..., r:r + h, :].clone()
        gx[..., :r, :] += gy[..., h + r:, :]
        gx[..., h - r:, :] += gy[..., :r, :]
        assert th.equal(x.grad, gx), it
    part.channel.check()


@pytest.mark.parametrize("shape,nhwc,r", [((1, 8, 12, 20), True, 1), ((1, 8, 12, 20), False, 1), ((2, 4, 9, 16), True, 2),
                                          ((1, 4, 7, 16), True, 2)])
def test_halo_refresh_in_place_loopback(shape, nhwc, r):
    """`halo_refresh`: the stale outer rows of a halo-padded map renewed in place (no copy of the slab), and its
    adjoint in place on the gradient; vs the closed form of the loop-back wiring.  (1, 4, 7, 16) with r = 2: three
    interior rows, both neighbours' gradients land on the middle one."""
    from sbmc_amd import dist as sdist
    part = _loop()
    g = th.Generator().manual_seed(3)
    b, c, hp, w = shape
    h = hp - 2 * r
    fmt = th.channels_last if nhwc else th.contiguous_format
    for it in range(5):
        base = th.randn(shape, generator=g).cuda().contiguous(memory_format=fmt)
        x = base.clone(memory_format=fmt).requires_grad_()
        xin = x * 1.0                                  # a non-leaf the refresh may write into
        y = sdist.halo_refresh(xin, r, part, nhwc)
        assert y is not None and y.data_ptr() == xin.data_ptr()
        want = base.clone()
        want[..., :r, :] = base[..., h:h + r, :]       # what went down (my last interior rows) comes from above
        want[..., r + h:, :] = base[..., r:2 * r, :]
        assert th.equal(y.detach(), want), it
        gy = th.randn(shape, generator=g).cuda().contiguous(memory_format=fmt)
        keep = gy.clone()
        y.backward(gy)
        gx = keep.clone()
        gx[..., r:2 * r, :] += keep[..., r + h:, :]
        gx[..., h:h + r, :] += keep[..., :r, :]
        gx[..., :r, :] = 0
        gx[..., r + h:, :] = 0
        assert th.allclose(x.grad, gx, rtol=0, atol=1e-6), it
    part.channel.check()


def test_halo_pad_loopback_thin_slab_and_split_messages():
    """h < 2r: both neighbours' gradients land on the same rows; a slot smaller than the message: the run is
    split into several messages on both sides."""
    from sbmc_amd import dist as sdist
    part = _loop(slot_bytes=4096, nslots=2)
    g = th.Generator().manual_seed(2)
    x = th.randn(2, 7, 4, 24, generator=g).cuda().requires_grad_()
    r = 3
    y = sdist.halo_pad(x, r, part)
    assert th.equal(y, th.cat([x[..., -r:, :], x, x[..., :r, :]], -2).detach())
    gy = th.randn(y.shape, generator=g).cuda()
    y.backward(gy)
    gx = gy[..., r:r + 4, :].clone()
    gx[..., :r, :] += gy[..., 4 + r:, :]
    gx[..., 4 - r:, :] += gy[..., :r, :]
    assert th.allclose(x.grad, gx, rtol=0, atol=1e-6)
    # one big channels-last image in a 4 KB slot: byte-range pieces of a single chunk
    x = th.randn(1, 16, 8, 40, generator=g).cuda().contiguous(memory_format=th.channels_last)
    y = sdist.halo_pad(x, 2, part, True)
    assert th.equal(y, th.cat([x[..., -2:, :], x, x[..., :2, :]], -2))
    part.channel.check()


@pytest.mark.parametrize("rows,p,c", [(24, 10, 3), (12, 10, 3), (10, 10, 1), (16, 2, 4)])
def test_merge_overhang_loopback(rows, p, c):
    """The one-kernel merge of the splat state across slab boundaries (+ its adjoint) vs the torch
    composition `_merge_rows` (reference sbmc/modules.py:450-471) under the same loop-back wiring, including
    slabs thinner than 2p (rows reached by both neighbours) and exact ties of the running maximum."""
    from sbmc_amd import dist as sdist
    part = _loop()
    g = th.Generator().manual_seed(rows)
    bs, w = 2, 70
    hd = rows + 2 * p
    ext = th.randn(bs, c + 2, hd, w, generator=g)
    ext[:, c] = ext[:, c].abs() + 0.1
    ext[:, c + 1, :p] = ext[:, c + 1, rows:rows + p]            # ties: overhang max == own max on some rows
    ext[:, c + 1, rows + p:, ::3] = ext[:, c + 1, p:2 * p, ::3]
    gout = th.randn(bs, c + 2, rows, w, generator=g).cuda()

    a = ext.clone().cuda().requires_grad_()
    out = sdist._MergeOverhangChannel.apply(a, p, part)
    out.backward(gout)

    b = ext.clone().cuda().requires_grad_()
    own, from_up, from_down = b[..., p:p + rows, :], b[..., hd - p:, :], b[..., :p, :]
    ref = sdist._merge_rows(own, from_up, 0, p, c)
    ref = sdist._merge_rows(ref, from_down, rows - p, rows, c)
    ref.backward(gout)
    assert th.allclose(out, ref, rtol=1e-6, atol=1e-6)
    assert th.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6)
    part.channel.check()


def test_wait_times_out_instead_of_hanging():
    """A `get` nobody sends to gives up after the time-out and the host hears about it."""
    from sbmc_amd.halo import HaloChannel, rows_run
    ch = HaloChannel("cuda:0", 1 << 16, 2, timeout_s=0.2).loopback()
    dst = th.zeros(1, 4, 2, 8, device="cuda")
    ch.get(up=rows_run(dst, 0, 2))
    th.cuda.synchronize()
    with pytest.raises(RuntimeError, match="gave up waiting"):
        ch.check()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _two_rank_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sbmc_amd import dist as sdist
        from sbmc_amd.halo import HaloChannel
        dev = th.device("cuda", 0)
        th.cuda.set_device(dev)
        h, w, r = 24, 40, 3
        part = sdist.SlabPartition(h, world, rank)
        part.channel = HaloChannel.connect(part, dev, 1 << 20)
        assert part.channel is not None, "IPC mailboxes between two processes on one GPU"
        g = th.Generator().manual_seed(5)
        for nhwc in (False, True):
            for it in range(6):
                full = th.randn(1, 8, h, w, generator=g).to(dev)
                gfull = th.randn(1, 8, h + 2 * r * (world - 1), w, generator=g).to(dev)   # padded rows of all ranks
                x = full[..., part.y0:part.y1, :].contiguous()
                if nhwc:
                    x = x.contiguous(memory_format=th.channels_last)
                x.requires_grad_()
                y = sdist.halo_pad(x, r, part, nhwc)
                lo, hi = max(part.y0 - r, 0), min(part.y1 + r, h)
                assert th.equal(y, full[..., lo:hi, :]), (rank, nhwc, it)
                # backward: rank k's padded gradient is rows [off_k, off_k + rows_k + halos) of gfull
                offs, o = [], 0
                for k in range(world):
                    pk = sdist.SlabPartition(h, world, k)
                    n = pk.rows + r * (int(pk.has_up) + int(pk.has_down))
                    offs.append((o, n, pk))
                    o += n
                o, n, _ = offs[rank]
                y.backward(gfull[..., o:o + n, :])
                want = th.zeros_like(full)
                for o, n, pk in offs:
                    a = max(pk.y0 - r, 0)
                    want[..., a:a + n, :] += gfull[..., o:o + n, :]
                assert th.allclose(x.grad, want[..., part.y0:part.y1, :], rtol=0, atol=1e-6), (rank, nhwc, it)
        part.channel.check()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_processes_exchange_through_ipc_mailboxes():
    mp.spawn(_two_rank_worker, args=(2, _free_port()), nprocs=2, join=True)


def _stress_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sbmc_amd import dist as sdist
        from sbmc_amd.halo import HaloChannel, rows_run
        dev = th.device("cuda", 0)
        th.cuda.set_device(dev)
        part = sdist.SlabPartition(4 * world, world, rank)
        ch = HaloChannel.connect(part, dev, 1 << 16, nslots=2)       # two slots: every third message waits for an ack
        assert ch is not None
        rng = th.Generator().manual_seed(9)                            # the same sizes on every rank
        bad = th.zeros((), device=dev)
        for it in range(1500):
            planes = int(th.randint(1, 9, (1,), generator=rng))
            w = 4 * int(th.randint(1, 300, (1,), generator=rng))
            # what rank r sends in exchange `it`: a ramp offset by 1000 r + it (fp32-exact)
            src = (th.arange(planes * 2 * w, device=dev, dtype=th.float32).view(1, planes, 2, w) + (1000.0 * rank + it))
            up = th.full((1, planes, 2, w), -1.0, device=dev)
            down = th.full((1, planes, 2, w), -1.0, device=dev)
            run = rows_run(src, 0, 2)
            ch.put(run if part.has_up else None, run if part.has_down else None)
            ch.get(rows_run(up, 0, 2) if part.has_up else None, rows_run(down, 0, 2) if part.has_down else None)
            if part.has_up:
                bad += (up != src - 1000.0).any()
            if part.has_down:
                bad += (down != src + 1000.0).any()
        assert bad.item() == 0, "%d corrupted exchanges on rank %d" % (int(bad.item()), rank)
        ch.check()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_three_processes_1500_exchanges_in_a_row():
    """Three ranks (the middle one has both neighbours), 1500 back-to-back exchanges of random sizes through
    two-slot rings, no host synchronisation in between: every flag, ack and slot-reuse rule under load; any
    stale or torn row would show in the data."""
    mp.spawn(_stress_worker, args=(3, _free_port()), nprocs=3, join=True)
