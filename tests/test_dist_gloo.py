"""Multi-rank H-slab path (sbmc_amd/dist.py) on CPU: world_size 2 and 3 over gloo.

Each rank runs its slab of one frame through ShardedDenoiser and compares with the
single-process full-frame Multisteps: outputs on its rows, the (global-mean) loss and the
all-reduced parameter gradients.  The operators run on the oracle via the test hook.

Bounds: outputs and loss 1e-5 of the full-frame fp32 model.  Parameter gradients are sums over all pixels
whose fp32 value depends on the order of addition, which sharding changes: they are held to a float64
evaluation of the same model (helpers.multisteps_fp64) -- within 1e-5 of it, or no further from it than
twice the distance of the full-frame fp32 gradient (helpers.no_worse_than, scales per module)."""
import os
import socket
import sys

import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _close(a, b, rtol, what):
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= rtol * max(scale, 1e-30) + 1e-12, "%s: err %.3e scale %.3e" % (what, err, scale)


def _worker(rank, world, port, height, train, merge_state, per_conv=True, bucket_bytes=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        th.set_num_threads(2)
        from oracle import sbmc_oracle
        from sbmc_amd import Multisteps, halide_ops, losses
        from sbmc_amd import dist as sdist
        from sbmc_amd.utils import crop_like
        halide_ops.register_cpu_ops_for_testing(sbmc_oracle)
        # the U-nets' halo exchange: one row before every convolution (thin slabs) or three before every chain
        sdist.PER_CONV_HALO_BELOW = 10 ** 9 if per_conv else 0
        if bucket_bytes:                              # several gradient buckets, reduced as the backward fills them
            sdist.ShardedDenoiser.BUCKET_BYTES = bucket_bytes

        nf, ks, spp, w = 6, 5, 2, 20
        th.manual_seed(3)
        ctor = ((nf, 3), dict(width=8, embedding_width=8, ksize=ks, nsteps=2))
        model = Multisteps(*ctor[0], **ctor[1])
        g = th.Generator().manual_seed(4)
        full = {
            "radiance": th.empty(1, spp, 3, height, w).exponential_(1.0, generator=g),
            "features": th.rand(1, spp, nf, height, w, generator=g),
            "global_features": th.rand(1, 3, 1, 1, generator=g),
            "target_image": th.empty(1, 3, height, w).exponential_(1.0, generator=g),
        }
        loss_fn = losses.TonemappedRelativeMSE()

        # single-process reference on the full frame
        model.train(train)
        ref_out = model(full)["radiance"]
        ref_loss = loss_fn(ref_out, crop_like(full["target_image"], ref_out))
        if train:
            ref_loss.backward()
            ref_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
            model.zero_grad()
            from helpers import module_scales, multisteps_fp64, no_worse_than
            m64 = multisteps_fp64(model, *ctor).train(True)
            o64 = m64({k: v.double() for k, v in full.items()})["radiance"]
            loss_fn(o64, crop_like(full["target_image"].double(), o64)).backward()
            g64 = {k: q.grad for k, q in m64.named_parameters()}

        part = sdist.SlabPartition(height, world, rank)
        slab = {k: (v if k == "global_features" else v[..., part.y0:part.y1, :].contiguous())
                for k, v in full.items()}
        runner = sdist.ShardedDenoiser(model, part, merge_state=merge_state)
        assert runner.merge_state == merge_state
        p = (ks - 1) // 2
        lo = max(part.y0, p) - p                      # this rank's rows in output coordinates
        hi = min(part.y1, height - p) - p
        if train:
            opt = th.optim.SGD(model.parameters(), lr=0.0)  # lr 0: keep weights, we check grads
            loss = runner.train_step(opt, loss_fn, slab)
            _close(loss, ref_loss.detach(), 1e-5, "loss")
            if bucket_bytes:
                assert len(runner._buckets) > 2
            scales = module_scales(g64)
            for k, q in model.named_parameters():
                no_worse_than(q.grad, ref_grads[k], g64[k], what="grad " + k, scale=scales[k])
        else:
            with th.no_grad():
                out = runner(slab)["radiance"]
            assert out.shape[-2] == hi - lo
            _close(out, ref_out[..., lo:hi, :].detach(), 1e-5, "output rows")
    finally:
        dist.destroy_process_group()


# merge_state=True: every rank splats its own samples and the overhang rows of the running state are
# exchanged and merged; False: the halo-recompute form (inputs padded by the kernel radius)
@pytest.mark.parametrize("world,height,train,merge_state,per_conv", [
    (2, 32, False, True, True), (2, 32, True, True, True), (3, 48, True, True, False), (3, 36, False, True, False),
    (2, 32, True, False, False), (3, 48, False, False, True), (3, 48, True, True, True)])
def test_sharded_denoiser_equals_full_frame(world, height, train, merge_state, per_conv):
    mp.spawn(_worker, args=(world, _free_port(), height, train, merge_state, per_conv), nprocs=world, join=True)


def test_gradient_buckets_are_reduced_as_the_backward_fills_them():
    """The flat gradient buffer cut into many small buckets: every bucket's cross-rank sum starts from a
    post-accumulate hook inside the backward; the result is the same."""
    mp.spawn(_worker, args=(2, _free_port(), 32, True, True, False, 4096), nprocs=2, join=True)


def test_slab_partition():
    from sbmc_amd.dist import SlabPartition
    for h, world in ((720, 8), (2160, 8), (720, 2), (64, 4)):
        parts = [SlabPartition(h, world, r) for r in range(world)]
        assert parts[0].y0 == 0 and parts[-1].y1 == h
        for a, b in zip(parts, parts[1:]):
            assert a.y1 == b.y0
        assert all(p.y0 % 4 == 0 and p.rows > 0 for p in parts)
        assert max(p.rows for p in parts) - min(p.rows for p in parts) <= 4
    with pytest.raises(ValueError):
        SlabPartition(722, 2, 0)
    one = SlabPartition(723, 1, 0)
    assert (one.y0, one.y1) == (0, 723)
