"""The sharded denoiser with the DEVICE kernels: two processes on one MI355X (gloo transport,
halos staged through host memory), k = 21 so that the strip kernels and the all-samples splat run
on halo-padded slabs.  Everything but the RCCL transport itself is exercised.

Bounds: output rows and loss against the single-process fp32 model on the same GPU; parameter gradients (sums
over all pixels, whose order sharding changes) against a float64 evaluation of the model on the CPU
(helpers.multisteps_fp64): within 1e-5 of it or no further from it than twice the single-process fp32
gradient is (helpers.no_worse_than, scales per module)."""
import os
import socket
import sys

import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, height, layout="auto", per_conv=True, transport="ipc", wbank=True):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["SBMC_WBANK"] = "1" if wbank else "0"
    os.environ["SBMC_UNET_LAYOUT"] = layout
    os.environ["SBMC_HALO_TRANSPORT"] = transport
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sbmc_amd import Multisteps, losses
        from sbmc_amd import dist as sdist
        from sbmc_amd.utils import crop_like
        sdist.PER_CONV_HALO_BELOW = 10 ** 9 if per_conv else 0
        dev = th.device("cuda", 0)
        nf, ks, spp, w = 6, 21, 2, 72
        th.manual_seed(3)
        ctor = ((nf, 3), dict(width=8, embedding_width=8, ksize=ks, nsteps=2))
        model = Multisteps(*ctor[0], **ctor[1]).to(dev)
        g = th.Generator().manual_seed(4)
        full = {
            "radiance": th.empty(1, spp, 3, height, w).exponential_(1.0, generator=g).to(dev),
            "features": th.rand(1, spp, nf, height, w, generator=g).to(dev),
            "global_features": th.rand(1, 3, 1, 1, generator=g).to(dev),
            "target_image": th.empty(1, 3, height, w).exponential_(1.0, generator=g).to(dev),
        }
        loss_fn = losses.TonemappedRelativeMSE()
        model.train(True)
        ref_out = model(full)["radiance"]
        ref_loss = loss_fn(ref_out, crop_like(full["target_image"], ref_out))
        ref_loss.backward()
        ref_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        model.zero_grad()
        from helpers import bias_term_sums, module_scales, multisteps_fp64, no_worse_than
        th.set_num_threads(8)
        m64 = multisteps_fp64(model, *ctor).train(True)
        terms = bias_term_sums(m64)
        o64 = m64({k: v.cpu().double() for k, v in full.items()})["radiance"]
        l64 = loss_fn(o64, crop_like(full["target_image"].cpu().double(), o64))
        l64.backward()
        g64 = {k: q.grad for k, q in m64.named_parameters()}

        part = sdist.SlabPartition(height, world, rank)
        slab = {k: (v if k == "global_features" else v[..., part.y0:part.y1, :].contiguous())
                for k, v in full.items()}
        runner = sdist.ShardedDenoiser(model, part)
        p = (ks - 1) // 2
        lo, hi = max(part.y0, p) - p, min(part.y1, height - p) - p
        with th.no_grad():
            out = runner(slab)["radiance"]
        assert out.shape[-2] == hi - lo
        no_worse_than(out, ref_out[..., lo:hi, :], o64[..., lo:hi, :], what="output rows")
        opt = th.optim.SGD(model.parameters(), lr=0.0)
        loss = runner.train_step(opt, loss_fn, slab)
        no_worse_than(loss, ref_loss, l64, what="loss")
        scales = module_scales(g64)
        for k, q in model.named_parameters():
            # 1e-5 of the module's gradient scale from float64, or twice the single-process evaluation's own distance from it;
            # a bias gradient -- an fp32 sum over ~9000 pixel-samples of mixed sign -- also to 8 ulp of the sum of its terms'
            # magnitudes (helpers.bias_term_sums: what an fp32 sum warrants; until round 6 these ran at a blanket 2e-5 / 3 x)
            no_worse_than(q.grad, ref_grads[k], g64[k], rtol=1e-5, what="grad " + k, scale=scales[k], slack=2.0,
                          terms=terms.get(k) if k.endswith(".bias") else None)
        assert (part.channel is not None) == (transport == "ipc")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["ipc", "p2p"])
@pytest.mark.parametrize("layout,per_conv", [("nchw", False), ("nhwc", False), ("nhwc", True), ("nchw", True)])
def test_sharded_denoiser_on_device_kernels(layout, per_conv, transport):
    """layout: the U-nets planar, or channels-last (MIOpen NHWC solvers, NHWC glue kernels in their
    row-slab forms, halo rows sent and received in place) -- forced, so that both run whatever the
    measurement would pick on this box.  per_conv: halo exchange before every convolution (thin slabs) or
    before every chain of three.  transport: neighbour rows through the IPC mailboxes (csrc/halo.hip; the splat
    state then merges in ONE kernel) or through torch.distributed P2P (gloo here: staged through the host)."""
    mp.spawn(_worker, args=(2, _free_port(), 64, layout, per_conv, transport), nprocs=2, join=True)


@pytest.mark.parametrize("transport", ["ipc", "p2p"])
def test_sharded_denoiser_without_the_weight_bank_at_the_tight_bound(transport):
    """SBMC_WBANK=0 (torch's weight norm layer by layer): every parameter gradient, biases included, at 1e-5 of the
    float64 evaluation or twice the single-process evaluation's own distance from it."""
    mp.spawn(_worker, args=(2, _free_port(), 64, "nhwc", False, transport, False), nprocs=2, join=True)


def _fallback_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sbmc_amd import Multisteps, losses
        from sbmc_amd import dist as sdist
        from sbmc_amd.halo import TICKS_PER_SECOND, rows_run
        dev = th.device("cuda", 0)
        height, nf, ks, spp, w = 64, 6, 5, 2, 72
        th.manual_seed(3)
        model = Multisteps(nf, 3, width=8, embedding_width=8, ksize=ks, nsteps=2).to(dev).train()
        g = th.Generator().manual_seed(4)
        full = {"radiance": th.empty(1, spp, 3, height, w).exponential_(1.0, generator=g).to(dev),
                "features": th.rand(1, spp, nf, height, w, generator=g).to(dev),
                "global_features": th.rand(1, 3, 1, 1, generator=g).to(dev),
                "target_image": th.empty(1, 3, height, w).exponential_(1.0, generator=g).to(dev)}
        part = sdist.SlabPartition(height, world, rank)
        slab = {k: (v if k == "global_features" else v[..., part.y0:part.y1, :].contiguous()) for k, v in full.items()}
        runner = sdist.ShardedDenoiser(model, part)
        loss_fn = losses.TonemappedRelativeMSE()
        opt = th.optim.SGD(model.parameters(), lr=0.0)
        first = float(runner.train_step(opt, loss_fn, slab))
        assert runner.transport == "ipc" and runner.settle_transport() == "ipc"      # nothing to fall back from
        grads = {k: q.grad.clone() for k, q in model.named_parameters()}
        if rank == 1:
            # rows that nobody ever sends: this rank's mailbox records a time-out (0.3 s) -- what a dead or
            # desynchronised neighbour looks like from here
            ch = part.channel
            ch.timeout_ticks = int(0.3 * TICKS_PER_SECOND)
            junk = th.empty(1, 1, 1, 64, device=dev)
            ch.get(rows_run(junk, 0, 1), None)
            th.cuda.synchronize(dev)
            with pytest.raises(RuntimeError):
                runner.check()
        dist.barrier()          # (rank 0 must not start the next step -- and answer that wait -- before it has timed out)
        # the next step: rank 1's waits return at once (its mailbox is poisoned), the flag rides in the last gradient
        # bucket's all-reduce, and BOTH ranks raise at the same point, having left the mailboxes together
        with pytest.raises(sdist.HaloTimeout):
            runner.train_step(opt, loss_fn, slab)
        assert runner.transport == "p2p" and part.channel is None and "fell back" in runner.transport_note
        second = float(runner.train_step(opt, loss_fn, slab))    # ... and the frame goes on over torch.distributed
        assert abs(second - first) <= 1e-6 * abs(first)
        for k, q in model.named_parameters():
            # (two fp32 evaluations of the step -- halo rows through the mailboxes, then over torch.distributed --, each
            # within 1e-5 of the exact gradient: their difference within twice that)
            assert (q.grad - grads[k]).abs().max().item() <= 2 * 1e-5 * max(grads[k].abs().max().item(), 1e-30), k
    finally:
        dist.destroy_process_group()


def test_a_mailbox_time_out_mid_run_falls_back_to_p2p_on_every_rank():
    """The defence around the first real multi-GPU run (bench.validate_sharded does the same): a rank whose halo wait
    timed out makes ALL ranks leave the IPC mailboxes at the next `settle_transport` and continue over
    torch.distributed P2P with the same results."""
    mp.spawn(_fallback_worker, args=(2, _free_port()), nprocs=2, join=True)
