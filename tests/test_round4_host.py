"""Host logic added in round 4 that needs no GPU: the weight bank's applicability, the C structures' layout, the gradient
hooks of a sharded runner going away with it."""
import ctypes

import torch as th


def test_weight_bank_structures_match_the_header():
    """include/sbmc_hip.h sbmc_wbank_entry / sbmc_wbank_grad as csrc/conv3x3.hip compiles them (LP64: 6 pointers + 4 ints;
    6 pointers + 4 longs + 4 ints)."""
    from sbmc_amd import _lib
    assert ctypes.sizeof(_lib.WBankEntry) == 6 * 8 + 4 * 4
    assert ctypes.sizeof(_lib.WBankGrad) == 6 * 8 + 4 * 8 + 4 * 4
    assert _lib.WBANK_MAX == 24


def test_no_bank_on_cpu_tensors_and_nothing_left_on_the_modules(cpu_ops):
    from sbmc_amd import Multisteps, wbank
    model = Multisteps(6, 3, width=8, embedding_width=8, ksize=3, nsteps=1)
    x = {"radiance": th.rand(1, 2, 3, 12, 12), "features": th.rand(1, 2, 6, 12, 12), "global_features": th.rand(1, 3, 1, 1)}
    assert model.weight_banks(x["radiance"]) == []
    assert not any(wbank.WeightBank.takes(m) for m in model.modules())      # (CPU parameters)
    with wbank.installed([]):
        out = model(x)["radiance"]
    assert out.shape == (1, 3, 10, 10)
    assert all("_sbmc_bank_w" not in m.__dict__ for m in model.modules())


def test_sharded_runner_close_removes_its_gradient_hooks():
    from sbmc_amd import Multisteps
    from sbmc_amd import dist as sdist
    model = Multisteps(6, 3, width=8, embedding_width=8, ksize=3, nsteps=1)
    part = sdist.SlabPartition(16, 1, 0)
    a = sdist.ShardedDenoiser(model, part)
    a._flat_grads()
    n = len(a._hook_handles)
    assert n == sum(1 for q in model.parameters() if q.requires_grad) and a.transport == "p2p"
    a.close()
    assert a._hook_handles == [] and a._flat is None
    p = next(model.parameters())
    assert len(p._post_accumulate_grad_hooks or {}) == 0
    assert a.settle_transport() == "p2p"               # world 1, no channel: nothing to settle
