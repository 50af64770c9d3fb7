"""Training interface / dataset / checkpointing (row N3) on CPU with the oracle behind the ops."""
import os
import sys

import torch as th
from torch.utils.data import DataLoader

from helpers import GOLDEN, close, multisteps_from_golden

sys.path.insert(0, GOLDEN)


def test_interface_step_matches_reference_fixture(cpu_ops):
    """backward() = the reference training step: same loss and gradients as the fixture captured
    from the reference (loss.backward() before clip / Adam)."""
    from sbmc_amd import interfaces
    g, model, batch = multisteps_from_golden("cpu")
    model.train(True)
    iface = interfaces.SampleBasedDenoiserInterface(model, lr=0.0, cuda=False)
    fwd = iface.forward(batch)
    stats = iface.backward(batch, fwd)
    assert abs(stats["loss"] - float(g["train.loss"])) <= 1e-5 * abs(float(g["train.loss"]))
    for k, p in model.named_parameters():
        close(p.grad, g["grad." + k], rtol=2e-5, what=k)
    run = iface.update_validation(batch, fwd, iface.init_validation())
    assert run["n"] == 1 and abs(run["loss"] - stats["loss"]) < 1e-6


def test_train_loop_dataset_checkpoint(cpu_ops, tmp_path):
    from make_golden import synthetic_scene
    from sbmc_amd import Multisteps, interfaces
    root = tmp_path / "data"
    synthetic_scene(str(root / "scene0"), 32, 16, 16, 2, seed=3)
    data = interfaces.TilesDataset(str(root))
    assert len(data) == 2 and data[0]["features"].shape == (2, 93, 16, 16)
    th.manual_seed(0)
    model = Multisteps(data.num_features, data.num_global_features, ksize=3, width=8, embedding_width=8,
                       nsteps=1)
    iface = interfaces.SampleBasedDenoiserInterface(model, lr=1e-3, cuda=False)
    ckpt = interfaces.Checkpointer(str(tmp_path / "ckpt"), model, iface.optimizer, meta={"a": 1})
    hist = interfaces.train(iface, DataLoader(data, batch_size=1), num_epochs=2,
                            val_dataloader=DataLoader(data, batch_size=1), checkpointer=ckpt)
    assert len(hist) == 4 and all(h["loss"] == h["loss"] for h in hist)
    model2 = Multisteps(data.num_features, data.num_global_features, ksize=3, width=8, embedding_width=8,
                        nsteps=1)
    extras, meta = interfaces.Checkpointer(str(tmp_path / "ckpt"), model2).load_latest()
    assert extras["epoch"] == 2 and meta == {"a": 1}
    for a, b in zip(model.state_dict().values(), model2.state_dict().values()):
        assert th.equal(a, b)
    assert interfaces.Checkpointer.load_meta(str(tmp_path / "ckpt")) == {"a": 1}
    multi = interfaces.MultiSampleCountDataset(str(root), spp=2)
    assert len(multi) == 2 and multi[1]["features"].shape[0] == 2
    import pytest
    with pytest.raises(RuntimeError):
        interfaces.MultiSampleCountDataset(str(root), spp=1)
