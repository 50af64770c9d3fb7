"""Training interface, dataset and checkpointing for the SBMC denoiser.

`SampleBasedDenoiserInterface` has the constructor, `forward` / `backward` /
`init_validation` / `update_validation` methods and numerics of the reference's
sbmc/interfaces.py:35-132 (Adam(lr), TonemappedRelativeMSE, non-finite guard, grad-norm clip
1000, RelativeMSE metric); the `ttools` base class, trainer and checkpointer the reference
relies on are not in its tree, so minimal stand-ins live here: `TilesDataset` (over
`sbmc_amd.binio`), `Checkpointer` (plain torch.save) and `train` (the epoch loop).
"""
import gc
import glob
import logging
import os

import numpy as np
import torch as th

from . import binio, losses
from .utils import crop_like

__all__ = ["SampleBasedDenoiserInterface", "TilesDataset", "MultiSampleCountDataset", "Checkpointer",
           "train"]

LOG = logging.getLogger(__name__)


class SampleBasedDenoiserInterface(object):
    """Args: model(nn.Module), lr(float), cuda(bool) -- as the reference."""

    def __init__(self, model, lr=1e-4, cuda=False):
        self.device = "cpu"
        self.model = model
        self.loss_fn = losses.TonemappedRelativeMSE()
        self.rmse_fn = losses.RelativeMSE()
        if cuda:
            LOG.debug("Using CUDA")
            self.device = "cuda"
            self.model.cuda()
        # fused=True on the GPU: the same Adam update (reference interfaces.py:58) in one kernel
        self.optimizer = th.optim.Adam(self.model.parameters(), lr=lr, fused=bool(cuda))

    def forward(self, batch):
        for k in batch:
            if isinstance(batch[k], th.Tensor):
                batch[k] = batch[k].to(self.device)
        return self.model(batch)

    def backward(self, batch, fwd):
        self.optimizer.zero_grad()
        out = fwd["radiance"]
        tgt = crop_like(batch["target_image"], out)
        loss = self.loss_fn(out, tgt)
        loss.backward()
        value = loss.item()
        if not np.isfinite(value):
            LOG.error("Loss is infinite, there might be outliers in the data.")
            raise RuntimeError("Infinite loss at train time.")
        if np.isnan(value):
            LOG.error("NaN in the loss, there might be outliers in the data.")
            raise RuntimeError("NaN loss at train time.")
        clip = 1000
        actual = th.nn.utils.clip_grad_norm_(self.model.parameters(), clip)
        if actual > clip:
            LOG.info("Clipped gradients {} -> {}".format(clip, actual))
        self.optimizer.step()
        with th.no_grad():
            rmse = self.rmse_fn(out, tgt)
        return {"loss": value, "rmse": rmse.item()}

    def init_validation(self):
        return {"loss": 0.0, "rmse": 0.0, "n": 0}

    def update_validation(self, batch, fwd, running):
        with th.no_grad():
            out = fwd["radiance"]
            tgt = crop_like(batch["target_image"], out)
            loss = self.loss_fn(out, tgt).item()
            rmse = self.rmse_fn(out, tgt).item()
        b = out.shape[0]
        n = running["n"] + b
        return {"loss": running["loss"] - (1.0 / n) * (running["loss"] - b * loss),
                "rmse": running["rmse"] - (1.0 / n) * (running["rmse"] - b * rmse), "n": n}


class TilesDataset(th.utils.data.Dataset):
    """`.bin` tiles under root/<scene>/*.bin (folder mode of the reference's TilesDataset,
    sbmc/datasets.py:243-300).  load_coords / load_gbuffer / load_p / load_ld / load_bt select the per-sample
    feature groups as in the reference (:194-215; "kpcn" mode ignores them)."""

    KEYS = {"sbmc": ("radiance", "features", "global_features", "target_image", "low_spp"),
            "kpcn": ("kpcn_diffuse_in", "kpcn_specular_in", "kpcn_diffuse_buffer", "kpcn_specular_buffer",
                     "kpcn_albedo", "target_image", "low_spp")}

    def __init__(self, path, spp=None, mode="sbmc", load_coords=True, load_gbuffer=True, load_p=True,
                 load_ld=True, load_bt=True):
        if mode not in self.KEYS:
            LOG.error("Unknown dataset loading mode %s", mode)
            raise RuntimeError("Unknown dataset loading mode %s" % mode)
        self.files = sorted(glob.glob(os.path.join(path, "*", "*.bin")))
        if not self.files:
            LOG.error("Dataset is empty, please check the file format / folder structure.")
            raise RuntimeError("Empty dataset")
        self.spp, self.mode = spp, mode
        self.flags = binio.feature_flags(mode, load_coords, load_gbuffer, load_p, load_ld, load_bt)
        self.labels = binio.feature_labels(**self.flags)
        self.num_features = 27 if mode == "kpcn" else len(self.labels)       # datasets.py:413-419
        self.num_global_features = len(binio.GLOBAL_LABELS)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        tile = binio.read_tile(self.files[idx], self.spp, mode=self.mode, **self.flags)
        return {k: th.from_numpy(np.ascontiguousarray(tile[k])) for k in self.KEYS[self.mode]}


class MultiSampleCountDataset(th.utils.data.ConcatDataset):
    """Every tile at every sample count 2..spp (reference sbmc/datasets.py:1015-1043); the
    sample dimension varies between items, so use batch_size = 1."""

    def __init__(self, path, spp=None, mode="sbmc", **flags):
        if spp is None:
            LOG.error("MultiSampleCountDataset requires a number of spps")
            raise RuntimeError("spp not provided.")
        if spp < 2:
            LOG.error("MultiSampleCountDataset needs at least 2spp")
            raise RuntimeError("spp too low to randomize sample count, should be at least 2.")
        parts = [TilesDataset(path, spp=s, mode=mode, **flags) for s in range(2, spp + 1)]
        super(MultiSampleCountDataset, self).__init__(parts)
        self.num_features = parts[0].num_features
        self.num_global_features = parts[0].num_global_features


class Checkpointer(object):
    """Keeps `<dir>/training_end.pth` / `epoch_XXXX.pth`: {"model", "optimizer", "meta", "epoch"}."""

    def __init__(self, root, model, optimizer=None, meta=None):
        self.root, self.model, self.optimizer, self.meta = root, model, optimizer, meta or {}
        os.makedirs(root, exist_ok=True)

    def save(self, name, epoch):
        obj = {"model": self.model.state_dict(), "meta": self.meta, "epoch": epoch}
        if self.optimizer is not None:
            obj["optimizer"] = self.optimizer.state_dict()
        th.save(obj, os.path.join(self.root, name + ".pth"))

    def load_latest(self):
        files = sorted(glob.glob(os.path.join(self.root, "*.pth")), key=os.path.getmtime)
        if not files:
            return None, None
        obj = th.load(files[-1], map_location="cpu")
        self.model.load_state_dict(obj["model"])
        if self.optimizer is not None and "optimizer" in obj:
            self.optimizer.load_state_dict(obj["optimizer"])
        return {"epoch": obj.get("epoch", 0)}, obj.get("meta", {})

    @staticmethod
    def load_meta(root):
        files = sorted(glob.glob(os.path.join(root, "*.pth")), key=os.path.getmtime)
        if not files:
            raise RuntimeError("no checkpoint in %s" % root)
        return th.load(files[-1], map_location="cpu").get("meta", {})


def train(interface, dataloader, num_epochs=1, val_dataloader=None, checkpointer=None,
          start_epoch=0, log_every=10, validation_log=None):
    """-> the per-step statistics.  validation_log: a list that receives each epoch's validation means."""
    history = []
    frozen = False
    for epoch in range(start_epoch, num_epochs):
        interface.model.train(True)
        for it, batch in enumerate(dataloader):
            stats = interface.backward(batch, interface.forward(batch))
            history.append(stats)
            if not frozen:
                # everything alive after the first step (modules, parameters, optimizer state, the
                # MIOpen / HIP runtime's Python side) stays for the whole run: take it out of the
                # cyclic collector's reach, so that a full collection -- 50 ms of host time with a
                # model of this size, which lands right after the step's loss.item() sync where
                # the GPU is waiting for the host -- only ever walks the objects of a few steps
                gc.collect()
                gc.freeze()
                frozen = True
            if it % log_every == 0:
                LOG.info("epoch %d it %d loss %.5f rmse %.5f", epoch, it, stats["loss"], stats["rmse"])
        if val_dataloader is not None:
            interface.model.train(False)
            running = interface.init_validation()
            with th.no_grad():
                for batch in val_dataloader:
                    running = interface.update_validation(batch, interface.forward(batch), running)
            LOG.info("epoch %d validation loss %.5f rmse %.5f", epoch, running["loss"], running["rmse"])
            if validation_log is not None:
                validation_log.append(dict(running, epoch=epoch))
        if checkpointer is not None:
            checkpointer.save("epoch_%04d" % epoch, epoch + 1)
    if checkpointer is not None:
        checkpointer.save("training_end", num_epochs)
    if frozen:
        gc.unfreeze()
    return history
