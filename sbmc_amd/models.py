"""The SBMC kernel-splatting denoiser, device-resident on MI355X.

``Multisteps`` has the constructor, sub-module names (``embedding_XX``,
``propagation_XX``, ``kernel_regressor``, ``kernel_update``) and numerics of the
reference's ``sbmc/models.py:35-218``; a reference state-dict loads unchanged.

What differs is *where the tensors live*.  The reference's eval path parks the
per-sample embeddings in host memory and calls ``empty_cache()`` after every
sample (models.py:136-169,195-209) to fit 11 GB GPUs; with 288 GB of HBM3E the
whole [bs, spp, C, H, W] working set stays on the device and both modes share one
batched path.  ``sample_chunk`` bounds the number of samples embedded per
convolution call for very large frames.
"""
import logging

import torch as th
import torch.nn as nn

from . import functions as funcs
from . import modules as ops
from .utils import crop_like, knob

__all__ = ["Multisteps", "KPCN"]

LOG = logging.getLogger(__name__)


class _SampleMean(th.autograd.Function):
    """features.mean(1) over the sample axis.  Backward hands autograd a broadcast VIEW of g / S
    (divided on the small tensor) instead of the materialised [bs, S, c, h, w] tensor torch's
    MeanBackward builds: the sum with the features' other gradient is then one pass, not three."""

    @staticmethod
    def forward(ctx, features):
        ctx.shape = features.shape
        return features.mean(1)

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.shape[1]).unsqueeze(1).expand(ctx.shape)


class Multisteps(nn.Module):
    """Sample-based Monte Carlo denoiser [Gharbi 2019].

    Args:
        n_features(int): per-sample input features.
        n_global_features(int): global (per-image) features.
        width(int): channels of the conv layers.
        embedding_width(int): channels of the per-sample embedding.
        ksize(int): splat kernel size (odd, >= 3).
        splat(bool): predict splat kernels (True) or gather kernels (ablation).
        nsteps(int): sample/pixel coordination steps.
        pixel(bool): average the samples first and treat the result as 1 spp (ablation).
        sample_chunk(int or None): embed at most this many samples per conv call
            (None: all at once).  Not a reference argument; does not change results.
        batch_samples(bool): on GPU, predict the kernels of all samples in one regressor pass
            and splat them with `functions.SplatAll` (3 launches per direction instead of 2-3
            per sample; same values up to fp32 rounding).  Needs the logits of all samples
            resident (1.63 GB per sample at 720p).  Not a reference argument.
        pointwise_gemm(bool): on GPU tensors run the per-sample 1x1 ConvChains as batched GEMMs
            (rocBLAS / hipBLASLt) instead of MIOpen convolutions: same arithmetic, no layout
            change, ~7% faster training step at 720p.  Not a reference argument.
            The same switch makes the U-nets run their 3x3 convolutions without bias followed by
            one fused in-place bias / activation pass per direction (functions.BiasAct / BiasActNHWC),
            and lets them run channels-last where that MEASURES faster (modules.unet_channels_last:
            MIOpen's NHWC solvers, given the find records of sbmc_amd.miopen_db), handing their result
            to the next 1x1 chain without a layout copy.
    """

    def __init__(self, n_features, n_global_features, width=128,
                 embedding_width=128, ksize=21, splat=True, nsteps=3,
                 pixel=False, sample_chunk=None, pointwise_gemm=True, batch_samples=True):
        super(Multisteps, self).__init__()
        if ksize < 3 or (ksize % 2 == 0):
            LOG.error("Kernel size should be odd and > 3.")
            raise ValueError("Kernel size should be odd and > 3.")
        if nsteps < 1:
            LOG.error("Multisteps requires at least one sample/pixel step.")
            raise ValueError("Multisteps requires at least one sample/pixel step.")

        self.ksize = ksize
        self.splat = splat
        self.pixel = pixel
        self.width = width
        self.embedding_width = embedding_width
        self.eps = 1e-8  # kernel normalisation (reference models.py:75)
        self.nsteps = nsteps
        self.sample_chunk = sample_chunk
        self.batch_samples = batch_samples

        for step in range(nsteps):
            n_in = n_features + n_global_features if step == 0 else embedding_width + width
            # per-sample transformation: 1x1 convolutions
            self.add_module("embedding_{:02d}".format(step), ops.ConvChain(
                n_in, embedding_width, width=width, depth=3, ksize=1, pad=False))
            # pixel-space propagation: U-net
            self.add_module("propagation_{:02d}".format(step), ops.Autoencoder(
                embedding_width, width, num_levels=3, increase_factor=2.0,
                num_convs=3, width=width, ksize=3, output_type="leaky_relu",
                pooling="max"))

        self.kernel_regressor = ops.ConvChain(
            width + embedding_width, ksize * ksize, depth=3, width=width, ksize=1,
            activation="leaky_relu", pad=False, output_type="linear")
        self.kernel_update = ops.ProgressiveKernelApply(splat=splat)
        if pointwise_gemm:
            self.kernel_regressor.pointwise_as_gemm = True
            for step in range(nsteps):
                getattr(self, "embedding_{:02d}".format(step)).pointwise_as_gemm = True
                # U-nets: MIOpen convolution + one fused bias / activation pass per direction
                for m in getattr(self, "propagation_{:02d}".format(step)).modules():
                    if isinstance(m, ops.ConvChain):
                        m.fuse_bias_act = True
                getattr(self, "propagation_{:02d}".format(step)).keep_channels_last = True

    def weight_banks(self, like):
        """The step's weight banks (sbmc_amd/wbank.py) for a forward pass on `like`'s device: one per
        embedding + propagation step and one for the kernel regressor -- a bank's backward runs once the LAST of its
        layers has its gradient, so banks per step keep the gradient all-reduce of a sharded frame (sbmc_amd/dist.py)
        overlapped with the rest of the backward.  Empty where the banks do not apply (CPU tensors, other dtypes,
        SBMC_WBANK=0): the modules then run torch's weight norm layer by layer."""
        import os
        from . import wbank
        if not like.is_cuda or knob("SBMC_WBANK") == 0:
            return []
        groups = [[getattr(self, "embedding_{:02d}".format(s)), getattr(self, "propagation_{:02d}".format(s))]
                  for s in range(self.nsteps)] + [[self.kernel_regressor]]
        banks = []
        for mods in groups:
            convs = [c for m in mods for c in m.modules() if wbank.WeightBank.takes(c) and c.weight_v.device == like.device]
            if convs:
                banks.append(wbank.WeightBank(convs))
        return banks

    def _embed(self, module, per_sample, per_pixel, want_mean=False):
        """Runs a 1x1 ConvChain on cat(per_sample[:, s], per_pixel) for every sample s.

        per_sample [bs, spp, c, h, w], per_pixel [bs, c', h, w] -> [bs, spp, e, h, w]; with
        want_mean: (that, its mean over the samples [bs, e, h, w]) -- the mean comes out of the
        chain's last fused layer when it can, whose backward then takes both gradients at once.
        """
        bs, spp, c, h, w = per_sample.shape
        chunk = self.sample_chunk or spp
        outs = []
        mean_out = [] if (want_mean and chunk >= spp) else None
        for s0 in range(0, spp, chunk):
            # (no slicing when all samples go at once: SliceBackward would zero-fill and copy
            # a whole [bs, spp, c, h, w] gradient)
            part = per_sample if chunk >= spp else per_sample[:, s0:s0 + chunk]
            n = part.shape[1]
            out = None
            if module.pointwise_as_gemm:
                # context half of the first layer once per pixel, no concatenation
                out = ops.pointwise_chain_with_context(module, part, per_pixel, mean_out)
            if out is None:
                ctx = per_pixel.expand(bs, per_pixel.shape[1], h, w).unsqueeze(1).expand(
                    bs, n, per_pixel.shape[1], h, w)
                flat = th.cat([part, ctx], 2).reshape(bs * n, c + per_pixel.shape[1], h, w)
                out = module(flat)
            outs.append(funcs.tagged_view(out, bs, n, out.shape[1], h, w))
        features = outs[0] if len(outs) == 1 else th.cat(outs, 1)
        if not want_mean:
            return features
        return features, (mean_out[0] if mean_out else _SampleMean.apply(features))

    def _predict_and_splat(self, features, context, radiance, slab=None):
        """Kernel regression + splat of every sample (reference models.py:193-209).

        features [bs, spp, e, h, w], context [bs, c, h, w], radiance [bs, spp, 3, h, w] ->
        running state (sum_r, sum_w, max_w) after the last sample.
        slab = (top, bot, zero_top, zero_bot): the tensors are one row slab of a frame sharded along
        H (sbmc_amd/dist.py); the state then comes out on top + h + bot rows, see functions.SplatAll.
        """
        bs, spp, _, h, w = features.shape
        top, bot, zero_top, zero_bot = slab or (0, 0, True, True)
        all_kernels = None
        if (self.batch_samples and self.splat and self.kernel_update.fused and radiance.is_cuda
                and radiance.dtype == th.float32
                and funcs.splat_all_supported_dims(radiance.shape[2], self.ksize, top + h + bot, w)):
            # all samples at once: one regressor pass over bs*spp images, three splat launches
            # per direction instead of 2-3 per sample (functions.SplatAll)
            kernels = ops.pointwise_chain_with_context(self.kernel_regressor, features, context) \
                if self.kernel_regressor.pointwise_as_gemm else None
            if kernels is None:
                ctx = context.unsqueeze(1).expand(bs, spp, context.shape[1], h, w)
                flat = th.cat([features, ctx], 2).reshape(bs * spp, -1, h, w)
                kernels = self.kernel_regressor(flat)
            kernels = funcs.tagged_view(kernels, bs, spp, kernels.shape[1], h, w)     # (d_kernels' word travels back)
            supported = (lambda kk: funcs.splat_slab_supported(radiance, kk, top, bot)) if slab else \
                (lambda kk: funcs.splat_all_supported(radiance, kk))
            if kernels.dtype != th.float32 and not supported(kernels):
                kernels = kernels.float()     # half logits where only the fp32 strip kernels apply (k != 21)
            if supported(kernels):
                return funcs.SplatAll.apply(radiance, kernels, top, bot, zero_top, zero_bot)
            all_kernels = kernels             # predicted already: the per-sample loop below consumes them
        # per-sample loop, as the reference.  A row slab is handled by padding: rows that belong to a
        # neighbouring slab get logit -1e30 (weight exp(-1e30 - max) = 0, i.e. no contribution) deep
        # enough that no destination row kept below sees Scatter2Gather's zero fill across an inner edge
        p = (self.ksize - 1) // 2
        pt = 0 if zero_top else top + p
        pb = 0 if zero_bot else bot + p
        sum_r, sum_w, max_w = None, None, None
        for sp in range(spp):
            if all_kernels is not None:
                kernels = all_kernels[:, sp]
            else:
                kernels = self.kernel_regressor(th.cat([features[:, sp], context], 1))
            r = crop_like(radiance[:, sp], kernels)
            if pt or pb:
                kernels = th.nn.functional.pad(kernels, (0, 0, pt, pb), value=-1e30)
                r = th.nn.functional.pad(r, (0, 0, pt, pb))
            sum_r, sum_w, max_w = self.kernel_update(r, kernels, sum_r, sum_w, max_w)
        if slab:
            r0, r1 = pt - top, pt + h + bot
            sum_r, sum_w, max_w = sum_r[..., r0:r1, :], sum_w[..., r0:r1, :], max_w[..., r0:r1, :]
        return sum_r, sum_w, max_w

    def forward(self, samples):
        """
        Args:
            samples(dict): "radiance" [bs, spp, 3, h, w], "features" [bs, spp, nf, h, w],
                "global_features" [bs, ngf, 1, 1].
        Returns:
            dict: "radiance" [bs, 3, h - (ksize-1), w - (ksize-1)] denoised radiance.
        """
        from . import wbank
        with wbank.installed(self.weight_banks(samples["radiance"])):
            return self._forward(samples)

    def _forward(self, samples):
        radiance = samples["radiance"]
        features = samples["features"].to(radiance.device)
        gfeatures = samples["global_features"].to(radiance.device)

        if self.pixel:
            radiance = radiance.mean(1, keepdim=True)
            features = features.mean(1, keepdim=True)

        bs, spp, nf, h, w = features.shape

        # -- alternate per-sample embedding and per-pixel propagation --------------
        # step 0 sees the global features, later steps the propagated pixel context
        # (reference models.py:142-189; batch elements are paired correctly for
        # bs > 1, where the reference's train path mis-tiles them).
        context = gfeatures          # [bs, ngf, 1, 1]: constant over the image at the first step
        for step in range(self.nsteps):
            features, reduced = self._embed(getattr(self, "embedding_{:02d}".format(step)),
                                            features, context, want_mean=True)
            context = getattr(self, "propagation_{:02d}".format(step))(reduced)

        # -- per-sample kernel prediction + progressive splat ----------------------
        sum_r, sum_w, max_w = self._predict_and_splat(features, context, radiance)

        output = sum_r / (sum_w + self.eps)
        crop = (self.ksize - 1) // 2
        output = output[..., crop:-crop, crop:-crop]
        return {"radiance": output}


class KPCN(nn.Module):
    """Kernel-predicting baseline [Bako 2017] (reference models.py:221-291): two 5x5
    ConvChains predict per-pixel gather kernels applied with a softmax KernelApply."""

    def __init__(self, n_in, ksize=21, depth=9, width=100):
        super(KPCN, self).__init__()
        self.ksize = ksize
        chain = dict(depth=depth, width=width, ksize=5, activation="relu",
                     weight_norm=False, pad=False, output_type="linear")
        self.diffuse = ops.ConvChain(n_in, ksize * ksize, **chain)
        self.specular = ops.ConvChain(n_in, ksize * ksize, **chain)
        self.kernel_apply = ops.KernelApply(softmax=True, splat=False)

    def forward(self, data):
        k_diffuse = self.diffuse(data["kpcn_diffuse_in"])
        k_specular = self.specular(data["kpcn_specular_in"])
        b_diffuse = crop_like(data["kpcn_diffuse_buffer"], k_diffuse).contiguous()
        b_specular = crop_like(data["kpcn_specular_buffer"], k_specular).contiguous()
        r_diffuse, _ = self.kernel_apply(b_diffuse, k_diffuse)
        r_specular, _ = self.kernel_apply(b_specular, k_specular)
        albedo = crop_like(data["kpcn_albedo"], r_diffuse)
        radiance = albedo * r_diffuse + (th.exp(r_specular) - 1)
        return dict(radiance=radiance, diffuse=r_diffuse, specular=r_specular)
