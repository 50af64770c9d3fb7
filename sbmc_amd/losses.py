"""Losses of the SBMC training step (API of the reference's ``sbmc/losses.py``).

``TonemappedRelativeMSE`` is the training loss (interfaces.py:49), ``RelativeMSE``
the reported metric (:50); the other two complete the reference's ``__all__``.
"""
import torch as th

__all__ = ["RelativeMSE", "SMAPE", "TonemappedMSE", "TonemappedRelativeMSE"]


def _tonemap(im):
    """Reinhard curve x / (1 + x) on the non-negative part (reference losses.py:111-121)."""
    im = th.clamp(im, min=0)
    return im / (1 + im)


class RelativeMSE(th.nn.Module):
    """0.5 * mean((im - ref)^2 / (ref^2 + eps))   (reference losses.py:26-51)."""

    def __init__(self, eps=1e-2):
        super(RelativeMSE, self).__init__()
        self.eps = eps

    def forward(self, im, ref):
        err = (im - ref) ** 2 / (ref ** 2 + self.eps)
        return 0.5 * err.mean()


class SMAPE(th.nn.Module):
    """mean(|im - ref| / (eps + |im| + |ref|)), denominator detached (reference losses.py:54-73)."""

    def __init__(self, eps=1e-2):
        super(SMAPE, self).__init__()
        self.eps = eps

    def forward(self, im, ref):
        denom = self.eps + im.detach().abs() + ref.detach().abs()
        return ((im - ref).abs() / denom).mean()


class TonemappedMSE(th.nn.Module):
    """0.5 * mean((t(im) - t(ref))^2)   (reference losses.py:76-91)."""

    def __init__(self, eps=1e-2):
        super(TonemappedMSE, self).__init__()
        self.eps = eps

    def forward(self, im, ref):
        return 0.5 * ((_tonemap(im) - _tonemap(ref)) ** 2).mean()


class TonemappedRelativeMSE(th.nn.Module):
    """0.5 * mean((t(im) - t(ref))^2 / (t(ref)^2 + eps))   (reference losses.py:94-108)."""

    def __init__(self, eps=1e-2):
        super(TonemappedRelativeMSE, self).__init__()
        self.eps = eps

    def forward(self, im, ref):
        im, ref = _tonemap(im), _tonemap(ref)
        err = (im - ref) ** 2 / (ref ** 2 + self.eps)
        return 0.5 * err.mean()
