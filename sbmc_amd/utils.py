"""Small host-side helpers (stand-ins for the two torch-tools symbols the hot path uses)."""
import logging
import os


def get_logger(name):
    return logging.getLogger(name)


def crop_like(src, tgt):
    """Centre-crops the last two dims of ``src`` to those of ``tgt``.

    Stand-in for ``ttools.modules.image_operators.crop_like`` (torch-tools 0.0.36,
    not in the reference tree -- parity of this helper is unpinned; in Multisteps
    the call is an identity, reference models.py:98-102,206).
    """
    sh, sw = src.shape[-2:]
    th_, tw = tgt.shape[-2:]
    if (sh, sw) == (th_, tw):
        return src
    dy, dx = (sh - th_) // 2, (sw - tw) // 2
    if dy < 0 or dx < 0:
        raise ValueError("crop_like: source is smaller than the target")
    return src[..., dy:dy + th_, dx:dx + tw]


def knob(name, default=1):
    """An integer development knob from the environment, read by ONE rule on both sides of the C ABI
    (csrc/common.hpp `env_knob`): unset -> default; "on" / "yes" / "true" -> 1; otherwise the leading integer as C's
    atoi reads it, i.e. "off" / "no" / "false" / the empty string / anything else -> 0 (note: an EMPTY value switches a knob
    off, it does not leave the default).  Whole tokens on both sides: "only" is not "on"."""
    v = os.environ.get(name)
    if v is None:
        return default
    v = v.strip().lower()
    if v in ("on", "yes", "true"):
        return 1
    i, n = 0, len(v)
    if i < n and v[i] in "+-":
        i += 1
    j = i
    while j < n and v[j] in "0123456789":         # (ASCII digits only: str.isdigit takes others that int() refuses)
        j += 1
    return int(v[:j]) if j > i else 0
