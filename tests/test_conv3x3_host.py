"""Host logic around the 3 x 3 convolution kernels that needs no GPU: the largest-magnitude tags that ride on
tensors, and the dispatch conditions (the kernels themselves: tests/test_gpu_conv3x3.py)."""
import torch as th
import torch.nn as nn

from sbmc_amd import _lib
from sbmc_amd import functions as funcs
from sbmc_amd import modules as ops


def _bits(v):
    return th.tensor([v], dtype=th.float32).view(th.int32)


def test_tag_is_bound_to_object_state_and_storage():
    t = th.randn(2, 8, 4, 4)
    assert funcs.known_amax(t) is None
    funcs.tag_amax(t, _bits(t.abs().max().item()))
    assert funcs.known_amax(t) is not None
    assert funcs.known_amax(t.clone()) is None                  # another tensor: no tag
    t.add_(1.0)                                                 # in-place change: the version counter moves
    assert funcs.known_amax(t) is None
    u = th.randn(4)
    funcs.tag_amax(u, _bits(9.0))
    u.set_(th.randn(4).untyped_storage())                             # same object, other storage
    assert funcs.known_amax(u) is None


def test_tags_can_be_switched_off(monkeypatch):
    t = th.randn(3)
    funcs.tag_amax(t, _bits(5.0))
    monkeypatch.setenv("SBMC_AMAX_TAGS", "0")
    assert funcs.known_amax(t) is None


def test_bound_of_several_tensors_is_the_largest_word():
    a, b, c = _bits(0.5), _bits(3.0e4), _bits(1.0e-9)
    out = funcs.bound_amax(a, b, c)
    assert out.view(th.float32).item() == 3.0e4                 # bit patterns of non-negative floats order like ints
    assert funcs.bound_amax(a, None) is None and funcs.bound_amax() is None


def test_dispatch_conditions_without_a_gpu():
    conv = nn.Conv2d(128, 128, 3, padding=1)
    x = th.randn(1, 128, 8, 8).contiguous(memory_format=th.channels_last)
    assert not funcs.Conv3x3NHWC.supported(x, conv)             # CPU tensor: never
    net = ops.Autoencoder(128, 128, num_levels=2, increase_factor=2.0, num_convs=2, width=128)
    assert not ops.unet_channels_last(net, x)                   # (and the layout rule does not fire either)
    L = _lib.lib()
    # what the kernels take: multiples of 32 input / 128 output channels (the adjoint exchanges them)
    assert L.sbmc_conv3x3_supported(1, 720, 1280, 128, 128) == 1
    assert L.sbmc_conv3x3_supported(1, 720, 1280, 384, 128) == 1
    assert L.sbmc_conv3x3_supported(1, 720, 1280, 128, 96) == 0
    assert L.sbmc_conv3x3_supported(0, 720, 1280, 128, 128) == 0
    assert L.sbmc_conv3x3_weights_bytes(128, 128) == 9 * 128 * 128 * 4 + 16     # two half planes = the fp32 bytes, + scale
    assert L.sbmc_conv3x3_wgrad_scratch_bytes(1, 720, 1280, 128, 128) > 0
    assert L.sbmc_conv3x3_wgrad_scratch_bytes(1, 720, 1280, 96, 128) == 0
