"""Round-5 host logic (no GPU): the environment-knob rule shared with the C side, magnitude tags and tagged views on host
tensors, the fixture checksums, the splat-state comparison's float64 fallback."""
import os

import numpy as np
import pytest
import torch as th

from helpers import state_close


def test_knob_rule(monkeypatch):
    from sbmc_amd.utils import knob
    for v, want in ((None, 1), ("0", 0), ("off", 0), ("no", 0), ("false", 0), ("1", 1), ("2", 2), ("on", 1), ("YES", 1),
                    (" 3x", 3), ("", 0), ("-1", -1), ("only", 0), ("on ", 1), ("truest", 0), ("\u0663", 0)):
        if v is None:
            monkeypatch.delenv("SBMC_TEST_KNOB", raising=False)
        else:
            monkeypatch.setenv("SBMC_TEST_KNOB", v)
        assert knob("SBMC_TEST_KNOB") == want, v
    monkeypatch.delenv("SBMC_TEST_KNOB", raising=False)
    assert knob("SBMC_TEST_KNOB", 2) == 2


def test_tags_ride_on_the_object_and_die_with_a_change():
    from sbmc_amd import functions as F
    x = th.randn(2, 3, 4)
    word = th.zeros(1, dtype=th.int32)
    F.tag_amax(x, word)
    assert F.known_amax(x) is word
    assert F.known_amax(x.view(6, 4)) is None             # another object: no tag (tagged_view is what carries it)
    y = F.carry_amax(x.view(6, 4), x)
    assert F.known_amax(y) is word
    assert F.known_amax(F.carry_amax(x[:1], x)) is None   # not the whole tensor: no word
    x.add_(1.0)                                           # any in-place change invalidates (version counter)
    assert F.known_amax(x) is None
    # host tensors take the plain reshape path of tagged_view, gradients included
    a = th.randn(2, 6, requires_grad=True)
    b = F.tagged_view(a, 3, 4)
    b.sum().backward()
    assert tuple(b.shape) == (3, 4) and tuple(a.grad.shape) == (2, 6)


def test_bits_checksum_is_exact_and_order_independent():
    from make_golden import bits_checksum
    g = th.Generator().manual_seed(1)
    x = th.randn(1000, generator=g)
    perm = th.randperm(1000, generator=g)
    assert np.array_equal(bits_checksum(x), bits_checksum(x[perm]))
    y = x.clone()
    y[17] = th.nextafter(y[17], th.tensor(10.0))
    assert not np.array_equal(bits_checksum(x), bits_checksum(y))


def test_state_close_rules():
    sr, sw, mw = th.rand(1, 3, 4, 5) + 1, th.rand(1, 1, 4, 5) + 1, th.randn(1, 1, 4, 5)
    state_close((sr, sw, mw), (sr.clone(), sw.clone(), mw.clone()))
    with pytest.raises(AssertionError):
        state_close((sr, sw, mw + 1e-7), (sr, sw, mw))            # the maximum is a selection: bit-equal
    off = sw.clone()
    off[0, 0, 0, 0] *= 1 + 3e-5
    with pytest.raises(AssertionError):
        state_close((sr, off, mw), (sr, sw, mw))                   # plain elementwise 1e-5
    # ... unless the float64 truth says the fp32 reference itself is that far off there
    truth = (sr.double(), off.double(), mw.double())
    state_close((sr, off, mw), (sr, sw, mw), truth=lambda: truth)
    with pytest.raises(AssertionError):
        state_close((sr, off, mw), (sr, sw, mw), truth=lambda: (sr.double(), sw.double(), mw.double()))


def test_kernel_timing_filter_skips_other_names_without_touching_the_device():
    """bench.py keeps the splat operators' event pairs on inside its timed steps (`only`): every other fused call must
    record nothing -- on the host that also means no CUDA event is ever created."""
    from sbmc_amd import functions as F
    store = []
    F.enable_kernel_timing(store, only=("splat_update_",))
    try:
        with F._timed("conv3x3_fwd 128x128@1x8x8", th.device("cpu")):
            pass
        with F._timed("pointwise_fwd 128x128", th.device("cpu")):
            pass
        assert store == []
    finally:
        F.enable_kernel_timing(None)
    assert F._KERNEL_TIMINGS is None and F._KERNEL_TIMINGS_ONLY is None
    with F._timed("splat_update_fwd_all", th.device("cpu")):        # nothing installed: nothing recorded, nothing created
        pass
