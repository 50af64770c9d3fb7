import os
import sys

import pytest

# MIOpen's exhaustive "find" costs minutes per new convolution shape at 1280x720; its FAST mode (what
# bench.py uses) picks solvers from the heuristics in seconds.  Read by MIOpen at the first convolution.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: CPU tests of more than ~10 s (hipcc resource reports, the exhaustive gradcheck): "
                                       "`-m 'not gpu and not slow'` is the pre-commit gate (~1 min), `-m 'not gpu'` the full CPU suite")


@pytest.fixture
def oracle():
    """The CPU oracle (test infrastructure, see oracle/sbmc_oracle.py)."""
    from oracle import sbmc_oracle
    sbmc_oracle.lib()
    return sbmc_oracle


@pytest.fixture
def cpu_ops(oracle):
    """Installs the oracle behind sbmc_amd's `*_cpu_float32` names for the duration of a
    test, so that the Python host logic above the operators can run on host tensors."""
    from sbmc_amd import halide_ops
    halide_ops.register_cpu_ops_for_testing(oracle)
    yield oracle
    halide_ops.register_cpu_ops_for_testing(None)
