"""Pin the oracle: run the reference's OWN unit tests on top of it.

TEST INFRASTRUCTURE, AUTHORING CONTAINER ONLY (needs /root/reference).

Executes, unmodified and from where they lie, the reference test files
  tests/test_functions.py   (KernelWeighting / Scatter2Gather KATs + gradcheck)
  tests/test_modules.py     (ConvChain, KernelApply, ProgressiveKernelApply KATs)
  tests/test_losses.py
with ``sbmc.halide_ops`` := the C oracle (see refload.py).  Prints the unittest
summary; exit status 0 iff every reference test passes.

    python -m oracle.pin_against_reference            # all (S2G delta test ~ minutes)
    python -m oracle.pin_against_reference --fast     # skips the 38k-call S2G sweep
"""
import importlib.util
import os
import sys
import unittest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refload  # noqa: E402


def main():
    fast = "--fast" in sys.argv
    refload.load_reference()
    suite = unittest.TestSuite()
    loader = unittest.TestLoader()
    for name in ("test_functions", "test_modules", "test_losses"):
        path = os.path.join(refload.REFERENCE_ROOT, "tests", name + ".py")
        spec = importlib.util.spec_from_file_location("ref_" + name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        tests = loader.loadTestsFromModule(mod)
        if fast:
            def keep(t):
                return "test_scatter2gather_cpu" not in t.id()
            flat = []

            def walk(s):
                for t in s:
                    if isinstance(t, unittest.TestSuite):
                        walk(t)
                    else:
                        flat.append(t)
            walk(tests)
            tests = unittest.TestSuite([t for t in flat if keep(t)])
        suite.addTests(tests)
    res = unittest.TextTestRunner(verbosity=2).run(suite)
    sys.exit(0 if res.wasSuccessful() else 1)


if __name__ == "__main__":
    main()
