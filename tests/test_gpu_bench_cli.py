"""bench.py as the driver calls it: `python bench.py --gpus N ...` must start its own ranks and print ONE JSON
line on rank 0.  Two ranks on ONE GPU over gloo (test hooks SBMC_BENCH_BACKEND / SBMC_BENCH_SINGLE_DEVICE):
everything but the RCCL transport, on a tiny frame."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--height", "64", "--width", "96", "--spp", "2", "--ksize", "5", "--steps", "2", "--warmup", "2",
         "--no-cpu-baseline", "--no-stages"]


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         env=e, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_carries_the_inference_matrix():
    """north_star's matrix in the default line: forward-only stages at 4 / 8 / 32 spp, the forward kernel's
    roofline entry, the whole step's TFLOP/s and the U-nets' layout (tiny frame here)."""
    d = _run([a for a in SMALL if a != "--no-stages"])
    for spp in (4, 8, 32):
        st = d["stages"]["infer_%dspp" % spp]
        assert st["value"] > 0 and st["unit"] == "Msamples/s" and st["steps"] == 10
    assert d["whole_step_tflops"] > 0 and 0 < d["multiples_of_fp32_mfma_peak"] < 3 and "unet_layout" in d
    assert "roofline" in d and "stages" in d and "splat" in d["stages"]


def test_bench_single_gpu_line():
    d = _run(SMALL)
    assert d["n_gpus"] == 1 and d["world_size"] == 1 and d["value"] > 0 and d["unit"] == "Msamples/s"
    assert d["steps"] == 2 and d["higher_is_better"] is True and d["dtype"] == "f32"
    assert d["ms_per_step_median"] > 0 and len(d["ms_per_step_min_max"]) == 2
    # the roofline kernel is timed INSIDE the training step's timed region: one launch of each splat operator per timed step
    ks = d["kernels_in_step"]
    assert ks["splat_update_bwd_all"]["calls"] == 2 and ks["splat_update_fwd_all"]["calls"] == 2
    assert d["roofline"]["launches"] == 2 and "inside the timed steps" in d["roofline"]["timed"]
    assert d["roofline"]["avg_launch_ms"] == ks["splat_update_bwd_all"]["avg_ms"]


@pytest.mark.parametrize("workload", ["model", "splat", "infer"])
def test_bench_starts_its_own_ranks(workload):
    d = _run(["--gpus", "2", "--workload", workload] + SMALL,
             env={"SBMC_BENCH_BACKEND": "gloo", "SBMC_BENCH_SINGLE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo"
    assert d["value"] > 0 and d["scaling"] == "strong"
    assert "H-slabs x2" in d["config"]["parallelism"]
    if workload in ("model", "infer"):
        # the per-rank split of the sharded step: neighbour exchanges, exposed all-reduce, the rest
        assert [r["rank"] for r in d["per_rank"]] == [0, 1]
        for r in d["per_rank"]:
            assert r["step_ms"] > 0 and r["exchanges"] > 0 and r["exchange_ms"] >= 0
            assert abs(r["compute_ms"] + r["exchange_ms"] + r["all_reduce_exposed_ms"] - r["step_ms"]) < 1e-2
        assert d["whole_step_tflops"] > 0 and "unet_layout" in d
        for r in d["per_rank"]:
            assert r["transport"] in ("ipc", "p2p")
    if workload == "model":
        # the first thing N ranks do is a correctness check against rank 0's single-GPU forward of the whole frame
        v = d["validation"]
        assert v["rel_diff"] is not None and v["rel_diff"] <= 1e-5, v
        assert d["transport"] in ("ipc", "p2p") and d["rccl_ranks"] == 2
