"""What can be executed of the RCCL transport on a ONE-GPU box: a single-rank `nccl` process group in which the
neighbour exchange of sbmc_amd.dist is a send/recv to the rank itself, and the gradient all-reduce a one-rank
all-reduce -- the `nccl` branches of `dist._exchange` / `dist._all_reduce_sum` with device tensors, exactly the
calls the multi-GPU run issues (RCCL refuses two ranks on one device, so two-rank tests use gloo)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch as th, torch.distributed as dist
from sbmc_amd import dist as sdist
dev = th.device("cuda", 0)
th.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"

class Loop(object):                       # a slab whose upper and lower neighbour are this very rank
    group, has_up, has_down, world, rank = None, True, True, 1, 0
    def peer(self, delta):
        return 0

part = Loop()
up = th.arange(2 * 3 * 4 * 5, dtype=th.float32, device=dev).view(2, 3, 4, 5)
down = -up - 1.0
# rows that are NOT contiguous in memory (a slice of a channels-last map), as the U-net hands them over
up_cl = up.contiguous(memory_format=th.channels_last)
from_up, from_down = sdist._exchange(part, up_cl, down)
th.cuda.synchronize()
# sends and receives to one peer match in order: what went "up" comes back as the upper neighbour's rows
assert from_up.is_cuda and th.equal(from_up, up) and th.equal(from_down, down)

flat = th.randn(1 << 22, device=dev)
ref = flat.clone()
out = sdist._all_reduce_sum(flat, part)
assert out.data_ptr() == flat.data_ptr() and th.equal(out, ref)          # in place on the device, one rank: identity
host = th.ones(3)
assert sdist._all_reduce_sum(host, part).is_cuda                          # RCCL takes device tensors only
dist.barrier()
dist.destroy_process_group()
print("RCCL-SINGLE-RANK-OK")
"""


def test_exchange_and_all_reduce_over_rccl_single_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, env=env,
                         timeout=600)
    assert res.returncode == 0 and "RCCL-SINGLE-RANK-OK" in res.stdout, (res.stdout + res.stderr)[-3000:]
