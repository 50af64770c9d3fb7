import os, torch as th, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dev = th.device("cuda", 0); th.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = th.ones(1 << 20, device=dev); dist.all_reduce(t); dist.barrier(); th.cuda.synchronize()
d = th.tensor([1.0, 2.0], dtype=th.float64, device=dev); dist.all_reduce(d, op=dist.ReduceOp.MAX)
parts = [th.empty_like(t)]; dist.all_gather(parts, t)
print("rccl single-rank ok", t[0].item(), d.tolist(), dist.get_backend())
dist.destroy_process_group()
