#!/usr/bin/env python3
"""World-size-1 RCCL rehearsal of the collectives ShardedRenderer issues (API acceptance on this image:
gather with a tensor list on dst, reduce, barrier, on a non-default stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    send = torch.arange(4096 * 4, dtype=torch.float32, device="cuda").reshape(4096, 4)
    recv = [torch.empty_like(send)]
    dist.gather(send, recv, dst=0)
    acc = torch.ones((64, 64, 4), dtype=torch.float32, device="cuda")
    dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)
    dist.barrier()
torch.cuda.synchronize()
assert torch.equal(recv[0], send) and float(acc.sum()) == 64 * 64 * 4
print("RCCL world-1 gather/reduce/barrier OK", dist.get_backend(), flush=True)
dist.destroy_process_group()
