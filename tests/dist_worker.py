"""Worker of tests/test_distributed_cpu.py::test_gloo_world2_exchange_rebuilds_frame: one of two
gloo ranks.  Drives the product's ShardedRenderer with a CPU stand-in for the GPU shard: the
oracle renders, numpy packs/unpacks with the host restatement of the kernel's slot mapping."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from importlib import import_module  # noqa: E402

import cases  # noqa: E402
from oracle import Oracle  # noqa: E402

d = import_module("opencl-raytracing_amd.distributed")
rank, world, _ = d.init_process_group("gloo")
assert world == 2 and dist.get_backend() == "gloo"
wl = cases.rt.workloads.get("all_kinds", width=93, height=56)  # ragged: 93 is not a multiple of 8
table = cases.rt.workloads.make_random_table(cases.SEED)
W, H, SPP = wl.width, wl.height, 3


class OracleShard:
    def __init__(self):
        self.accum = torch.zeros((H, W, 4), dtype=torch.float32)
        self.slots = d.shard_slots(W, H, world)
        self.img = None

    def new_packed(self):
        return torch.zeros((self.slots, 4), dtype=torch.float32)

    def trace(self, camera, first, spp):
        s = Oracle().linear_sum(wl.scene, camera, table, W, H, (0, 0, W, H), first, spp).astype(np.float32)
        own = d.tile_owner_map(W, H, world) == rank
        a = np.zeros((H, W, 4), np.float32)
        a[..., :3] = s
        a[..., 3] = spp
        a[~own] = 0
        self.accum.copy_(torch.from_numpy(a))

    def pack(self, out):
        valid, y, x = d.slot_pixels(W, H, rank, world)
        v = self.accum.numpy()[y, x]
        v[~valid] = 0
        out.copy_(torch.from_numpy(v))

    def unpack(self, packed, src):
        valid, y, x = d.slot_pixels(W, H, src, world)
        self.accum.numpy()[y[valid], x[valid]] = packed.numpy()[valid]

    def resolve(self):
        a = self.accum.numpy()
        self.img = np.sqrt(a[..., :3] / np.maximum(a[..., 3:], 1))

    def image(self):
        return self.img


full = Oracle().linear_sum(wl.scene, wl.camera, table, W, H, (0, 0, W, H), 0, SPP).astype(np.float32)
for exchange in ("gather", "reduce"):
    r = d.ShardedRenderer(OracleShard(), rank, world, exchange=exchange)
    r.render(wl.camera, SPP)
    if rank == 0:
        acc = r.shard.accum.numpy()
        assert np.array_equal(acc[..., :3].view(np.uint32), full.view(np.uint32)), exchange
        assert (acc[..., 3] == SPP).all(), exchange
        print("EXCHANGE_OK", exchange, flush=True)
    dist.barrier()
dist.destroy_process_group()
