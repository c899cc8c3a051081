"""Multi-GPU rendering: one process per GPU, frame tiles interleaved over ranks,
one RCCL reduce of the radiance buffer to rank 0 (torch.distributed, backend
"nccl" = RCCL over xGMI).  The reference is single device (src/kernelgl.cpp:76);
this module is new, mandated by BASELINE.json's north_star.

The path shards naturally: every pixel-sample is independent, so
the only exchange is the final gather of disjoint tiles, expressed as a sum of
zero-initialised full-frame buffers (each pixel is non-zero on exactly one rank,
so the sum is exact and bit-identical to the single-GPU frame).

torch is plumbing here: process group, device tensors wrapped around the
library's own HIP buffers (zero copy, via __cuda_array_interface__).
"""
import os

import numpy as np

TILE = 8  # 8×8-pixel tiles: fine-grained interleave balances sky vs glass across ranks


def tile_owner_map(width, height, world, tile_w=TILE, tile_h=TILE):
    """(h, w) int array: which rank owns each pixel — the host restatement of the
    kernel's slot→pixel mapping (rt_set_shard in include/rt_amd.h): tiles are numbered
    row-major and tile t belongs to rank t % world."""
    tiles_x = (width + tile_w - 1) // tile_w
    ty, tx = np.meshgrid(np.arange(height) // tile_h, np.arange(width) // tile_w, indexing="ij")
    return (ty * tiles_x + tx) % world


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT).  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def reduce_frame(buf, dst=0):
    """Sum the ranks' full-frame buffers into rank `dst` (torch tensor, in place)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM)
    return buf


class ShardedRenderer:
    """One rank's share of a tile-sharded frame.

    ``tracer`` is this rank's RayTracer (its own GPU).  ``render(camera, spp)`` traces
    this rank's tiles with the fused kernel, then reduces the linear accumulator
    (RGB sum + sample count per pixel) to rank 0, where ``resolve`` turns it into the
    gamma-space image.  Everything is enqueued on torch's current stream.
    """

    def __init__(self, tracer, rank, world, tile=TILE):
        import torch
        self.torch = torch
        self.tracer, self.rank, self.world = tracer, rank, world
        tracer.setShard(rank, world, tile, tile)
        tracer.setStream(torch.cuda.current_stream().cuda_stream)
        self.accum = torch.as_tensor(tracer.deviceAccum(), device="cuda")

    def render(self, camera, spp, first_sample=0):
        t = self.tracer
        t.clear()
        t.renderSamples(camera, first_sample, spp)
        reduce_frame(self.accum, dst=0)
        if self.rank == 0:
            t.resolve()

    def image(self):
        """Gamma-space frame on rank 0 (h, w, 4)."""
        return self.tracer.transferImage()
