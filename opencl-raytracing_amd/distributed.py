"""Multi-GPU rendering: one process per GPU, frame tiles interleaved over ranks,
one exchange of the radiance (accumulator) buffer to rank 0 over xGMI
(torch.distributed, backend "nccl" = RCCL).  The reference is single device
(src/kernelgl.cpp:76); this module is new, mandated by BASELINE.json's north_star.

The path shards naturally: every pixel-sample is independent, pixels
keep their whole-frame coordinates, so each pixel is bit-identical to the
single-GPU render and the only exchange is the final collection of disjoint tiles:

  exchange="gather" (default)  every rank packs the accumulator pixels it owns
      (1/world of the frame, slot order, rt_pack_accum) and rank 0 gathers and
      scatters them into its frame (rt_unpack_accum): 33.2 MB·(1−1/N) arrive at
      rank 0 over N−1 xGMI links in parallel.
  exchange="reduce"            sum of the zero-initialised full-frame buffers
      (each pixel is non-zero on exactly one rank, so the sum is exact): the
      collective BASELINE.json names; moves the whole frame per rank.

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


def shard_slots(width, height, world, tile_w=TILE, tile_h=TILE):
    """Pixel slots every rank packs (rt_shard_slots): whole tiles, equal for all ranks."""
    tiles = ((width + tile_w - 1) // tile_w) * ((height + tile_h - 1) // tile_h)
    return (tiles + world - 1) // world * tile_w * tile_h


def slot_pixels(width, height, rank, world, tile_w=TILE, tile_h=TILE):
    """Host restatement of slot_to_pixel (csrc/rt_amd.hip): for rank's slots 0..n-1 →
    (valid mask, y, x).  Slot s = k·tile_area + j lives in tile t = rank + k·world."""
    n = shard_slots(width, height, world, tile_w, tile_h)
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_total = tiles_x * ((height + tile_h - 1) // tile_h)
    s = np.arange(n)
    k, j = s // (tile_w * tile_h), s % (tile_w * tile_h)
    t = rank + k * world
    x = (t % tiles_x) * tile_w + j % tile_w
    y = (t // tiles_x) * tile_h + j // tile_w
    valid = (t < tiles_total) & (x < width) & (y < height)
    return valid, np.where(valid, y, 0), np.where(valid, x, 0)


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
            # RT_DIST_BACKEND=gloo rehearses the multi-rank path on fewer GPUs than ranks (gloo moves CUDA
            # tensors through the host for reduce/all_reduce; RCCL refuses two ranks on one device)
            backend = os.environ.get("RT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


class GpuShard:
    """Adapter between ShardedRenderer and a RayTracer on this rank's GPU."""

    def __init__(self, tracer, rank, world, tile=TILE, chunk=None):
        """``chunk``: samples per fused trace call (None = all in one call).  bench.py's weak-scaling run gives
        every pixel 64·N samples in ONE call per rank (a launch over an N-th of the pixels is mostly tail: N calls of 64
        cost a rank 3.4 ms at N = 8, one call of 512 costs 1.4 ms — tools/shard_chunk_probe.py); above 512 samples per
        pixel (the capacity of a wave's sample queue) it falls back to calls of the base sample count."""
        import torch
        self.torch = torch
        self.tracer, self.rank, self.world, self.chunk = tracer, rank, world, chunk
        tracer.setShard(rank, world, tile, tile)
        # One dedicated (non-default) HIP stream carries the library's kernels AND is torch's
        # current stream while collectives are issued, so RCCL orders itself after the trace
        # kernels and the unpack/resolve kernels after RCCL — no host synchronisation.
        self.stream = torch.cuda.Stream()
        tracer.setStream(self.stream.cuda_stream)
        self.accum = torch.as_tensor(tracer.deviceAccum(), device="cuda")          # (h, w, 4), zero copy
        self.slots = tracer.shardSlots(world)

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def new_packed(self):
        import torch
        return torch.empty((self.slots, 4), dtype=torch.float32, device="cuda")

    def trace(self, camera, first_sample, spp):
        self.tracer.clear()
        step = self.chunk or spp
        for s in range(0, spp, step):
            self.tracer.renderSamples(camera, first_sample + s, min(step, spp - s))

    def pack(self, out):
        self.tracer.packAccum(out.data_ptr(), out.numel() * 4)

    def unpack(self, packed, src_rank):
        self.tracer.unpackAccum(packed.data_ptr(), packed.numel() * 4, src_rank, self.world)

    def resolve(self):
        self.tracer.resolve()

    def image(self):
        return self.tracer.transferImage()


class ShardedRenderer:
    """One rank's share of a tile-sharded frame.  ``shard`` is a GpuShard (or any object
    with its trace / pack / unpack / resolve / accum / new_packed surface — the CPU tests
    drive this class over gloo with an oracle-backed stand-in).  Everything is enqueued
    on torch's current stream; nothing here synchronises the host."""

    def __init__(self, shard, rank, world, exchange="gather"):
        assert exchange in ("gather", "reduce")
        self.shard, self.rank, self.world, self.exchange = shard, rank, world, exchange
        if world > 1 and exchange == "gather":
            import contextlib
            with (shard.stream_context() if hasattr(shard, "stream_context") else contextlib.nullcontext()):
                self.send = shard.new_packed()
                self.recv = [shard.new_packed() for _ in range(world)] if rank == 0 else None

    def render(self, camera, spp, first_sample=0):
        import contextlib
        import torch.distributed as dist
        sh = self.shard
        ctx = sh.stream_context() if hasattr(sh, "stream_context") else contextlib.nullcontext()
        with ctx:
            sh.trace(camera, first_sample, spp)
            if self.world > 1:
                if self.exchange == "reduce":
                    reduce_frame(sh.accum, dst=0)
                else:
                    sh.pack(self.send)
                    try:
                        dist.gather(self.send, self.recv, dst=0)
                    except (RuntimeError, ValueError, NotImplementedError) as e:
                        # a backend without gather for these tensors raises on every rank alike:
                        # all ranks fall back to the full-frame reduce, this frame included
                        import warnings
                        warnings.warn("gather exchange unavailable (%s); using the full-frame reduce" % e)
                        self.exchange = "reduce"
                        reduce_frame(sh.accum, dst=0)
                    else:
                        if self.rank == 0:
                            for r in range(1, self.world):   # rank 0's own tiles are already in place
                                sh.unpack(self.recv[r], r)
            if self.rank == 0:
                sh.resolve()

    def image(self):
        """Gamma-space frame on rank 0 (h, w, 4)."""
        return self.shard.image()
