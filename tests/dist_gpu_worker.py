"""Worker of tests/test_gpu_bench.py::test_sharded_frame_equals_unsharded_frame: one of three gloo
ranks sharing the one GPU.  The product's ShardedRenderer + GpuShard render a tile-sharded frame,
rank 0 collects it (both exchange forms) and compares with its own unsharded render, bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from importlib import import_module  # noqa: E402

import opencl_raytracing_amd as rt  # noqa: E402

d = import_module("opencl-raytracing_amd.distributed")
rank, world, _ = d.init_process_group("gloo")
torch.cuda.set_device(0)
wl = rt.workloads.get("all_kinds", width=301, height=171)   # ragged in both directions
SPP = 24
for exchange in ("gather", "reduce"):
    tracer = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=rt.workloads.SEED)
    r = d.ShardedRenderer(d.GpuShard(tracer, rank, world), rank, world, exchange=exchange)
    for _ in range(2):                                       # twice: buffers are reused between frames
        r.render(wl.camera, SPP)
    tracer.sync()
    if rank == 0:
        got_lin, got_img = tracer.readLinear(), r.image()
        ref = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=rt.workloads.SEED)
        ref.clear()
        ref.renderSamples(wl.camera, 0, SPP)
        ref.resolve()
        exp_lin, exp_img = ref.readLinear(), ref.transferImage()
        assert np.array_equal(got_lin.view(np.uint32), exp_lin.view(np.uint32)), exchange
        assert np.array_equal(got_img.view(np.uint32), exp_img.view(np.uint32)), exchange
        assert (exp_lin[..., :3].sum(-1) > 0).mean() > 0.3
        ref.close()
        print("SHARDED_OK", exchange, r.exchange, flush=True)
    dist.barrier()
    tracer.close()
dist.destroy_process_group()
