#!/usr/bin/env python3
"""Kernel-time survey over workloads (GPU box): tools/quick_bench.py c2:64 c3:256 c4:2 c5:2"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opencl_raytracing_amd as rt

args = sys.argv[1:]
if args and args[0].startswith("--lib="):
    rt.load_library(args.pop(0)[6:])
queue = None
if args and args[0].startswith("--queue="):
    queue = int(args.pop(0)[8:])
for spec in args:
    parts = spec.split(":")
    name, spp = parts[0], int(parts[1])
    kw = {}
    if len(parts) > 2:
        kw = dict(width=int(parts[2]), height=int(parts[3]))
    wl = rt.workloads.get(name, **kw)
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
    if queue is not None:
        t.setOption(t.OPT_SAMPLE_QUEUE, queue)
    t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
    ms = []
    for _ in range(3):
        t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync(); ms.append(t.lastKernelMs())
    n = wl.width * wl.height * spp
    print("%-4s %dx%d spp %-4d kernel %.3f ms  %.1f Msamples/s" % (name, wl.width, wl.height, spp, min(ms), n / min(ms) / 1e3), flush=True)
    t.close()
