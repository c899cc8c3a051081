#!/usr/bin/env python3
"""Kernel-time survey over workloads and library variants (GPU box):
    tools/quick_bench.py [--lib=PATH] [--queue=0|1] [--arith=0|1|2] [--tree=0|1|2] [--fill=0|1] c2:64 c3:256 c4:2 c5:2[:W:H]
Prints the best of 5 HIP-event times of the fused trace call and a hash of the accumulator (every
variant of the library must print the same hash for the same spec: results never depend on tuning)."""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opencl_raytracing_amd as rt

args = sys.argv[1:]
lib = "default"
if args and args[0].startswith("--lib="):
    lib = args.pop(0)[6:]
    rt.load_library(lib)
queue = None
if args and args[0].startswith("--queue="):
    queue = int(args.pop(0)[8:])
arith, tree, fill = 2, None, None
while args and args[0].startswith(("--arith=", "--tree=", "--fill=")):
    k, v = args.pop(0).split("=")
    if k == "--arith":
        arith = int(v)
    elif k == "--fill":
        fill = int(v)
    else:
        tree = int(v)
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
    t.setArith(arith)
    if tree is not None:
        t.setOption(t.OPT_PREFIX_TREE, tree)
    if fill is not None:
        t.setOption(t.OPT_WAVE_FILL, fill)
    t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
    digest = hashlib.sha1(t.readLinear().tobytes()).hexdigest()[:12]
    ms = []
    for _ in range(5):
        t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync(); ms.append(t.lastKernelMs())
    n = wl.width * wl.height * spp
    print("%-28s %-4s %dx%d spp %-4d kernel %8.3f ms  %9.1f Msamples/s  image %s" %
          (os.path.basename(lib), name, wl.width, wl.height, spp, min(ms), n / min(ms) / 1e3, digest), flush=True)
    t.close()
