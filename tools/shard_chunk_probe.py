#!/usr/bin/env python3
"""tools/shard_chunk_probe.py — GPU box: what one rank of an N-rank WEAK-scaling run costs, on one GPU.
Rank 0 of N owns every N-th 8x8 tile of C2's frame and traces 64·N samples per pixel — the pixel-samples of one N = 1
frame.  Issued as N calls of 64 (round 2's choice: the kernels of the N = 1 line) or as fewer, larger calls?"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opencl_raytracing_amd as rt

wl = rt.workloads.get(sys.argv[1] if len(sys.argv) > 1 else "c2")
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
t.setArith(2)
for world in (1, 2, 4, 8):
    t.setShard(0, world, 8, 8)
    spp = wl.spp * world
    for chunk in sorted({wl.spp, spp, min(spp, 256)}):
        if spp % chunk:
            continue
        def frame():
            t.clear()
            for s in range(0, spp, chunk):
                t.renderSamples(wl.camera, s, chunk)
            t.resolve()
        for _ in range(3):
            frame()
        t.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            frame()
        t.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print("world %d  spp %4d in calls of %4d: %8.3f ms per step  (%.1f %% of ideal weak scaling)" %
              (world, spp, chunk, ms, 0.0 if world == 1 and chunk != wl.spp else 100.0 * BASE / ms if 'BASE' in dir() else 100.0), flush=True)
        if world == 1 and chunk == wl.spp:
            BASE = ms
t.close()
