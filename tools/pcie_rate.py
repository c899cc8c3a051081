#!/usr/bin/env python3
"""tools/pcie_rate.py — the boundary's host-buffer legs (GPU box): rt_transfer_image (device image → host, the
reference's clEnqueueReadImage-equivalent hand-over) and rt_set_scene, timed beside the trace call of C2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opencl_raytracing_amd as rt

wl = rt.workloads.get("c2")
t0 = time.perf_counter()
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
t.sync()
print("context + rt_set_scene (C2): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for _ in range(3):
    t.clear(); t.renderSamples(wl.camera, 0, wl.spp); t.resolve(); t.sync()
best_frame, best_xfer, best_both = 1e9, 1e9, 1e9
for _ in range(10):
    a = time.perf_counter()
    t.clear(); t.renderSamples(wl.camera, 0, wl.spp); t.resolve(); t.sync()
    b = time.perf_counter()
    img = t.transferImage()
    c = time.perf_counter()
    best_frame, best_xfer, best_both = min(best_frame, b - a), min(best_xfer, c - b), min(best_both, c - a)
n = wl.width * wl.height * wl.spp
print("frame (clear + trace + resolve, host clock) %.3f ms; rt_transfer_image of %.1f MB %.3f ms = %.1f GB/s; both %.3f ms"
      % (best_frame * 1e3, img.nbytes / 1e6, best_xfer * 1e3, img.nbytes / best_xfer / 1e9, best_both * 1e3))
print("pixel-samples/s: resident %.1f G, with the image handed to the host %.1f G" % (n / best_frame / 1e9, n / best_both / 1e9))
t.close()
