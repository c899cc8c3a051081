#!/usr/bin/env python3
"""Progressive (compat) path timing: render + (spp-1) x renderAgain, one launch + host sync per sample,
exactly the reference's interactive loop (src/raytracer.cpp:127-165)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencl_raytracing_amd as rt
for name, w, h, spp in (("c2", 1920, 1080, 64), ("all_kinds", 1200, 800, 64)):
    wl = rt.workloads.get(name, width=w, height=h)
    t = rt.RayTracer(w, h, scene=wl.scene)
    t.render(wl.camera); t.renderAgain(wl.camera)
    t0 = time.perf_counter()
    t.render(wl.camera)
    for _ in range(spp - 1):
        t.renderAgain(wl.camera)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter(); img = t.renderFrame(wl.camera, spp); dt_f = time.perf_counter() - t1
    print("%s %dx%d %d spp: progressive %.2f ms (%.3f ms per sample-frame, %.0f Msamples/s); fused incl. read-back %.2f ms" %
          (name, w, h, spp, dt * 1e3, dt * 1e3 / spp, w * h * spp / dt / 1e6, dt_f * 1e3), flush=True)
    t.close()
