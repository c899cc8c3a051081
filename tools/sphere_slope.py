#!/usr/bin/env python3
"""Cost of one more sphere test on C2: add spheres no ray can meet and watch the kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opencl_raytracing_amd as rt
for extra in (0, 8, 24, 56):
    wl = rt.workloads.get("c2")
    for i in range(extra):
        wl.scene.addSphere((5000.0 + 3 * i, -4000.0, 7000.0), 0.25, 3)
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
    t.setOption(t.OPT_ACCEL, 0)
    ms = []
    for _ in range(4):
        t.clear(); t.renderSamples(wl.camera, 0, 64); t.sync(); ms.append(t.lastKernelMs())
    print("c2 + %2d unreachable spheres: %.3f ms" % (extra, min(ms)), flush=True)
    t.close()
