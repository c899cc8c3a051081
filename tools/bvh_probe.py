#!/usr/bin/env python3
"""C4 scaling probe: kernel time vs sphere count with the BVH on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencl_raytracing_amd as rt
for n in (1000, 10000, 100000):
    wl = rt.workloads.get("c4", width=480, height=270, n_spheres=n)
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
    for accel in (2, 0):
        if accel == 0 and n > 10000:
            continue
        t.setOption(t.OPT_ACCEL, accel)
        for spp in (1, 16):
            t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
            t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
            t.enableCounters(True); t.resetCounters(); t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
            cn = t.counters(); dbg = t.debugCounters(); t.enableCounters(False)
            t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
            print("n=%6d accel=%d spp=%2d  %.3f ms  bounces/sample %.2f  nodes/ray %.1f  tests/ray %.1f" % (n, accel, spp, t.lastKernelMs(), cn.bounces / max(cn.samples, 1), dbg[0] / max(cn.bounces, 1), dbg[1] / max(cn.bounces, 1)), flush=True)
    t.close()
