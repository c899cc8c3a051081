#!/usr/bin/env python3
"""C5 probe: kernel time with the mesh BVH on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencl_raytracing_amd as rt
if len(sys.argv) > 1: rt.load_library(sys.argv[1])
wl = rt.workloads.get("c5", width=480, height=270)
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
for accel, spps in ((1, (1, 16)),):
    t.setOption(t.OPT_ACCEL, accel)
    for spp in spps:
        t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
        t.enableCounters(True); t.resetCounters(); t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
        cn = t.counters(); dbg = t.debugCounters(); t.enableCounters(False)
        t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
        print("c5 480x270 accel=%d spp=%2d  %.3f ms  bounces/sample %.2f  tri tests/bounce (reference) %.0f  bvh nodes/bounce %.1f  faces tested/bounce %.1f" % (accel, spp, t.lastKernelMs(), cn.bounces / max(cn.samples, 1), cn.t_tri / max(cn.bounces, 1), dbg[0] / max(cn.bounces, 1), dbg[1] / max(cn.bounces, 1)), flush=True)
t.close()
