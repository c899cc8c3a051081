#!/usr/bin/env python3
"""Lane utilisation of pt_samples_q's iterations (build with -DPT_QSTAT): tools/qstat.py lib.so [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencl_raytracing_amd as rt
rt.load_library(sys.argv[1])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = rt.workloads.get("c2")
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
t.setOption(t.OPT_SAMPLE_QUEUE, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
t.enableCounters(True); t.resetCounters(); t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
d = t.debugCounters(); cn = t.counters()
print("spp %d: lane-iterations used %d of %d offered = %.1f %%; per sample %.2f iterations" % (spp, d[0], d[1], 100.0 * d[0] / max(d[1], 1), d[0] / cn.samples))
t.close()
