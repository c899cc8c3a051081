#!/usr/bin/env python3
"""C5: interleaved walk slices (pt_samples_w) against walks in place (pt_samples_q): time and bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opencl_raytracing_amd as rt
if len(sys.argv) > 1: rt.load_library(sys.argv[1])
wl = rt.workloads.get("c5")
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
out = []
for slices in (1, 0, 1, 0):
    t.setOption(t.OPT_WALK_SLICES, slices)
    ms = []
    for _ in range(2):
        t.clear(); t.renderSamples(wl.camera, 0, 16); t.sync(); ms.append(t.lastKernelMs())
    out.append(t.readLinear())
    print("c5 4K 16 spp walk_slices=%d: %.2f ms" % (slices, min(ms)), flush=True)
print("bit-identical:", np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32)))
t.close()
