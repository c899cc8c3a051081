#!/usr/bin/env python3
"""tools/wstat.py — lane census of pt_samples_w (GPU box; needs the diagnostic build
`tools/build_variant.sh wstat -DPT_WSTAT=1`): how many lanes sit in which phase per outer iteration, and inside the
mesh walk how many lanes test a node / wait per step.  Prints the raw counters ([wstat] lines on stderr)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencl_raytracing_amd as rt
rt.load_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opencl-raytracing_amd", "variants", "wstat.so"))
w, h, spp = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1920, 1080, 64)))
wl = rt.workloads.get("c5", width=w, height=h)
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
t.debugCounters()            # clears the census
t.clear(); t.renderSamples(wl.camera, 0, spp); t.sync()
t.debugCounters()            # prints it
t.close()
