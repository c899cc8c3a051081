#!/usr/bin/env python3
"""Diagnostic: s_memtime shares of pt_samples_q's sections (needs a -DPT_STAMPS=1 build):
   tools/build_variant.sh stamps -DPT_STAMPS=1 && python tools/stamps.py opencl-raytracing_amd/variants/stamps.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencl_raytracing_amd as rt
rt.load_library(sys.argv[1])
wl = rt.workloads.get(sys.argv[2] if len(sys.argv) > 2 else "c2")
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
t.setArith(int(os.environ.get("RT_ARITH", "2")))   # the policy bench.py times
t.clear(); t.renderSamples(wl.camera, 0, wl.spp); t.sync()
t.debugCounters()
t.close()
