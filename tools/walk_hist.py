#!/usr/bin/env python3
"""tools/walk_hist.py — where do the mesh-walk nodes of a frame go?  (GPU box; needs the diagnostic build
`tools/build_variant.sh probenodes -DPT_PROBE_NODES=1`, whose probe returns (nodes tested, walks, bounces) per sample.)
Prints the distribution of nodes per sample over a pixel grid of C5 and saves the per-pixel mean as
gpurun_out/walk_nodes.npy."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opencl_raytracing_amd as rt
rt.load_library(os.path.join(ROOT, "opencl-raytracing_amd", "variants", "probenodes.so"))
w, h, step, ns = 1920, 1080, 4, 8
wl = rt.workloads.get(sys.argv[1] if len(sys.argv) > 1 else "c5", width=w, height=h)
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
ys, xs, ss = np.meshgrid(np.arange(0, h, step), np.arange(0, w, step), np.arange(ns), indexing="ij")
shape = xs.shape
out = t.traceSamples(wl.camera, xs.ravel().astype(np.uint32), ys.ravel().astype(np.uint32), ss.ravel().astype(np.uint32))
t.close()
nodes, walks, bounces = (out[:, k].reshape(shape) for k in range(3))
tot = nodes.sum()
print("samples %d  nodes/sample %.1f  walks/sample %.2f  bounces/sample %.2f  nodes/walk %.1f" %
      (nodes.size, nodes.mean(), walks.mean(), bounces.mean(), tot / max(walks.sum(), 1)))
flat = np.sort(nodes.ravel())
cum = np.cumsum(flat) / tot
for q in (0.5, 0.75, 0.9, 0.95, 0.99, 0.999):
    i = int(q * flat.size)
    print("  %5.1f %% of the samples use <= %6.0f nodes each and %5.1f %% of all nodes" % (100 * q, flat[i], 100 * cum[i]))
for lim in (1, 2, 8, 32, 128, 512, 2048):
    m = nodes <= lim * np.maximum(walks, 1)
    print("  samples with <= %4d nodes per walk: %5.1f %% of samples, %5.1f %% of nodes" % (lim, 100 * m.mean(), 100 * nodes[m].sum() / tot))
pix = nodes.mean(axis=2)
np.save(os.path.join(ROOT, "gpurun_out", "walk_nodes.npy"), pix.astype(np.float32))
# coarse map: 12 x 24 cells, mean nodes per sample
gh, gw = pix.shape[0] // 12, pix.shape[1] // 24
print("coarse map (mean nodes per sample):")
for r in range(12):
    print(" ".join("%5.0f" % pix[r * gh:(r + 1) * gh, c * gw:(c + 1) * gw].mean() for c in range(24)))
