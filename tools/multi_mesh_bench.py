#!/usr/bin/env python3
"""tools/multi_mesh_bench.py [--lib=PATH] [n_meshes ...] — fused 64-spp frames (960x540) of a scene of N uv-sphere meshes
(MESH_GRID=segments x rings, default 120x80 = 18 960 faces each; every one with a BVH: pt_samples_w<true>) on a diffuse plane under a light sphere: how the frame time
grows with the number of meshes.  GPU box."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opencl_raytracing_amd as rt
args = sys.argv[1:]
tag = "default"
if args and args[0].startswith("--lib="):
    rt.load_library(args[0][6:]); tag = os.path.basename(args.pop(0)[6:])
A = rt._abi
W, H, SPP = 960, 540, 64
SEG, RINGS = (int(v) for v in os.environ.get("MESH_GRID", "120x80").split("x"))   # 120x80: 18 960 faces, edges of 0.07 (the cap margin applies); 40x27: 2 080 faces, edges of 0.2 (it does not: wide-cone nodes are never culled)
for n in [int(a) for a in args] or [1, 2, 4, 8, 16]:
    s = rt.SceneCreator()
    s.addMaterial(A.T_DIFFUSE, (0.8, 0.8, 0.8), 1)    # 0 floor
    s.addMaterial(A.T_LIGHT, (1, 1, 1), 0)            # 1
    for k, col in enumerate(((0.9, 0.3, 0.2), (0.2, 0.8, 0.3), (0.3, 0.4, 0.9))):
        s.addMaterial(A.T_DIFFUSE, col, 1)            # 2..4
    s.addMaterial(A.T_DIELECTRIC, (1, 1, 1), 1.4)     # 5
    s.addMaterial(A.T_REFLECTIVE, (1, 1, 1), 0.9)     # 6
    side = int(np.ceil(np.sqrt(n)))
    for k in range(n):
        cx, cz = (k % side - (side - 1) / 2) * 3.2, (k // side - (side - 1) / 2) * 3.2 + 6.0
        pos, uv, idx = rt.workloads.uv_sphere(SEG, RINGS, radius=1.3, centre=(cx, 3.6, cz))
        s.addMesh(pos, uv, idx)
        s.addModel(1, 2 + k % 5)
    s.addSphere((1, -200, 0), 100, 1)
    s.addPlane((0, 5, 0), (0, 1, 0), 0)
    cam = rt.Camera(60, W / H, (0, 1.0, -6.0), 0.0, 12.0).transferData()
    t = rt.RayTracer(W, H, scene=s)
    t.setArith(2)
    t.clear(); t.renderSamples(cam, 0, SPP); t.sync()
    digest = hashlib.sha1(t.readLinear().tobytes()).hexdigest()[:12]
    ms = []
    for _ in range(4):
        t.clear(); t.renderSamples(cam, 0, SPP); t.sync(); ms.append(t.lastKernelMs())
    print("%-12s %2d meshes x %d faces  %dx%d x %d spp  %8.3f ms  image %s" % (tag, n, len(idx) // 3, W, H, SPP, min(ms), digest), flush=True)
    assert t.walkOverflow() == 0
    t.close()
