import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import opencl_raytracing_amd as rt
from oracle import Oracle
wl = rt.workloads.get("c2", width=8192, height=4608)
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=rt.workloads.SEED)
t.clear(); t.renderSamples(wl.camera, 0, 8); t.sync()
print("8192x4608 x 8 spp kernel %.2f ms" % t.lastKernelMs())
lin = t.readLinear()
print("lit fraction", float((lin[..., :3].sum(-1) > 0).mean()))
rng = np.random.RandomState(3)
xs, ys, ss = rng.randint(0, wl.width, 300), rng.randint(0, wl.height, 300), rng.randint(0, 8, 300)
got = t.traceSamples(wl.camera, xs, ys, ss)
exp, _ = Oracle().samples(wl.scene, wl.camera, t.getRandomTable(), wl.width, wl.height, xs, ys, ss)
print("probes bit-exact:", int((got.view(np.uint32) == exp.view(np.uint32)).all(axis=1).sum()), "of 300")
# the frame's pixel = mean of its 8 samples: check a few pixels against per-sample probes
for k in range(5):
    x, y = int(xs[k]), int(ys[k])
    s8, _ = Oracle().samples(wl.scene, wl.camera, t.getRandomTable(), wl.width, wl.height, [x]*8, [y]*8, list(range(8)))
    print(x, y, np.abs(lin[y, x, :3] - s8.mean(0)).max())
t.close()
