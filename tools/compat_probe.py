import os, sys, time
sys.path.insert(0, "/root/repo")
import opencl_raytracing_amd as rt
wl = rt.workloads.get("c2")
t = rt.RayTracer(wl.width, wl.height, scene=wl.scene)
t.render(wl.camera); t.renderAgain(wl.camera)
ks = []
t0 = time.perf_counter()
t.render(wl.camera); k0 = t.lastKernelMs()
for _ in range(63):
    t.renderAgain(wl.camera); ks.append(t.lastKernelMs())
dt = time.perf_counter() - t0
print("trace kernel %.3f ms; retrace kernel mean %.3f ms; wall per call %.3f ms" % (k0, sum(ks)/len(ks), dt*1e3/64))
