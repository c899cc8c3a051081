#!/usr/bin/env python3
"""tools/fuzz_more.py <first seed> <last seed> [odd] — the random scenes of tests/test_gpu_fuzz.py at 8 / 64 / 256 / 512 samples per
pixel, all three acceleration modes, three repeats each (the order of pt_prefix's appends differs from run to run), fused
frames bit for bit against the oracle's per-sample values summed in kernel order.  Round 2: seeds 200-399, no mismatch."""
import sys, os, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/oracle')
import cases
rt=cases.rt
from oracle import Oracle
from test_gpu_fuzz import random_scene
from test_gpu_parity import fused_sum_in_kernel_order
o=Oracle(); tab=rt.workloads.make_random_table(cases.SEED)
bad=0; t0=time.time()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    s,cam=random_scene(seed)
    W,H=48,27
    spp=([8,64,256][seed%3] if seed%9 else 512) if len(sys.argv) < 4 else [5,24,47,100,200,300,511][seed%7]   # any 4th argument: counts off the lane-group grid
    t=rt.RayTracer(W,H,scene=s,seed=cases.SEED)
    t.setOption(t.OPT_WAVE_FILL, seed%2)   # even seeds: as many pixels per wave as the LDS share holds (the frames are tiny)
    yy,xx,sm=np.meshgrid(np.arange(H),np.arange(W),np.arange(spp),indexing='ij')
    per,_=o.samples(s,cam,tab,W,H,xx.ravel(),yy.ravel(),sm.ravel())
    exp=(fused_sum_in_kernel_order(per.reshape(H*W,spp,3),spp).reshape(H,W,3)/np.float32(spp)).astype(np.float32)
    ok=True
    for accel in (1,0,2):
        t.setOption(t.OPT_ACCEL,accel)
        for rep in range(3):
            t.clear(); t.renderSamples(cam,0,spp); t.sync()
            f=t.readLinear()[...,:3]
            if not np.array_equal(f.view(np.uint32),exp.view(np.uint32)):
                ok=False; print('MISMATCH seed',seed,'spp',spp,'accel',accel,'rep',rep,'pixels',int((f.view(np.uint32)!=exp.view(np.uint32)).any(-1).sum()),flush=True)
    bad+=not ok
    t.close()
    print('seed',seed,'spp',spp,'spheres',len(s.spheres),'meshes',[int(x) for x in s.meshes['face_count']],'ok' if ok else 'BAD','%.0fs'%(time.time()-t0),flush=True)
print('scenes with mismatches:',bad)
