#!/usr/bin/env python3
"""tools/fuzz_ref950.py <first seed> <last seed> — the random scenes of tests/test_gpu_fuzz.py (policies 1 and 2 alternating)
against the reference kernel file as ROCm's OpenCL tool chain builds it for gfx950 (oracle/_ref_gfx950/*.hsaco): samples 0-2
of every pixel bit for bit with the acceleration structures on and off, the fused frame within 1e-4.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import test_gpu_fuzz as F
bad, t0 = 0, time.time()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        F.test_random_scene_bit_exact_against_the_rocm_opencl_build(seed)
        print("seed", seed, "policy", 1 + seed % 2, "ok", "%.0fs" % (time.time() - t0), flush=True)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "MISMATCH", e, flush=True)
print("scenes with mismatches:", bad)
