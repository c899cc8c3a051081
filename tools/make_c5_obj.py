#!/usr/bin/env python3
"""tools/make_c5_obj.py — (re)write assets/models/c5_sphere.obj, BASELINE.json's "50k-triangle OBJ mesh" of
configuration 5 (generator: opencl-raytracing_amd/workloads.py write_c5_obj)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencl_raytracing_amd as rt  # noqa: E402

print(rt.workloads.write_c5_obj())
