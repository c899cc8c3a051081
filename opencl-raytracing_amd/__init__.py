"""MI355X-native path tracer — host-side Python mirror of the reference's
RayTracer / SceneCreator / Camera API over the C ABI of librt_amd.so
(include/rt_amd.h).  The directory name carries a hyphen; import it with
``import opencl_raytracing_amd`` (repo-root shim) or
``importlib.import_module("opencl-raytracing_amd")``.
"""
from . import _abi, workloads  # noqa: F401
from .camera import Camera  # noqa: F401
from .raytracer import (ARITH_IEEE, ARITH_NAMES, ARITH_ROCM_OCL, ARITH_ROCM_OCL_NOCONTRACT, LIB_PATH, RayTracer,  # noqa: F401
                        RtError, check_accel, load_library, make_random_table)
from .scene import SceneCreator, SceneError  # noqa: F401

__all__ = ["Camera", "RayTracer", "RtError", "SceneCreator", "SceneError", "workloads", "load_library",
           "make_random_table", "LIB_PATH"]
