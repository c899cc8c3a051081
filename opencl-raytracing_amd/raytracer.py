"""RayTracer — host-side mirror of the reference's class (include/raytracer.h:17-47,
src/raytracer.cpp:24-174) over the C ABI of librt_amd.so (include/rt_amd.h).

Same surface as the reference: ``RayTracer(w, h, kernel_path)``, ``render(camera)``,
``renderAgain(camera)``, ``transferImage()``, plus the two methods the reference
declares but never defines (``resize``, ``setTime``) and the native fused path
(``renderSamples``).  The HIP library is the only compute path: if it cannot be
loaded, or no gfx950 device is present, construction raises — there is no CPU
fallback here.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from .scene import SceneCreator

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_amd.so")

# every symbol include/rt_amd.h declares
SYMBOLS = (
    "rt_abi_version", "rt_last_error", "rt_create", "rt_destroy", "rt_resize", "rt_set_stream", "rt_set_scene",
    "rt_set_textures", "rt_set_seed", "rt_set_random_table", "rt_get_random_table", "rt_make_random_table",
    "rt_set_shard", "rt_render", "rt_render_again", "rt_sample_counter", "rt_clear", "rt_render_spp", "rt_resolve",
    "rt_sync", "rt_trace_samples", "rt_read_image", "rt_read_linear", "rt_device_image", "rt_device_accum",
    "rt_enable_counters", "rt_reset_counters", "rt_get_counters", "rt_counters_bytes", "rt_last_kernel_ms",
    "rt_kernel_ms_history", "rt_stage_ms_history", "rt_debug_hit", "rt_debug_material", "rt_debug_div3", "rt_device_info", "rt_set_option", "rt_shard_slots", "rt_pack_accum", "rt_unpack_accum",
    "rt_get_debug_counters", "rt_debug_check_accel", "rt_walk_overflow", "rt_debug_builtin",
)

# rt_set_option: options and the arithmetic policies of RT_OPT_ARITH (include/rt_amd.h)
OPT_PREFIX_SHARING, OPT_MAX_THREADS_PER_LAUNCH, OPT_SAMPLE_QUEUE, OPT_ACCEL, OPT_WALK_SLICES, OPT_ARITH, OPT_PREFIX_TREE, OPT_WAVE_FILL = 1, 2, 3, 4, 5, 6, 7, 8
ARITH_IEEE, ARITH_ROCM_OCL_NOCONTRACT, ARITH_ROCM_OCL = 0, 1, 2
ARITH_NAMES = {"ieee": ARITH_IEEE, "rocm-opencl-nocontract": ARITH_ROCM_OCL_NOCONTRACT, "rocm-opencl": ARITH_ROCM_OCL}


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librt_amd: %s (code %d)" % (msg, code))
        self.code = code


_lib = None


def load_library(path=LIB_PATH):
    """dlopen librt_amd.so and type its entry points.  Raises if it is missing:
    build it with ``python -c 'import __graft_entry__ as g; g.build()'``."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(path):
        raise OSError("%s not found — the HIP library is not built (run __graft_entry__.build()); "
                      "there is no CPU fallback" % path)
    # PyTorch's ROCm wheel bundles its own libamdhip64.so.7.  A process must hold ONE HIP runtime:
    # if librt_amd.so pulled in /opt/rocm's copy first, a later `import torch` would find "No HIP
    # GPUs".  Importing torch first (when it is installed) makes both bind to the same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for s in SYMBOLS:
        getattr(lib, s)  # AttributeError if the library does not export what the header declares
    vp, u32, u64, sz, fp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t, C.POINTER(C.c_float)
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_last_error.argtypes = [vp]
    lib.rt_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_destroy.restype = None
    lib.rt_resize.argtypes = [vp, C.c_int, C.c_int]
    lib.rt_set_stream.argtypes = [vp, vp]
    lib.rt_set_scene.argtypes = [vp, C.POINTER(_abi.SceneDesc)]
    lib.rt_set_textures.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    lib.rt_set_seed.argtypes = [vp, u64]
    lib.rt_set_random_table.argtypes = [vp, vp, sz]
    lib.rt_get_random_table.argtypes = [vp, vp, sz]
    lib.rt_make_random_table.argtypes = [u64, vp, sz]
    lib.rt_set_shard.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rt_render.argtypes = [vp, vp]
    lib.rt_render_again.argtypes = [vp, vp]
    lib.rt_sample_counter.argtypes = [vp, C.POINTER(u32)]
    lib.rt_clear.argtypes = [vp]
    lib.rt_render_spp.argtypes = [vp, vp, u32, u32]
    lib.rt_resolve.argtypes = [vp]
    lib.rt_sync.argtypes = [vp]
    lib.rt_trace_samples.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.rt_read_image.argtypes = [vp, vp, sz]
    lib.rt_read_linear.argtypes = [vp, vp, sz]
    lib.rt_device_image.argtypes = [vp, C.POINTER(vp)]
    lib.rt_device_accum.argtypes = [vp, C.POINTER(vp)]
    lib.rt_enable_counters.argtypes = [vp, C.c_int]
    lib.rt_reset_counters.argtypes = [vp]
    lib.rt_get_counters.argtypes = [vp, C.POINTER(_abi.Counters)]
    lib.rt_counters_bytes.argtypes = [C.POINTER(_abi.Counters)]
    lib.rt_counters_bytes.restype = u64
    lib.rt_last_kernel_ms.argtypes = [vp, fp]
    lib.rt_kernel_ms_history.argtypes = [vp, fp, sz, C.POINTER(sz)]
    lib.rt_stage_ms_history.argtypes = [vp, fp, fp, sz, C.POINTER(sz)]
    lib.rt_debug_hit.argtypes = [vp, C.c_int, vp, vp, vp, sz, vp]
    lib.rt_debug_material.argtypes = [vp, C.c_int, vp, sz, vp]
    lib.rt_debug_div3.argtypes = [vp, vp, sz, vp]
    lib.rt_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.rt_shard_slots.argtypes = [vp, C.c_int, C.POINTER(u32)]
    lib.rt_pack_accum.argtypes = [vp, vp, sz]
    lib.rt_unpack_accum.argtypes = [vp, vp, sz, C.c_int, C.c_int]
    lib.rt_device_info.argtypes = [vp, C.c_char_p, sz, C.POINTER(C.c_int), C.c_char_p, sz]
    lib.rt_walk_overflow.argtypes = [vp, C.POINTER(u32)]
    lib.rt_debug_builtin.argtypes = [vp, C.c_int, vp, sz, vp]
    if lib.rt_abi_version() != _abi.RT_ABI_VERSION:
        raise OSError("librt_amd.so ABI %d != expected %d" % (lib.rt_abi_version(), _abi.RT_ABI_VERSION))
    _lib = lib
    return lib


def check_accel(scene):
    """Host-only self-check of the BVHs rt_set_scene would build for `scene` (no device needed).
    → stats dict; raises RtError with the violated invariant."""
    lib = load_library()
    d = scene.desc()
    stats = (C.c_uint64 * 8)()
    err = C.create_string_buffer(256)
    rc = lib.rt_debug_check_accel(C.byref(d), stats, err, 256)
    if rc:
        raise RtError(rc, err.value.decode() or "accel check failed")
    keys = ("sphere_nodes", "sphere_leaves", "sphere_depth", "mesh_nodes", "mesh_leaves", "mesh_depth", "meshes")
    return dict(zip(keys, [int(v) for v in stats]))


def make_random_table(seed):
    """The table rt_set_seed(seed) uploads (host-only; no device needed)."""
    lib = load_library()
    out = np.empty(_abi.RANDOM_TABLE_FLOATS, dtype=np.float32)
    rc = lib.rt_make_random_table(seed, out.ctypes.data, out.size)
    if rc:
        raise RtError(rc, lib.rt_last_error(None).decode())
    return out


def _cam_block(camera):
    block = camera.transferData() if hasattr(camera, "transferData") else camera
    block = np.ascontiguousarray(block, dtype=np.float32)
    if block.shape != (12,):
        raise ValueError("camera block must be 12 floats")
    return block


class DeviceBuffer:
    """A W×H×4 float32 device buffer owned by a RayTracer, exposed through
    ``__cuda_array_interface__`` so torch can wrap it without a copy
    (``torch.as_tensor(buf, device="cuda")``) for the RCCL reduce."""

    def __init__(self, ptr, h, w, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": (h, w, 4), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


class RayTracer:
    # the reference hard-codes "assets/scenes/scene.scene" relative to the working directory (src/raytracer.cpp:95): that
    # path is tried as written first, then this repository's own file of that name
    DEFAULT_SCENE = os.path.join("assets", "scenes", "scene.scene")
    REPO_DEFAULT_SCENE = os.path.join(os.path.dirname(_HERE), "assets", "scenes", "scene.scene")

    def __init__(self, w, h, kernel_path=None, scene=None, device=0, seed=0xC0FFEE):
        """``kernel_path`` is accepted for source compatibility with
        RayTracer(w, h, "kernels/raytracer.cl") and ignored: the kernels are
        compiled into librt_amd.so.  ``scene``: a SceneCreator, a .scene path, or
        None for the default scene file (the reference hard-codes
        "assets/scenes/scene.scene", src/raytracer.cpp:95)."""
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.rt_create(device, w, h, C.byref(self._ctx))
        if rc:
            self._ctx = C.c_void_p()
            raise RtError(rc, self._lib.rt_last_error(None).decode())
        self.width, self.height = w, h
        if seed != 0xC0FFEE:
            self.setSeed(seed)
        if scene is None:
            scene = self.DEFAULT_SCENE if os.path.isfile(self.DEFAULT_SCENE) else self.REPO_DEFAULT_SCENE
        if isinstance(scene, str):
            path = scene
            scene = SceneCreator()
            scene.loadScene(path)
            scene.loadTextures()
        self.setScene(scene)

    # -- plumbing -------------------------------------------------------------------
    def _check(self, rc):
        if rc:
            raise RtError(rc, self._lib.rt_last_error(self._ctx).decode())

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs ---------------------------------------------------------------------
    def setScene(self, scene):
        self.scene = scene
        d = scene.desc()
        self._check(self._lib.rt_set_scene(self._ctx, C.byref(d)))
        tex, tw, th, layers = scene.texture_args()
        self._check(self._lib.rt_set_textures(self._ctx, tex, tw, th, layers))

    def setSeed(self, seed):
        self._check(self._lib.rt_set_seed(self._ctx, seed))

    def setRandomTable(self, table):
        table = np.ascontiguousarray(table, dtype=np.float32)
        self._check(self._lib.rt_set_random_table(self._ctx, table.ctypes.data, table.size))

    def getRandomTable(self):
        out = np.empty(_abi.RANDOM_TABLE_FLOATS, dtype=np.float32)
        self._check(self._lib.rt_get_random_table(self._ctx, out.ctypes.data, out.size))
        return out

    def setShard(self, rank, world, tile_w=8, tile_h=8):
        self._check(self._lib.rt_set_shard(self._ctx, rank, world, tile_w, tile_h))

    OPT_PREFIX_SHARING, OPT_MAX_THREADS_PER_LAUNCH, OPT_SAMPLE_QUEUE, OPT_ACCEL, OPT_WALK_SLICES, OPT_ARITH, OPT_PREFIX_TREE, OPT_WAVE_FILL = 1, 2, 3, 4, 5, 6, 7, 8

    def setOption(self, option, value):
        self._check(self._lib.rt_set_option(self._ctx, option, int(value)))

    def shardSlots(self, world):
        n = C.c_uint32()
        self._check(self._lib.rt_shard_slots(self._ctx, world, C.byref(n)))
        return n.value

    def packAccum(self, device_ptr, nbytes):
        """This rank's owned accumulator pixels → device buffer (slot order)."""
        self._check(self._lib.rt_pack_accum(self._ctx, C.c_void_p(device_ptr), nbytes))

    def unpackAccum(self, device_ptr, nbytes, src_rank, world):
        self._check(self._lib.rt_unpack_accum(self._ctx, C.c_void_p(device_ptr), nbytes, src_rank, world))

    def setStream(self, hip_stream):
        self._check(self._lib.rt_set_stream(self._ctx, C.c_void_p(hip_stream or 0)))

    def resize(self, w, h):
        self._check(self._lib.rt_resize(self._ctx, w, h))
        self.width, self.height = w, h

    def setTime(self, time):
        """Declared, never defined in the reference (include/raytracer.h:45); a stub."""

    # -- rendering --------------------------------------------------------------------
    def render(self, camera):
        self._check(self._lib.rt_render(self._ctx, _cam_block(camera).ctypes.data))

    def renderAgain(self, camera):
        self._check(self._lib.rt_render_again(self._ctx, _cam_block(camera).ctypes.data))

    @property
    def sample_counter(self):
        v = C.c_uint32()
        self._check(self._lib.rt_sample_counter(self._ctx, C.byref(v)))
        return v.value

    def clear(self):
        self._check(self._lib.rt_clear(self._ctx))

    def renderSamples(self, camera, first_sample, n_samples):
        """Fused path: samples first..first+n-1 of every owned pixel in one launch (async)."""
        self._check(self._lib.rt_render_spp(self._ctx, _cam_block(camera).ctypes.data, first_sample, n_samples))

    def resolve(self):
        self._check(self._lib.rt_resolve(self._ctx))

    def sync(self):
        self._check(self._lib.rt_sync(self._ctx))

    def renderFrame(self, camera, spp):
        """clear + fused render of samples 0..spp-1 + resolve + sync → gamma image (h,w,4)."""
        self.clear()
        self.renderSamples(camera, 0, spp)
        self.resolve()
        return self.transferImage()

    def renderFrameOnDevice(self, camera, spp):
        """clear + fused render + resolve, all enqueued, nothing read back (the image stays in HBM)."""
        self.clear()
        self.renderSamples(camera, 0, spp)
        self.resolve()

    def traceSamples(self, camera, xs, ys, samples):
        xs = np.ascontiguousarray(xs, dtype=np.uint32)
        ys = np.ascontiguousarray(ys, dtype=np.uint32)
        ss = np.ascontiguousarray(samples, dtype=np.uint32)
        out = np.zeros((len(xs), 3), dtype=np.float32)
        self._check(self._lib.rt_trace_samples(self._ctx, _cam_block(camera).ctypes.data, xs.ctypes.data,
                                               ys.ctypes.data, ss.ctypes.data, len(xs), out.ctypes.data))
        return out

    # -- outputs ------------------------------------------------------------------------
    def transferImage(self, screen=None, shader_tex_id=None):
        """The reference binds a GL texture (src/raytracer.cpp:167-174); here the
        gamma-space RGBA32F image comes back as an (h, w, 4) array, row 0 = y 0."""
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._lib.rt_read_image(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def readLinear(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._lib.rt_read_linear(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def deviceImage(self):
        p = C.c_void_p()
        self._check(self._lib.rt_device_image(self._ctx, C.byref(p)))
        return DeviceBuffer(p.value, self.height, self.width, self)

    def deviceAccum(self):
        p = C.c_void_p()
        self._check(self._lib.rt_device_accum(self._ctx, C.byref(p)))
        return DeviceBuffer(p.value, self.height, self.width, self)

    # -- measurement ----------------------------------------------------------------------
    def enableCounters(self, on=True):
        self._check(self._lib.rt_enable_counters(self._ctx, 1 if on else 0))

    def resetCounters(self):
        self._check(self._lib.rt_reset_counters(self._ctx))

    def counters(self):
        c = _abi.Counters()
        self._check(self._lib.rt_get_counters(self._ctx, C.byref(c)))
        return c

    def debugCounters(self):
        """(BVH nodes entered, sphere tests executed) since resetCounters(), counting build only."""
        out = (C.c_uint64 * 2)()
        self._check(self._lib.rt_get_debug_counters(self._ctx, out))
        return int(out[0]), int(out[1])

    def walkOverflow(self):
        """Sticky PT_OVF_* bits since resetCounters(): a BVH walk that ended on its loop bound (must be 0)."""
        out = C.c_uint32()
        self._check(self._lib.rt_walk_overflow(self._ctx, C.byref(out)))
        return int(out.value)

    def setArith(self, arith):
        """Select the arithmetic policy of the trace kernels (RT_OPT_ARITH): ARITH_IEEE (default, the CPU oracle's
        contract), ARITH_ROCM_OCL_NOCONTRACT or ARITH_ROCM_OCL (the reference as ROCm's OpenCL builds it); a name of
        ARITH_NAMES is accepted too."""
        if isinstance(arith, str):
            arith = ARITH_NAMES[arith]
        self.setOption(OPT_ARITH, int(arith))
        self.arith = int(arith)

    def lastKernelMs(self):
        ms = C.c_float()
        self._check(self._lib.rt_last_kernel_ms(self._ctx, C.byref(ms)))
        return ms.value

    def kernelMsHistory(self, n=64):
        """Device durations (ms) of the last <= min(n, 64) render launches, oldest first."""
        buf = (C.c_float * n)()
        got = C.c_size_t()
        self._check(self._lib.rt_kernel_ms_history(self._ctx, buf, n, C.byref(got)))
        return [buf[i] for i in range(got.value)]

    def stageMsHistory(self, n=64):
        """(first-stage ms, second-stage ms) of the last <= min(n, 64) render launches, oldest first: pt_prefix and
        the per-sample kernel of a fused call."""
        a, b = (C.c_float * n)(), (C.c_float * n)()
        got = C.c_size_t()
        self._check(self._lib.rt_stage_ms_history(self._ctx, a, b, n, C.byref(got)))
        return [a[i] for i in range(got.value)], [b[i] for i in range(got.value)]

    # -- unit probes of the device routines (tests) -------------------------------------
    def debugHit(self, kind, rays, prim=None, face=None):
        """kind 0 sphere, 1 plane, 2 lens, 3 scene, 4 triangle (mesh prim[i], face[i]) → n × 12 floats (rt_debug_hit)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = len(rays)
        prim = np.zeros(n, np.uint32) if prim is None else np.ascontiguousarray(prim, dtype=np.uint32)
        face = np.zeros(n, np.uint32) if face is None else np.ascontiguousarray(face, dtype=np.uint32)
        out = np.zeros((n, 12), dtype=np.float32)
        self._check(self._lib.rt_debug_hit(self._ctx, kind, rays.ctypes.data, prim.ctypes.data, face.ctypes.data, n,
                                           out.ctypes.data))
        return out

    def debugMaterial(self, routine, vec):
        """routine 0 rayReflect, 1 rayRefract, 2 rayScatter, 3 rayRefractDielectric on n × 16 records → n × 9 floats."""
        vec = np.ascontiguousarray(vec, dtype=np.float32).reshape(-1, 16)
        out = np.zeros((len(vec), 9), dtype=np.float32)
        self._check(self._lib.rt_debug_material(self._ctx, routine, vec.ctypes.data, len(vec), out.ctypes.data))
        return out

    def debugBuiltin(self, op, vec):
        """One builtin of the selected arithmetic policy per record (rt_debug_builtin): n × 8 floats → n × 4 floats."""
        vec = np.ascontiguousarray(vec, dtype=np.float32).reshape(-1, 8)
        out = np.zeros((len(vec), 4), dtype=np.float32)
        self._check(self._lib.rt_debug_builtin(self._ctx, op, vec.ctypes.data, len(vec), out.ctypes.data))
        return out

    def debugDiv3(self, vec):
        vec = np.ascontiguousarray(vec, dtype=np.float32).reshape(-1, 4)
        out = np.zeros((len(vec), 6), dtype=np.float32)
        self._check(self._lib.rt_debug_div3(self._ctx, vec.ctypes.data, len(vec), out.ctypes.data))
        return out

    def deviceInfo(self):
        name, arch, cu = C.create_string_buffer(128), C.create_string_buffer(64), C.c_int()
        self._check(self._lib.rt_device_info(self._ctx, name, 128, C.byref(cu), arch, 64))
        return {"name": name.value.decode(), "arch": arch.value.decode(), "cu_count": cu.value}
