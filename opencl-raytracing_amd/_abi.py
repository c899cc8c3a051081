"""ctypes / numpy mirror of include/rt_amd.h (the C ABI of librt_amd.so).

Layouts are the reference's device structs (kernels/raytracer.cl:25-91,
include/scene.h:32-81); sizes are asserted at import.
"""
import ctypes as C

import numpy as np

RT_ABI_VERSION = 3
DEPTH = 30
RANDOM_BUFFER_SIZE = 100000
RANDOM_TABLE_FLOATS = 4 * RANDOM_BUFFER_SIZE
MAX_DIM = 16384
MAX_SAMPLE = 65535

# enum MatType, kernels/raytracer.cl:23
T_REFRACTIVE, T_REFLECTIVE, T_DIELECTRIC, T_DIFFUSE, T_TEXTURED, T_LIGHT = range(6)
MAT_NAMES = {
    "refractive": T_REFRACTIVE,
    "reflective": T_REFLECTIVE,
    "dielectric": T_DIELECTRIC,
    "diffuse": T_DIFFUSE,
    "textured": T_TEXTURED,
    "light": T_LIGHT,
}

MATERIAL = np.dtype([("type", "<i4"), ("_p0", "<i4", (3,)), ("color", "<f4", (4,)), ("extra_data", "<f4"),
                     ("_p1", "<i4", (3,))])
SPHERE = np.dtype([("pos", "<f4", (4,)), ("r", "<f4"), ("mat_ID", "<u4"), ("_p", "<u4", (2,))])
PLANE = np.dtype([("pos", "<f4", (4,)), ("normal", "<f4", (4,)), ("mat_ID", "<u4"), ("_p", "<u4", (3,))])
LENS = np.dtype([("pos", "<f4", (4,)), ("p1", "<f4", (4,)), ("p2", "<f4", (4,)), ("r1", "<f4"), ("r2", "<f4"),
                 ("mat_ID", "<u4"), ("_p", "<u4")])
MESH = np.dtype([("vertex_anchor", "<u4"), ("index_anchor", "<u4"), ("face_count", "<u4"), ("texture_ID", "<u4")])
MODEL = np.dtype([("mesh_anchor", "<u4"), ("mesh_count", "<u4"), ("mat_ID", "<u4")])
assert (MATERIAL.itemsize, SPHERE.itemsize, PLANE.itemsize, LENS.itemsize, MESH.itemsize, MODEL.itemsize) == \
    (48, 32, 48, 64, 16, 12)


class SceneDesc(C.Structure):
    """rt_scene_desc"""
    _fields_ = [(n, C.c_void_p) for n in ("materials", "spheres", "planes", "lenses", "vertices", "uvs", "indices",
                                          "meshes", "models")] + \
               [(n, C.c_uint32) for n in ("material_count", "sphere_count", "plane_count", "lens_count",
                                          "vertex_count", "uv_count", "index_count", "mesh_count", "model_count",
                                          "_pad")]


COUNTER_FIELDS = ("samples", "bounces", "t_sphere", "t_plane", "t_lens", "t_model", "t_mesh", "t_tri", "h_tri",
                  "h_bounce", "n_scatter", "n_dielectric", "n_texfetch", "image_reads")


class Counters(C.Structure):
    """rt_counters"""
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}

    def algorithmic_bytes(self):
        """SURVEY §8d B_alg: the reference kernel's logical global-memory traffic."""
        c = self
        return (32 * c.t_sphere + 48 * c.t_plane + 64 * c.t_lens + 12 * c.t_model + 16 * c.t_mesh + 60 * c.t_tri +
                36 * c.h_tri + 48 * c.h_bounce + 12 * c.n_scatter + 4 * c.n_dielectric + 64 * c.n_texfetch +
                64 * c.samples + 16 * c.image_reads)


def ptr(a):
    """void* of a numpy array (None/empty → NULL)."""
    if a is None or a.size == 0:
        return None
    return a.ctypes.data_as(C.c_void_p)
