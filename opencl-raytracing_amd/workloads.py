"""Synthetic workloads C1–C5 (BASELINE.json `configs`).

Every workload is a pure function of its parameters and a seed (Philox-4x32-10
counters, integer arithmetic only → identical on any host).  The same arrays
feed the oracle and the GPU.  ``get(name)`` → Workload(scene, camera block, w, h, spp).

Scene conventions follow the reference: world up is −y (src/camera.cpp:23), the
floor is a plane at y = +5 and the light is a huge emissive sphere far above
(assets/scenes/scene.scene:20,25 in the reference).
"""
import os
from dataclasses import dataclass

import numpy as np

from . import _abi
from .camera import Camera
from .scene import SceneCreator, _identity, _radians, _rotate

f32 = np.float32
SEED = 0xC0FFEE
_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


# ---- Philox-4x32-10, vectorised ----------------------------------------------------
def philox4x32(counter0, counter1, seed):
    """counter0/1: uint32 arrays (c2 = c3 = 0), key = (seed lo, seed hi) → (n,4) uint32."""
    c0 = np.asarray(counter0, dtype=np.uint64)
    c1 = np.broadcast_to(np.asarray(counter1, dtype=np.uint64), c0.shape).copy()
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M
        n1 = p1 & M
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M
        n3 = p0 & M
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def u24(words):
    """uint32 → float32 in [0,1) with 24 random bits."""
    return (words >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)


def uniforms(n, stream, seed=SEED):
    """n×4 float32 U[0,1), counter = (i, stream)."""
    return u24(philox4x32(np.arange(n, dtype=np.uint32), np.uint32(stream), seed))


def make_random_table(seed=SEED):
    """numpy twin of rt_make_random_table (include/rt_amd.h): the 400 000-float table."""
    n = _abi.RANDOM_BUFFER_SIZE
    idx = np.arange(n, dtype=np.uint32)
    w = philox4x32(idx, np.uint32(0), seed)
    table = np.zeros(4 * n, dtype=f32)
    table[3 * n:] = u24(w[:, 3])
    xyz = np.zeros((n, 3), dtype=f32)
    todo = np.arange(n)
    attempt = 0
    while len(todo):
        if attempt:
            w = philox4x32(idx[todo], np.uint32(attempt), seed)
        p = f32(2.0) * u24(w[:, :3]) - f32(1.0)
        ok = (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2] < f32(1.0)
        xyz[todo[ok]] = p[ok]
        todo = todo[~ok]
        w = None
        attempt += 1
    table[:3 * n] = xyz.reshape(-1)
    return table


# ---- workloads -----------------------------------------------------------------------
@dataclass
class Workload:
    name: str
    scene: SceneCreator
    camera: np.ndarray  # float32[12]
    width: int
    height: int
    spp: int
    description: str = ""


def _camera(width, height, pos, yaw, pitch=0.0, fov=60):
    return Camera(fov, f32(width) / f32(height), pos, yaw, pitch).transferData()


def c1(width=256, height=256, spp=1):
    """C1: single sphere + plane, 256×256, 1 spp (plumbing)."""
    s = SceneCreator()
    s.loadScene(os.path.join(_ASSETS, "scenes", "c1_sphere.scene"))
    return Workload("c1", s, _camera(width, height, (0, 0, 0), 0.0), width, height, spp,
                    "1 sphere + 1 plane (assets/scenes/c1_sphere.scene)")


def c2(width=1920, height=1080, spp=64):
    """C2: Cornell-style: 8 spheres (2 reflective, 1 refractive, 2 diffuse, 1 light r=100
    overhead, 2 dielectric) + diffuse floor plane; the bench's headline workload."""
    s = SceneCreator()
    s.loadScene(os.path.join(_ASSETS, "scenes", "c2_cornell.scene"))
    return Workload("c2", s, _camera(width, height, (-8, -1, -8), 45.0), width, height, spp,
                    "8 spheres + 1 plane (assets/scenes/c2_cornell.scene)")


def checker_texture(size=1024, cells=8):
    """Procedural RGBA32F checker with a colour ramp (one layer)."""
    j, i = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    on = (((i * cells) // size + (j * cells) // size) % 2).astype(f32)
    tex = np.empty((1, size, size, 4), dtype=f32)
    tex[0, :, :, 0] = f32(0.15) + f32(0.8) * on
    tex[0, :, :, 1] = f32(0.2) + f32(0.7) * (i.astype(f32) / f32(size))
    tex[0, :, :, 2] = f32(0.9) - f32(0.6) * on * (j.astype(f32) / f32(size))
    tex[0, :, :, 3] = f32(1.0)
    return tex


def c3(width=1920, height=1080, spp=256, tex_size=1024):
    """C3: textured 12-triangle cube (OBJ, rotated 45° about y) + 4 spheres."""
    s = SceneCreator()
    s.loadScene(os.path.join(_ASSETS, "scenes", "c3_cube.scene"), base_dir=_ASSETS)
    s.setTextures(checker_texture(tex_size))
    return Workload("c3", s, _camera(width, height, (-6, -2, -7), 40.0, 8.0), width, height, spp,
                    "12-tri textured cube + 4 spheres + plane (assets/scenes/c3_cube.scene)")


def c4(width=1920, height=1080, spp=64, n_spheres=100000, seed=SEED):
    """C4: n random spheres + ground plane + one r=100 light overhead."""
    s = SceneCreator()
    for col in ((0.9, 0.2, 0.2), (0.2, 0.9, 0.3), (0.25, 0.35, 0.95), (0.9, 0.9, 0.2), (0.85, 0.85, 0.85)):
        s.addMaterial(_abi.T_DIFFUSE, col, 1)          # 0..4
    s.addMaterial(_abi.T_REFLECTIVE, (1, 1, 1), 0.8)   # 5
    s.addMaterial(_abi.T_DIELECTRIC, (1, 1, 1), 1.3)   # 6
    s.addMaterial(_abi.T_LIGHT, (1, 1, 1), 0)          # 7
    s.addMaterial(_abi.T_DIFFUSE, (1, 1, 1), 1)        # 8 floor
    n = n_spheres - 1
    u = uniforms(n, 1, seed)
    pos = np.empty((n, 3), dtype=f32)
    pos[:, 0] = f32(-50.0) + f32(100.0) * u[:, 0]
    pos[:, 1] = f32(-2.0) + f32(6.5) * u[:, 1]
    pos[:, 2] = f32(-50.0) + f32(100.0) * u[:, 2]
    r = f32(0.1) + f32(0.4) * u[:, 3]
    k = np.arange(n)
    mat = (k % 7).astype(np.uint32)          # diffuse×5, reflective, dielectric round-robin
    mat[k % 50 == 49] = 7                    # 1 in 50 emissive
    s.addSphere((1, -200, 0), 100, 7)        # the light overhead is sphere 0
    s.addSpheres(pos, r, mat)
    s.addPlane((0, 5, 0), (0, 1, 0), 8)
    return Workload("c4", s, _camera(width, height, (-8, -1, -8), 45.0), width, height, spp,
                    "%d random spheres + plane" % n_spheres)


def uv_sphere_grid(segments=200, rings=126, radius=1.0, centre=(0.0, 0.0, 0.0)):
    """Closed convex uv-sphere as an indexed mesh: (rings+1)·(segments+1) grid vertices (float64 positions, uv) and
    2·segments·(rings−1) triangles as index triples into the grid, every face counter-clockwise seen from outside so
    that the reference's hitMeshOut assumptions hold (raytracer.cl:284,292).  Degenerate triangles at the poles are
    left out."""
    th = (np.arange(rings + 1, dtype=np.float64) / rings) * np.pi            # polar
    ph = (np.arange(segments + 1, dtype=np.float64) / segments) * 2 * np.pi  # azimuth
    ii, jj = np.meshgrid(np.arange(rings + 1), np.arange(segments + 1), indexing="ij")
    unit = np.stack([np.sin(th[ii]) * np.cos(ph[jj]), np.cos(th[ii]) * np.ones_like(ph[jj]), np.sin(th[ii]) * np.sin(ph[jj])],
                    axis=-1).reshape(-1, 3)
    uv = np.stack([ph[jj] / (2 * np.pi), th[ii] / np.pi * np.ones_like(ph[jj])], axis=-1).reshape(-1, 2)

    def vid(i, j):
        return i * (segments + 1) + j

    tris = []
    j = np.arange(segments)
    for i in range(rings):
        a, b, c, d = vid(i, j), vid(i, j + 1), vid(i + 1, j + 1), vid(i + 1, j)
        if i > 0:
            tris.append(np.stack([a, c, b], axis=1))      # upper triangle (degenerate at the north pole → skipped)
        if i < rings - 1:
            tris.append(np.stack([a, d, c], axis=1))      # lower triangle (degenerate at the south pole → skipped)
    tri = np.concatenate(tris)
    # make every face counter-clockwise seen from outside (normal·centroid > 0)
    P = unit[tri]
    nrm = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    flip = np.einsum("ij,ij->i", nrm, P.mean(axis=1)) < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]
    return unit * radius + np.asarray(centre), uv, tri.astype(np.uint32)


def uv_sphere(segments=200, rings=126, radius=1.0, centre=(0.0, 0.0, 0.0)):
    """The same sphere with one vertex per face corner (the layout Assimp emits, src/scene.cpp:234-262):
    positions, uvs, indices 0..n-1."""
    pos, uv, tri = uv_sphere_grid(segments, rings, radius, centre)
    return pos[tri.reshape(-1)].astype(f32), uv[tri.reshape(-1)].astype(f32), np.arange(3 * len(tri), dtype=np.uint32)


C5_OBJ = os.path.join(_ASSETS, "models", "c5_sphere.obj")


def write_c5_obj(path=C5_OBJ, segments=200, rings=126):
    """assets/models/c5_sphere.obj (tools/make_c5_obj.py): BASELINE's "50k-triangle OBJ mesh" as a Wavefront file —
    the grid vertices once, faces as v/vt index triples.  Coordinates are written with 9 significant digits (every
    binary32 value survives the trip through text); the v coordinate of `vt` is stored flipped, the way an OBJ
    exporter writes it and aiProcess_FlipUVs (src/scene.cpp:195) undoes it."""
    pos, uv, tri = uv_sphere_grid(segments, rings, radius=2.5, centre=(0.0, 2.0, 0.0))
    p32, uv32 = pos.astype(f32), uv.astype(f32)
    with open(path, "w") as fh:
        fh.write("# c5_sphere: closed uv-sphere, %d segments x %d rings, %d triangles, radius 2.5 about (0, 2, 0);\n"
                 "# counter-clockwise seen from outside.  Written by tools/make_c5_obj.py — do not edit.\no c5_sphere\n" %
                 (segments, rings, len(tri)))
        for x, y, z in p32:
            fh.write("v %.9g %.9g %.9g\n" % (x, y, z))
        for u, v in uv32:
            fh.write("vt %.9g %.9g\n" % (u, f32(1.0) - v))
        for a, b, c in tri + 1:
            fh.write("f %d/%d %d/%d %d/%d\n" % (a, a, b, b, c, c))
    return path


def c5(width=3840, height=2160, spp=512, segments=200, rings=126):
    """C5: 50 000-triangle dielectric OBJ mesh on a diffuse plane under a light sphere (assets/scenes/c5_mesh.scene →
    assets/models/c5_sphere.obj through the OBJ reader).  Other segment / ring counts (tests, rehearsals) build the
    same scene from the generator directly."""
    s = SceneCreator()
    if (segments, rings) == (200, 126):
        if not os.path.isfile(C5_OBJ):
            write_c5_obj()
        s.loadScene(os.path.join(_ASSETS, "scenes", "c5_mesh.scene"), base_dir=_ASSETS)
    else:
        s.addMaterial(_abi.T_DIELECTRIC, (1, 1, 1), 1.3)  # 0
        s.addMaterial(_abi.T_DIFFUSE, (0.8, 0.8, 0.8), 1)  # 1
        s.addMaterial(_abi.T_LIGHT, (1, 1, 1), 0)          # 2
        s.addMaterial(_abi.T_DIFFUSE, (0.9, 0.3, 0.2), 1)  # 3
        pos, uv, idx = uv_sphere(segments, rings, radius=2.5, centre=(0.0, 2.0, 0.0))
        s.addMesh(pos, uv, idx)
        s.addModel(1, 0)
        s.addSphere((1, -200, 0), 100, 2)
        s.addSphere((-4.5, 3.5, 1.0), 1.5, 3)
        s.addPlane((0, 5, 0), (0, 1, 0), 1)
    return Workload("c5", s, _camera(width, height, (-7, 0, -7), 45.0, 8.0), width, height, spp,
                    "%d-triangle dielectric uv-sphere (OBJ) + 2 spheres + plane" % (len(s.indices) // 3))


def shipped_like(width=1200, height=800, spp=16):
    """Every primitive and material kind in one frame (the shape of the reference's
    shipped scene: spheres, plane, lens, two textured cubes) — parity test scene."""
    s = SceneCreator()
    s.loadScene(os.path.join(_ASSETS, "scenes", "all_kinds.scene"), base_dir=_ASSETS)
    s.setTextures(np.concatenate([checker_texture(64, 4), checker_texture(64, 8)[:, ::-1].copy()]))
    return Workload("all_kinds", s, _camera(width, height, (-8, -1, -8), 45.0), width, height, spp,
                    "spheres + plane + lens + 2 textured cubes (assets/scenes/all_kinds.scene)")


def untextured(wl):
    """The same workload with every t_textured material turned into t_diffuse and no texture array: the reference's
    real OpenCL build (oracle/_ref_gfx950) can only be driven without an OpenCL image object, and everything but the
    texel fetch (raytracer.cl:105-107) — lenses, small meshes by face scan, uv interpolation — is still exercised."""
    wl.scene.materials["type"][wl.scene.materials["type"] == _abi.T_TEXTURED] = _abi.T_DIFFUSE
    wl.scene.textures = None
    wl.scene.texture_paths = []
    wl.name += "_untextured"
    return wl


_REGISTRY = {"c1": c1, "c2": c2, "c3": c3, "c4": c4, "c5": c5, "all_kinds": shipped_like}


def get(name, **kw):
    return _REGISTRY[name](**kw)
