"""SceneCreator — host-side scene model, mirror of the reference's class
(include/scene.h:83-153, src/scene.cpp:110-461).

Produces the nine arrays + counts the trace kernels read, in exactly the
reference's device layouts (see _abi.py), and hands them to the C ABI as an
``rt_scene_desc``.  Same method names and argument meaning as the reference:
``addMaterial / addSphere / addPlane / addLens / loadModel / loadScene /
loadTextures``; scene errors raise ``SceneError`` where the reference prints and
calls ``exit(-1)`` (src/scene.cpp:29-32).

Importer boundary: the reference reads meshes through Assimp and
textures through stb_image, neither of which exists here.  ``loadModel`` is a
minimal OBJ reader that reproduces Assimp's observable output for
``aiProcess_Triangulate | aiProcess_FlipUVs`` (src/scene.cpp:195): one vertex
per face corner in file order, fan triangulation, v → 1−v.  Parity with Assimp
/ stb is unpinned; parity of the kernels is pinned at the kernel-input boundary
(the arrays built here feed oracle and GPU alike).
"""
import math
import os
import re

import numpy as np

from . import _abi
from ._abi import LENS, MATERIAL, MESH, MODEL, PLANE, SPHERE

f32 = np.float32


class SceneError(RuntimeError):
    pass


# ---- glm-style column-major float32 4x4 helpers (gtc/matrix_transform) ---------
def _identity():
    return np.eye(4, dtype=f32)  # m[col][row] stored as m[col, row]


def _translate(m, v):
    r = m.copy()
    r[3] = m[0] * f32(v[0]) + m[1] * f32(v[1]) + m[2] * f32(v[2]) + m[3]
    return r


def _scale(m, v):
    r = m.copy()
    r[0] = m[0] * f32(v[0])
    r[1] = m[1] * f32(v[1])
    r[2] = m[2] * f32(v[2])
    return r


def _rotate(m, angle, axis):
    from .camera import cosf, sinf  # libm's single-precision functions, as glm::rotate<float> uses
    a = f32(angle)
    c, s = cosf(a), sinf(a)
    ax = np.asarray(axis, dtype=f32)
    ax = ax / f32(np.sqrt(f32(np.dot(ax, ax))))
    t = (f32(1) - c) * ax
    rot = np.zeros((3, 3), dtype=f32)
    rot[0, 0] = c + t[0] * ax[0]
    rot[0, 1] = t[0] * ax[1] + s * ax[2]
    rot[0, 2] = t[0] * ax[2] - s * ax[1]
    rot[1, 0] = t[1] * ax[0] - s * ax[2]
    rot[1, 1] = c + t[1] * ax[1]
    rot[1, 2] = t[1] * ax[2] + s * ax[0]
    rot[2, 0] = t[2] * ax[0] + s * ax[1]
    rot[2, 1] = t[2] * ax[1] - s * ax[0]
    rot[2, 2] = c + t[2] * ax[2]
    r = m.copy()
    for i in range(3):
        r[i] = m[0] * rot[i, 0] + m[1] * rot[i, 1] + m[2] * rot[i, 2]
    return r


def _radians(deg):
    return f32(deg) * f32(0.01745329251994329576923690768489)


def _f3(v):
    a = np.zeros(4, dtype=f32)
    a[:3] = np.asarray(v, dtype=f32)[:3]
    return a


class SceneCreator:
    def __init__(self):
        self.materials = np.zeros(0, dtype=MATERIAL)
        self.spheres = np.zeros(0, dtype=SPHERE)
        self.planes = np.zeros(0, dtype=PLANE)
        self.lenses = np.zeros(0, dtype=LENS)
        self.meshes = np.zeros(0, dtype=MESH)
        self.models = np.zeros(0, dtype=MODEL)
        self.vertices = np.zeros((0, 4), dtype=f32)
        self.texture_uv = np.zeros((0, 2), dtype=f32)
        self.indices = np.zeros(0, dtype=np.uint32)
        self.texture_paths = []
        self.model_dirs = []   # directories of the OBJ files loaded so far (texture lookup)
        self.scene_dir = ""
        self.base_dir = ""
        self.textures = None  # (layers, h, w, 4) float32
        self._mesh_count_total = 0  # `static cl_uint mesh_count_total`, src/scene.cpp:193
        self._keep = None

    # -- src/scene.cpp:110-143 -----------------------------------------------------
    def addMaterial(self, type, color, extra_data):
        m = np.zeros(1, dtype=MATERIAL)
        m["type"] = int(type)
        m["color"][0] = _f3(color)
        m["extra_data"] = f32(extra_data)
        self.materials = np.concatenate([self.materials, m])

    def addSphere(self, pos, r, mat_ID):
        s = np.zeros(1, dtype=SPHERE)
        s["pos"][0] = _f3(pos)
        s["r"] = f32(r)
        s["mat_ID"] = int(mat_ID)
        self.spheres = np.concatenate([self.spheres, s])

    def addSpheres(self, pos, r, mat_ID):
        """Bulk form of addSphere (n×3 positions) for generated scenes."""
        n = len(r)
        s = np.zeros(n, dtype=SPHERE)
        s["pos"][:, :3] = np.asarray(pos, dtype=f32)
        s["r"] = np.asarray(r, dtype=f32)
        s["mat_ID"] = np.asarray(mat_ID, dtype=np.uint32)
        self.spheres = np.concatenate([self.spheres, s])

    def addPlane(self, pos, normal, mat_ID):
        p = np.zeros(1, dtype=PLANE)
        p["pos"][0] = _f3(pos)
        p["normal"][0] = _f3(normal)
        p["mat_ID"] = int(mat_ID)
        self.planes = np.concatenate([self.planes, p])

    def addLens(self, pos, normal, r1, r2, h, mat_ID):
        r1, r2, h = f32(r1), f32(r2), f32(h)
        if not (r1 >= h and r2 >= h):  # assert, src/scene.cpp:123
            raise SceneError("ERROR: SCENE: LENS RADII MUST BE >= h")
        pos, normal = _f3(pos), _f3(normal)
        l = np.zeros(1, dtype=LENS)
        l["pos"][0] = pos
        t1 = f32(np.sqrt(r1 * r1 - h * h))
        t2 = f32(np.sqrt(r2 * r2 - h * h))
        l["p1"][0] = pos + normal * t1
        l["p2"][0] = pos - normal * t2
        l["r1"], l["r2"], l["mat_ID"] = r1, r2, int(mat_ID)
        self.lenses = np.concatenate([self.lenses, l])

    # -- src/scene.cpp:192-295 (Assimp replaced by a minimal OBJ reader) -----------
    def addMesh(self, vertices, uvs, indices, texture_ID=0xFFFFFFFF):
        """Append one mesh (vertex positions n×3, uvs n×2 or None, triangle indices
        3·faces, mesh-local) exactly as processMesh does (src/scene.cpp:234-295)."""
        vertices = np.asarray(vertices, dtype=f32).reshape(-1, 3)
        indices = np.asarray(indices, dtype=np.uint32).reshape(-1)
        mesh = np.zeros(1, dtype=MESH)
        mesh["vertex_anchor"] = len(self.vertices)
        mesh["index_anchor"] = len(self.indices)
        mesh["face_count"] = len(indices) // 3
        mesh["texture_ID"] = texture_ID & 0xFFFFFFFF
        v4 = np.zeros((len(vertices), 4), dtype=f32)
        v4[:, :3] = vertices
        # meshes without uv leave the reference's uv array short (src/scene.cpp:246-252),
        # which makes the kernel's uv fetch read out of bounds; here uv is zero-filled
        uv = np.zeros((len(vertices), 2), dtype=f32) if uvs is None else np.asarray(uvs, dtype=f32).reshape(-1, 2)
        if len(self.texture_uv) < len(self.vertices):
            pad = np.zeros((len(self.vertices) - len(self.texture_uv), 2), dtype=f32)
            self.texture_uv = np.concatenate([self.texture_uv, pad])
        self.vertices = np.concatenate([self.vertices, v4])
        self.texture_uv = np.concatenate([self.texture_uv, uv])
        self.indices = np.concatenate([self.indices, indices])
        self.meshes = np.concatenate([self.meshes, mesh])

    def addModel(self, mesh_count, mat_ID):
        """models.push_back({mesh_count_total, mesh_count, mat_ID}), src/scene.cpp:205-207"""
        mo = np.zeros(1, dtype=MODEL)
        mo["mesh_anchor"], mo["mesh_count"], mo["mat_ID"] = self._mesh_count_total, mesh_count, int(mat_ID)
        self.models = np.concatenate([self.models, mo])
        self._mesh_count_total += mesh_count

    def _texture_id(self, path):
        for j, p in enumerate(self.texture_paths):  # dedup by path string, src/scene.cpp:272-283
            if p == path:
                return j
        self.texture_paths.append(path)
        return len(self.texture_paths) - 1

    def loadModel(self, path, mat_ID, transform=None):
        if transform is None:
            transform = _identity()
        if len(self.materials) <= mat_ID:
            raise SceneError("ERROR: MATERIAL OF ID: %d DOES NOT EXIST" % mat_ID)
        if not os.path.isfile(path):
            raise SceneError("ERROR: Assimp: Unable to open file \"%s\"." % path)
        textured = int(self.materials["type"][mat_ID]) == _abi.T_TEXTURED
        meshes = _read_obj(path)
        d = os.path.dirname(path) or "."
        if d not in self.model_dirs:
            self.model_dirs.append(d)
        for pos, uv, idx, tex_path in meshes:
            # transformVertex, src/scene.cpp:226-232: column-major mat4 × (x,y,z,1)
            m = transform
            x, y, z = pos[:, 0], pos[:, 1], pos[:, 2]
            out = np.empty_like(pos)
            for k in range(3):
                out[:, k] = m[0, k] * x + m[1, k] * y + m[2, k] * z + m[3, k]
            tex_id = 0xFFFFFFFF
            if textured:
                if tex_path is None:
                    raise SceneError("ERROR: MESH HAS NO TEXTURE APPLIED, USE A DIFFERENT MATERIAL")
                tex_id = self._texture_id(tex_path)
            self.addMesh(out, uv, idx, tex_id)
        self.addModel(len(meshes), mat_ID)

    # -- src/scene.cpp:145-190 ------------------------------------------------------
    @staticmethod
    def _ldr_to_hdr_tables():
        """stb_image's 8-bit → float rule (stbi__ldr_to_hdr, default gamma 2.2f / scale 1.0f — what
        stbi_loadf applies, src/scene.cpp:158): colour = (float)pow(byte / 255.0f, 2.2f) with a FLOAT
        quotient and the float constant 2.2f promoted to double for pow(); alpha = byte / 255.0f.
        stb_image is not in the reference tree, so this restates its published formula (parity
        unpinned).  math.pow is libm's pow, as in the C++ mirror (host/scene.cpp)."""
        import math
        q = (np.arange(256, dtype=f32) / f32(255.0)).astype(f32)
        g = float(f32(2.2))
        colour = np.array([math.pow(float(x), g) for x in q], dtype=np.float64).astype(f32)
        return colour, q

    def _texture_dirs(self, search_dirs):
        """Where a texture is looked up by BASE NAME when its map_Kd path does not exist as written —
        the reference's own .mtl files name an absolute path on the author's machine
        (assets/cube/cube.mtl:13).  Same order as host/scene.cpp."""
        dirs = []
        for d in list(search_dirs) + [os.path.join(m, "..", "textures") for m in self.model_dirs] + \
                list(self.model_dirs) + ([os.path.join(self.scene_dir, "..", "textures")] if self.scene_dir else []) + \
                [os.path.join("assets", "textures")] + \
                ([os.path.join(self.base_dir, "textures"), self.base_dir] if self.base_dir else []):
            if d not in dirs:
                dirs.append(d)
        return dirs

    def loadTextures(self, search_dirs=()):
        if len(self.models) == 0:
            self.textures = None
            return
        if not self.texture_paths:
            # the reference exits with "ERROR: TEXTURE COUNT = 0" here (src/scene.cpp:147-148) for every scene whose
            # models are not t_textured; accepted instead (BASELINE's configuration 5 is such a scene): no layers
            self.textures = None
            return
        from PIL import Image
        colour, alpha = self._ldr_to_hdr_tables()
        layers = []
        for p in self.texture_paths:
            pn = p.replace("\\", "/")
            cand = [p]
            if not os.path.isabs(pn):
                cand += ([os.path.join(self.base_dir, pn)] if self.base_dir else []) + \
                        [os.path.join(m, pn) for m in self.model_dirs]
            cand += [os.path.join(d, os.path.basename(pn)) for d in self._texture_dirs(search_dirs)]
            found = next((c for c in cand if os.path.isfile(c)), None)
            if found is None:
                raise SceneError("ERROR: STBimage: COULD NOT FIND THE TEXTURE: " + p)
            im = Image.open(found)
            if im.mode != "RGBA":
                raise SceneError("ERROR: STBimage: TEXTURE HAS A WRONG FORMAT: %d INSTEAD OF 4 (RGBA)" %
                                 len(im.getbands()))
            px = np.asarray(im, dtype=np.uint8)
            a = np.empty(px.shape, dtype=f32)
            a[..., :3] = colour[px[..., :3]]
            a[..., 3] = alpha[px[..., 3]]
            if layers and a.shape != layers[0].shape:
                raise SceneError("ERROR: TEXTURES HAVE DIFFERENT SIZES")
            layers.append(a)
        self.textures = np.ascontiguousarray(np.stack(layers))

    def setTextures(self, rgba):
        """Install texture layers directly: (layers, h, w, 4) float32."""
        rgba = np.ascontiguousarray(rgba, dtype=f32)
        if rgba.ndim != 4 or rgba.shape[3] != 4:
            raise SceneError("textures must be (layers, h, w, 4)")
        self.textures = rgba

    # -- src/scene.cpp:297-403 ------------------------------------------------------
    def loadScene(self, path, base_dir=None):
        """``load:`` paths are used as written — relative to the working directory, like the
        reference (src/scene.cpp:355 → :195; its scene file says "assets/cube/cube.obj") — and only
        when that file does not exist against ``base_dir`` (default: the parent of the scene's
        directory, assets/scenes/x.scene → assets/) and the scene's own directory."""
        try:
            with open(path, "r") as fh:
                text = fh.read()
        except OSError as e:
            raise SceneError("ERROR: SCENE: NOT SUCCESFULLY READ: " + str(e))
        self.scene_dir = os.path.dirname(path)
        if base_dir is None:
            base_dir = os.path.dirname(self.scene_dir)
        self.loadSceneText(text, base_dir)

    def _resolve_model_path(self, p):
        if os.path.isfile(p) or os.path.isabs(p):
            return p
        for d in (self.base_dir, self.scene_dir):
            if d and os.path.isfile(os.path.join(d, p)):
                return os.path.join(d, p)
        return p

    def loadSceneText(self, text, base_dir=""):
        self.base_dir = base_dir or ""
        mode = None
        model = _identity()
        for line in text.split("\n"):
            pos = line.find("#")
            if pos != -1:
                line = line[:pos]
            if line == "":
                continue
            pos = line.find(":")
            if pos != -1:
                word = line[:pos]
                if word in ("MATERIALS", "SPHERES", "PLANES", "LENSES", "MODELS"):
                    mode = word
                    continue
                if mode == "MODELS":
                    it = _Fields(line[pos + 1:])
                    if word == "translate":
                        model = _translate(model, it.vec())
                    elif word == "rotate":
                        ang = _radians(it.float())
                        model = _rotate(model, ang, it.vec())
                    elif word == "scale":
                        model = _scale(model, it.vec())
                    elif word == "load":
                        p = it.path()
                        mat = it.uint()
                        self.loadModel(self._resolve_model_path(p), mat, model)
                        model = _identity()
                    # other words with a colon inside MODELS are ignored by the reference
                else:
                    raise SceneError("ERROR: SCENE: OPERATION " + word + " DOES NOT EXIST")
            else:
                it = _Fields(line)
                if mode == "MATERIALS":
                    word = it.raw()
                    if word not in _abi.MAT_NAMES:
                        raise SceneError("ERROR: SCENE: MATERIAL: " + word + " DOES NOT EXIST")
                    self.addMaterial(_abi.MAT_NAMES[word], it.vec(), it.float())
                elif mode == "SPHERES":
                    self.addSphere(it.vec(), it.float(), it.uint())
                elif mode == "PLANES":
                    self.addPlane(it.vec(), it.vec(), it.uint())
                elif mode == "LENSES":
                    self.addLens(it.vec(), it.vec(), it.float(), it.float(), it.float(), it.uint())
                else:
                    raise SceneError("ERROR: SCENE: OPERATION NOT SPECIFIED")

    # -- hand-off to the C ABI (SceneCreator::setupBuffers/createScene/setKernelArgs) --
    def desc(self):
        """rt_scene_desc pointing into this object's arrays (kept alive by self)."""
        uv = self.texture_uv
        if len(uv) < len(self.vertices):
            uv = np.concatenate([uv, np.zeros((len(self.vertices) - len(uv), 2), dtype=f32)])
        arrs = dict(materials=np.ascontiguousarray(self.materials), spheres=np.ascontiguousarray(self.spheres),
                    planes=np.ascontiguousarray(self.planes), lenses=np.ascontiguousarray(self.lenses),
                    vertices=np.ascontiguousarray(self.vertices), uvs=np.ascontiguousarray(uv),
                    indices=np.ascontiguousarray(self.indices), meshes=np.ascontiguousarray(self.meshes),
                    models=np.ascontiguousarray(self.models))
        d = _abi.SceneDesc()
        for k, a in arrs.items():
            setattr(d, k, _abi.ptr(a))
        d.material_count, d.sphere_count, d.plane_count = len(self.materials), len(self.spheres), len(self.planes)
        d.lens_count, d.vertex_count, d.uv_count = len(self.lenses), len(self.vertices), len(uv)
        d.index_count, d.mesh_count, d.model_count = len(self.indices), len(self.meshes), len(self.models)
        self._keep = arrs
        return d

    def texture_args(self):
        """(ptr, w, h, layers) for rt_set_textures / the oracle."""
        if self.textures is None:
            return None, 0, 0, 0
        l, h, w, _ = self.textures.shape
        return _abi.ptr(self.textures), w, h, l


# ---- field tokenizer of the .scene format (src/scene.cpp:405-461) ---------------
_NUM = r"(?=[\d.+-])([-+]?\d*(?:\.\d+)?)"  # decimal only, no exponent (src/scene.cpp:423,448)
_RE_VEC = re.compile(r"\s*\(" + _NUM + r",\s*" + _NUM + r",\s*" + _NUM + r"\)\s*$")
_RE_FLT = re.compile(r"\s*" + _NUM + r"\s*$")
_RE_UINT = re.compile(r"\s*(\d)\s*$")  # a single digit, src/scene.cpp:455
_RE_PATH = re.compile(r"\s*\"(.*?)\"\s*$")
_RE_DELIM = re.compile(r",(?![^(]*\))")  # commas outside parentheses, src/scene.cpp:314


class _Fields:
    """Left-to-right consumption of comma separated fields (the reference relies
    on clang's left-to-right argument evaluation, src/scene.cpp:351-392)."""

    def __init__(self, line):
        self.f = _RE_DELIM.split(line)
        self.i = 0

    def raw(self):
        if self.i >= len(self.f):
            raise SceneError("ERROR: SCENE: NOT ENOUGH PARAMETERS")
        w = self.f[self.i]
        self.i += 1
        return w

    def vec(self):
        w = self.raw()
        m = _RE_VEC.match(w)
        if not m:
            raise SceneError("ERROR: SCENE: IMPROPER VECTOR: " + w)
        try:
            return [f32(float(g)) for g in m.groups()]
        except ValueError:
            raise SceneError("ERROR: SCENE: IMPROPER VECTOR: " + w)

    def float(self):
        w = self.raw()
        m = _RE_FLT.match(w)
        if not m:
            raise SceneError("ERROR: SCENE: IMPROPER FLOAT: " + w)
        try:
            return f32(float(m.group(1)))
        except ValueError:
            raise SceneError("ERROR: SCENE: IMPROPER FLOAT: " + w)

    def uint(self):
        w = self.raw()
        m = _RE_UINT.match(w)
        if not m:
            raise SceneError("ERROR: SCENE: IMPROPER UNSIGNED INT: " + w)
        return int(m.group(1))

    def path(self):
        w = self.raw()
        m = _RE_PATH.match(w)
        if not m:
            raise SceneError("ERROR: SCENE: IMPROPER PATH: " + w)
        return m.group(1)


# ---- minimal OBJ reader ------------------------------------------------------------
def _read_obj(path):
    """→ list of meshes (positions n×3, uv n×2 or None, indices, diffuse texture path or None).
    One mesh per (object/group, material) run, one vertex per face corner, quads and
    n-gons fan-triangulated (0,1,2),(0,2,3),..., v → 1−v."""
    v, vt = [], []
    meshes = []
    cur = {"pos": [], "uv": [], "idx": [], "mtl": None, "has_uv": True}
    mtl_tex = {}
    base = os.path.dirname(path)

    def flush():
        if cur["idx"]:
            pos = np.asarray(cur["pos"], dtype=f32).reshape(-1, 3)
            uv = np.asarray(cur["uv"], dtype=f32).reshape(-1, 2) if cur["has_uv"] else None
            meshes.append((pos, uv, np.asarray(cur["idx"], dtype=np.uint32), mtl_tex.get(cur["mtl"])))
        cur["pos"], cur["uv"], cur["idx"], cur["has_uv"] = [], [], [], True

    with open(path, "r") as fh:
        for line in fh:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                v.append([float(t[1]), float(t[2]), float(t[3])])
            elif t[0] == "vt":
                vt.append([float(t[1]), float(t[2]) if len(t) > 2 else 0.0])
            elif t[0] == "mtllib":
                mtl_tex.update(_read_mtl(os.path.join(base, " ".join(t[1:]))))
            elif t[0] in ("o", "g"):
                flush()
            elif t[0] == "usemtl":
                flush()
                cur["mtl"] = " ".join(t[1:])
            elif t[0] == "f":
                corners = []
                for c in t[1:]:
                    parts = c.split("/")
                    vi = int(parts[0])
                    vi = vi - 1 if vi > 0 else len(v) + vi
                    ti = None
                    if len(parts) > 1 and parts[1] != "":
                        ti = int(parts[1])
                        ti = ti - 1 if ti > 0 else len(vt) + ti
                    corners.append((vi, ti))
                b = len(cur["pos"])
                for vi, ti in corners:
                    cur["pos"].append(v[vi])
                    if ti is None:
                        cur["has_uv"] = False
                        cur["uv"].append([0.0, 0.0])
                    else:
                        cur["uv"].append([f32(vt[ti][0]), f32(1.0) - f32(vt[ti][1])])  # FlipUVs, in float
                for k in range(1, len(corners) - 1):
                    cur["idx"] += [b, b + k, b + k + 1]
    flush()
    if not meshes:
        raise SceneError("ERROR: Assimp: OBJ: no faces in \"%s\"" % path)
    return meshes


def _read_mtl(path):
    out, name = {}, None
    if not os.path.isfile(path):
        return out
    with open(path, "r") as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "newmtl":
                name = " ".join(t[1:])
            elif t[0] == "map_Kd" and name is not None:
                out[name] = line.split(None, 1)[1].strip()
    return out
