"""ctypes front-end of the parity checkers — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.

  Oracle()     liboracle.so — the CPU restatement (pt_oracle.c), always available
               after `make -C oracle` / __graft_entry__.build().
  Reference()  _ref/libref.so — the unmodified reference kernel compiled for
               x86-64 and linked with the IEEE-plain builtin stand-ins of
               ref_shim.cpp (build container only: .gpurunignore keeps it off the
               GPU box).  "Parity unpinned" in the sense of the build rules — see
               the header of pt_oracle.c.
  ReferenceGfx950()  _ref_gfx950/ — the same kernel file built for gfx950 with ROCm's REAL OpenCL
               builtin library and run on the GPU through the HIP module API (GPU box only;
               measurement tool of DESIGN.md §3, no pass/fail parity test depends on it).
All take the product's host-side scene object (anything with .desc() and
.texture_args(), i.e. opencl-raytracing_amd.scene.SceneCreator) so that oracle and
GPU are fed byte-identical inputs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref.so")
REF950_SO = os.path.join(HERE, "_ref_gfx950", "libref950.so")
REF950_HSACO = os.path.join(HERE, "_ref_gfx950", "ref950.hsaco")
REF950_HSACO_NOCONTRACT = os.path.join(HERE, "_ref_gfx950", "ref950_nocontract.hsaco")

COUNTER_FIELDS = ("samples", "bounces", "t_sphere", "t_plane", "t_lens", "t_model", "t_mesh", "t_tri", "h_tri",
                  "h_bounce", "n_scatter", "n_dielectric", "n_texfetch", "image_reads")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


def build(ref=True):
    """make -C oracle (liboracle.so, and _ref/ when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all" if ref else os.path.join(HERE, "liboracle.so")])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


class _Base:
    def _scene_args(self, scene):
        d = scene.desc()
        tex, tw, th, layers = scene.texture_args()
        self._keep = (d, scene)
        return C.byref(d), tex, tw, th, layers


class Oracle(_Base):
    def __init__(self, path=ORACLE_SO):
        if not os.path.isfile(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        self.lib.oracle_counters_bytes.restype = C.c_uint64

    def make_random_table(self, seed):
        out = np.empty(400000, dtype=np.float32)
        rc = self.lib.oracle_make_random_table(C.c_uint64(seed), _fp(out), C.c_size_t(out.size))
        assert rc == 0
        return out

    def render(self, scene, cam, table, w, h, mode, first=0, count=1, region=None, image=None, threads=1):
        """mode 0: `trace`; 1: one `retrace` of sample `first` over `image`; 2: trace + count-1
        retraces.  → (image h×w×4 float32 gamma space, Counters)."""
        x0, y0, cw, ch = region if region else (0, 0, w, h)
        img = np.zeros((h, w, 4), dtype=np.float32) if image is None else _f32(image).copy()
        cn = Counters()
        sd, tex, tw, th, layers = self._scene_args(scene)
        cam, table = _f32(cam), _f32(table)
        rc = self.lib.oracle_render(_fp(img), w, h, x0, y0, cw, ch, _fp(cam), _fp(table), sd, tex, tw, th, layers,
                                    mode, C.c_uint32(first), C.c_uint32(count), threads, C.byref(cn))
        assert rc == 0, "oracle_render failed"
        return img, cn

    def samples(self, scene, cam, table, w, h, xs, ys, ss):
        xs, ys, ss = _u32(xs), _u32(ys), _u32(ss)
        out = np.zeros((len(xs), 3), dtype=np.float32)
        cn = Counters()
        sd, tex, tw, th, layers = self._scene_args(scene)
        cam, table = _f32(cam), _f32(table)
        rc = self.lib.oracle_samples(w, h, _fp(cam), _fp(table), sd, tex, tw, th, layers, _fp(xs), _fp(ys), _fp(ss),
                                     C.c_size_t(len(xs)), _fp(out), C.byref(cn))
        assert rc == 0
        return out, cn

    def linear_sum(self, scene, cam, table, w, h, region, first, count):
        x0, y0, cw, ch = region
        out = np.zeros((ch, cw, 3), dtype=np.float64)
        sd, tex, tw, th, layers = self._scene_args(scene)
        cam, table = _f32(cam), _f32(table)
        rc = self.lib.oracle_linear_sum(_fp(out), w, h, x0, y0, cw, ch, _fp(cam), _fp(table), sd, tex, tw, th, layers,
                                        C.c_uint32(first), C.c_uint32(count))
        assert rc == 0
        return out

    def hit(self, kind, scene, rays, prim):
        rays, prim = _f32(rays), _u32(prim)
        out = np.zeros((len(prim), 12), dtype=np.float32)
        sd = self._scene_args(scene)[0]
        assert self.lib.oracle_hit(kind, sd, _fp(rays), _fp(prim), C.c_size_t(len(prim)), _fp(out)) == 0
        return out

    def hit_triangle(self, scene, rays, mesh, face):
        rays, mesh, face = _f32(rays), _u32(mesh), _u32(face)
        out = np.zeros((len(mesh), 12), dtype=np.float32)
        sd = self._scene_args(scene)[0]
        assert self.lib.oracle_hit_triangle(sd, _fp(rays), _fp(mesh), _fp(face), C.c_size_t(len(mesh)), _fp(out)) == 0
        return out

    def material(self, routine, scene, table, vec):
        """routine 0 rayReflect, 1 rayRefract, 2 rayScatter, 3 rayRefractDielectric on n × 16 input records
        (cases.material_vectors) → n × 9 floats {new origin, new dir, colour}."""
        vec, table = _f32(vec), _f32(table)
        out = np.zeros((len(vec), 9), dtype=np.float32)
        sd = self._scene_args(scene)[0]
        assert self.lib.oracle_material(routine, sd, _fp(table), _fp(vec), C.c_size_t(len(vec)), _fp(out)) == 0
        return out

    def counters_bytes(self, cn):
        return int(self.lib.oracle_counters_bytes(C.byref(cn)))


class Reference(_Base):
    """The compiled reference kernel (oracle/_ref/libref.so)."""

    @staticmethod
    def available():
        return os.path.isfile(REF_SO)

    def __init__(self, path=REF_SO):
        self.lib = C.CDLL(path)

    def trace(self, scene, cam, table, w, h, threads=1, region=None):
        x0, y0, cw, ch = region if region else (0, 0, w, h)
        img = np.zeros((h, w, 4), dtype=np.float32)
        sd, tex, tw, th, layers = self._scene_args(scene)
        cam, table = _f32(cam), _f32(table)
        assert self.lib.ref_trace(_fp(img), w, h, x0, y0, cw, ch, _fp(cam), _fp(table), sd, tex, tw, th, layers,
                                  threads) == 0
        return img

    def retrace(self, image, scene, cam, table, w, h, sample, threads=1, region=None):
        x0, y0, cw, ch = region if region else (0, 0, w, h)
        img = _f32(image).copy()
        sd, tex, tw, th, layers = self._scene_args(scene)
        cam, table = _f32(cam), _f32(table)
        assert self.lib.ref_retrace(_fp(img), w, h, x0, y0, cw, ch, _fp(cam), _fp(table), sd, tex, tw, th, layers,
                                    C.c_uint32(sample), threads) == 0
        return img

    def progressive(self, scene, cam, table, w, h, spp, threads=1, region=None):
        """trace + (spp-1) retrace launches — the reference's render()/renderAgain() sequence."""
        img = self.trace(scene, cam, table, w, h, threads, region)
        for s in range(1, spp):
            img = self.retrace(img, scene, cam, table, w, h, s, threads, region)
        return img

    def samples(self, scene, cam, table, w, h, xs, ys, ss):
        xs, ys, ss = _u32(xs), _u32(ys), _u32(ss)
        out = np.zeros((len(xs), 3), dtype=np.float32)
        sd, tex, tw, th, layers = self._scene_args(scene)
        cam, table = _f32(cam), _f32(table)
        assert self.lib.ref_samples(w, h, _fp(cam), _fp(table), sd, tex, tw, th, layers, _fp(xs), _fp(ys), _fp(ss),
                                    C.c_size_t(len(xs)), _fp(out)) == 0
        return out

    def hit(self, kind, scene, rays, prim):
        rays, prim = _f32(rays), _u32(prim)
        out = np.zeros((len(prim), 12), dtype=np.float32)
        sd = self._scene_args(scene)[0]
        assert self.lib.ref_hit(kind, sd, _fp(rays), _fp(prim), C.c_size_t(len(prim)), _fp(out)) == 0
        return out

    def hit_triangle(self, scene, rays, mesh, face):
        rays, mesh, face = _f32(rays), _u32(mesh), _u32(face)
        out = np.zeros((len(mesh), 12), dtype=np.float32)
        sd = self._scene_args(scene)[0]
        assert self.lib.ref_hit_triangle(sd, _fp(rays), _fp(mesh), _fp(face), C.c_size_t(len(mesh)), _fp(out)) == 0
        return out


def _ref_material(self, routine, scene, table, vec):
    vec, table = _f32(vec), _f32(table)
    out = np.zeros((len(vec), 9), dtype=np.float32)
    sd = self._scene_args(scene)[0]
    assert self.lib.ref_material(routine, sd, _fp(table), _fp(vec), C.c_size_t(len(vec)), _fp(out)) == 0
    return out


Reference.material = _ref_material


class ReferenceGfx950(_Base):
    """The reference's kernel file as ROCm's OpenCL tool chain builds it for gfx950 (real builtin
    library: fma-contracted dot, rsq-based normalize, rcp-based divide), executed on the GPU.
    Several code objects (the default build, the -ffp-contract=off build) can be open in one process."""

    @staticmethod
    def available(hsaco=REF950_HSACO):
        return os.path.isfile(REF950_SO) and os.path.isfile(hsaco)

    def __init__(self, hsaco=REF950_HSACO, device=0):
        self.lib = C.CDLL(REF950_SO)
        self.lib.ref950_last_error.restype = C.c_char_p
        self.handle = self.lib.ref950_open(hsaco.encode(), device)
        if self.handle < 0:
            raise RuntimeError(self.lib.ref950_last_error().decode())

    def render(self, scene, cam, table, w, h, first, count, want_last=False, grid=None):
        """→ (sum over samples first..first+count-1 of the LINEAR radiance, h×w×4 with the count in .w,
        and optionally the last sample's own frame).  grid = (gw, gh): only the frame's corner x < gw, y < gh is
        rendered (whole-frame coordinates; everything else stays 0)."""
        if scene.texture_args()[3]:
            raise ValueError("textured scenes need an OpenCL image object; not supported by this tool")
        acc = np.zeros((h, w, 4), dtype=np.float32)
        last = np.zeros((h, w, 4), dtype=np.float32) if want_last else None
        sd = self._scene_args(scene)[0]
        cam, table = _f32(cam), _f32(table)
        gw, gh = grid if grid else (0, 0)
        rc = self.lib.ref950_render(self.handle, sd, _fp(cam), _fp(table), w, h, C.c_uint32(first), C.c_uint32(count),
                                    int(gw), int(gh), _fp(acc), _fp(last) if want_last else None)
        if rc:
            raise RuntimeError(self.lib.ref950_last_error().decode())
        return (acc, last) if want_last else acc

    def builtin(self, op, vec):
        """One OpenCL builtin per record, evaluated by ROCm's OpenCL library: n × 8 floats → n × 4 floats."""
        vec = np.ascontiguousarray(vec, dtype=np.float32).reshape(-1, 8)
        out = np.zeros((len(vec), 4), dtype=np.float32)
        rc = self.lib.ref950_builtin(self.handle, int(op), _fp(vec), C.c_size_t(len(vec)), _fp(out))
        if rc:
            raise RuntimeError(self.lib.ref950_last_error().decode())
        return out
