"""CPU: host-side logic — scene file grammar, OBJ reader, camera block, C ABI surface
(symbols, struct layouts, error behaviour without a device)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cases

rt = cases.rt
ROOT = cases.ROOT


def test_abi_exports_every_declared_symbol(built):
    """librt_amd.so loads and exports every function include/rt_amd.h declares."""
    hdr = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    declared = set(re.findall(r"^(?:int|void|uint64_t|const char \*)\s*\*?\s*(rt_[a-z_0-9]+)\(", hdr, re.M))
    assert len(declared) >= 30
    assert declared == set(rt.raytracer.SYMBOLS), declared ^ set(rt.raytracer.SYMBOLS)
    lib = C.CDLL(rt.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.rt_abi_version() == 3


def test_header_is_plain_c_and_a_c_program_links(built, tmp_path):
    """include/rt_amd.h is a C99 header (no C++, no torch types) and a C caller links against the library."""
    import subprocess
    src = tmp_path / "caller.c"
    src.write_text('#include <stdio.h>\n#include "rt_amd.h"\n'
                   'int main(void) {\n'
                   '    rt_context *ctx = NULL;\n'
                   '    rt_scene_desc d = {0};\n'
                   '    (void)d;\n'
                   '    printf("%d %zu %zu %zu\\n", rt_abi_version(), sizeof(rt_sphere), sizeof(rt_material), sizeof(rt_counters));\n'
                   '    return rt_render(ctx, NULL) == RT_OK;   /* a NULL context is an error, not a crash */\n'
                   '}\n')
    exe = tmp_path / "caller"
    pkg = os.path.dirname(rt.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe), "-L", pkg, "-lrt_amd", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["3", "32", "48", "112"]


def test_struct_layouts_match_reference_device_structs():
    a = rt._abi
    assert (a.MATERIAL.itemsize, a.SPHERE.itemsize, a.PLANE.itemsize, a.LENS.itemsize, a.MESH.itemsize,
            a.MODEL.itemsize) == (48, 32, 48, 64, 16, 12)        # SURVEY §8a, verified against the compiled .cl
    assert a.MATERIAL.fields["color"][1] == 16 and a.MATERIAL.fields["extra_data"][1] == 32
    assert a.SPHERE.fields["r"][1] == 16 and a.SPHERE.fields["mat_ID"][1] == 20
    assert a.PLANE.fields["normal"][1] == 16 and a.PLANE.fields["mat_ID"][1] == 32
    assert a.LENS.fields["p1"][1] == 16 and a.LENS.fields["r1"][1] == 48 and a.LENS.fields["mat_ID"][1] == 56
    assert C.sizeof(a.SceneDesc) == 9 * 8 + 10 * 4
    assert C.sizeof(a.Counters) == 14 * 8


def test_no_device_means_failure_not_fallback(built):
    """Without a GPU the product fails loudly: there is no CPU path behind the C ABI."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = rt.load_library()
    ctx = C.c_void_p()
    assert lib.rt_create(0, 64, 64, C.byref(ctx)) == -2
    assert b"no HIP device" in lib.rt_last_error(None) and not ctx.value
    with pytest.raises(rt.RtError):
        rt.RayTracer(64, 64)


def test_product_never_touches_the_oracle():
    """No file of the product (package, include/, host/, bench's GPU leg) references oracle/."""
    pkg = os.path.join(ROOT, "opencl-raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "pt_oracle" not in text and "import oracle" not in text and \
                    "from oracle" not in text, os.path.join(dirpath, f)


SCENE_TEXT = """
# comment line
MATERIALS:
diffuse, (1, 0.5, .25), 1   # trailing comment
light, (1, 1, 1), 0
SPHERES:
(0, -1.5, +3), 1.25, 0
PLANES:
(0, 5, 0), (0, 1, 0), 1
LENSES:
(5, 0, 0), (1, 0, 0), 10, 10, 2, 0
"""


def test_scene_grammar():
    s = rt.SceneCreator()
    s.loadSceneText(SCENE_TEXT)
    assert len(s.materials) == 2 and len(s.spheres) == 1 and len(s.planes) == 1 and len(s.lenses) == 1
    assert s.materials["type"].tolist() == [rt._abi.T_DIFFUSE, rt._abi.T_LIGHT]
    assert np.allclose(s.materials["color"][0], [1, 0.5, 0.25, 0])
    assert np.allclose(s.spheres["pos"][0], [0, -1.5, 3, 0]) and s.spheres["r"][0] == np.float32(1.25)
    # addLens: p1,p2 = pos ± normal·sqrt(r² − h²)  (src/scene.cpp:122-143)
    t = np.float32(np.sqrt(np.float32(100 - 4)))
    assert np.allclose(s.lenses["p1"][0][:3], [5 + t, 0, 0]) and np.allclose(s.lenses["p2"][0][:3], [5 - t, 0, 0])
    d = s.desc()
    assert (d.material_count, d.sphere_count, d.plane_count, d.lens_count, d.model_count) == (2, 1, 1, 1, 0)


@pytest.mark.parametrize("text,msg", [
    ("SPHERES:\n(0, 0, 3), 1, 12\n", "IMPROPER UNSIGNED INT"),            # getUInt takes ONE digit (scene.cpp:455)
    ("SPHERES:\n(0, 0, 3), 1e3, 0\n", "IMPROPER FLOAT"),                  # no exponents (scene.cpp:448)
    ("SPHERES:\n(0, 0), 1, 0\n", "IMPROPER VECTOR"),
    ("SPHERES:\n(0, 0, 3), 1\n", "NOT ENOUGH PARAMETERS"),
    ("MATERIALS:\nshiny, (1, 1, 1), 1\n", "MATERIAL:  shiny DOES NOT EXIST".replace("  ", " ")),
    ("MATERIALS:\n# nothing\nSPHERES:\n", None),
    ("SPHERES:\nfoo: 1\n", "OPERATION foo DOES NOT EXIST"),
    ("LENSES:\n(0, 0, 0), (1, 0, 0), 1, 1, 2, 0\n", "LENS"),
])
def test_scene_errors(text, msg):
    if msg is None:
        rt.SceneCreator().loadSceneText(text)       # empty sections are fine
        return
    with pytest.raises(rt.SceneError) as e:
        rt.SceneCreator().loadSceneText("MATERIALS:\ndiffuse, (1, 1, 1), 1\n" + text if "MATERIALS" not in text
                                        else text)
    assert msg in str(e.value)


def test_scene_line_before_any_section():
    with pytest.raises(rt.SceneError) as e:
        rt.SceneCreator().loadSceneText("(0, 0, 3), 1, 0\n")
    assert "OPERATION NOT SPECIFIED" in str(e.value)


def test_model_ops_and_obj_reader():
    """translate/rotate/scale post-multiply (glm), reset after load; one vertex per face corner,
    fan triangulation, v → 1−v (what Assimp's Triangulate|FlipUVs produces, src/scene.cpp:195)."""
    s = rt.SceneCreator()
    s.loadScene(os.path.join(ROOT, "assets", "scenes", "all_kinds.scene"), base_dir=os.path.join(ROOT, "assets"))
    assert len(s.models) == 2 and len(s.meshes) == 2
    assert s.models["mesh_anchor"].tolist() == [0, 1] and s.models["mat_ID"].tolist() == [2, 2]
    assert s.meshes["face_count"].tolist() == [12, 12]
    assert s.meshes["vertex_anchor"].tolist() == [0, 24] and s.meshes["index_anchor"].tolist() == [0, 36]
    assert len(s.vertices) == 48 and len(s.texture_uv) == 48 and len(s.indices) == 72
    assert s.indices[:6].tolist() == [0, 1, 2, 0, 2, 3]
    assert s.texture_paths == ["textures/checker_a.png", "textures/checker_b.png"]
    assert s.meshes["texture_ID"].tolist() == [0, 1]
    # first box: rotate 40° about y then scale 1.1 → every vertex at distance 1.1·sqrt(3) from the origin
    r = np.linalg.norm(s.vertices[:24, :3].astype(np.float64), axis=1)
    assert np.allclose(r, 1.1 * np.sqrt(3), atol=1e-5)
    # second box: translate then rotate → centred on the translation
    assert np.allclose(s.vertices[24:, :3].mean(0), [-6.2, 0.6, 0.2], atol=1e-5)
    # all faces counter-clockwise seen from outside (hitMeshOut's assumption, raytracer.cl:284)
    for m in range(2):
        v = s.vertices[s.meshes["vertex_anchor"][m]:][:24, :3].astype(np.float64)
        c = v.mean(0)
        tri = v[s.indices[36 * m:36 * m + 36]].reshape(12, 3, 3)
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        assert (np.einsum("ij,ij->i", n, tri.mean(1) - c) > 0).all()
    assert s.texture_uv.min() >= 0 and s.texture_uv.max() <= 1


def test_camera_block():
    cam = rt.Camera(60, 1.5)
    d = cam.transferData()
    assert d.dtype == np.float32 and d.shape == (12,)
    hh = np.tan(np.pi / 6)
    # yaw = pitch = 0: w = +z, u = +x, v = −y, so the lower-left corner has y = +hh (SURVEY §8a "image")
    assert np.allclose(d[0:3], 0) and np.allclose(d[3:6], [-1.5 * hh, hh, 1], atol=1e-6)
    assert np.allclose(d[6:9], [3 * hh, 0, 0], atol=1e-6) and np.allclose(d[9:12], [0, -2 * hh, 0], atol=1e-6)
    cam.setFasterSpeed(True)
    cam.move(rt.camera.FORWARD, 2.0)
    assert np.allclose(cam.transferData()[0:3], [0, 0, 10])
    cam.rotate(10, 500)                                   # pitch is clamped to ±89°
    assert cam.pitch == np.float32(89.0)
    cam.zoom(1000)                                        # fov clamped to [10, 90]
    assert cam.fov == np.float32(90.0)
    cam.setSize(2.0)
    assert np.isclose(cam.half_width, 2.0 * cam.half_height)


def test_workload_shapes():
    w = rt.workloads
    assert len(w.get("c2").scene.spheres) == 8 and len(w.get("c2").scene.planes) == 1
    c3 = w.get("c3", tex_size=16)
    assert c3.scene.meshes["face_count"].tolist() == [12] and len(c3.scene.spheres) == 4
    c4 = w.get("c4")
    assert len(c4.scene.spheres) == 100000 and c4.scene.spheres.nbytes == 3200000
    c5 = w.get("c5")
    assert int(c5.scene.meshes["face_count"][0]) == 50000 and (c5.width, c5.height, c5.spp) == (3840, 2160, 512)
    assert [w.get(n).spp for n in ("c1", "c2", "c3", "c4")] == [1, 64, 256, 64]


def test_default_scene_is_the_references_path_and_the_repositorys_own_file():
    """RayTracer(w, h, kernel_path) loads "assets/scenes/scene.scene" like the reference (src/raytracer.cpp:95) — the
    repository ships its own file of that name (spheres, plane, lens, two textured boxes, ten materials)."""
    assert rt.RayTracer.DEFAULT_SCENE == os.path.join("assets", "scenes", "scene.scene")
    assert 'return "assets/scenes/scene.scene"' in open(os.path.join(ROOT, "host", "raytracer.h")).read()
    s = rt.SceneCreator()
    s.loadScene(rt.RayTracer.REPO_DEFAULT_SCENE)
    s.loadTextures()
    assert (len(s.materials), len(s.spheres), len(s.planes), len(s.lenses), len(s.models)) == (10, 10, 1, 1, 2)
    assert s.textures.shape[0] == 2 and list(s.meshes["face_count"]) == [12, 12]
    ref = "/root/reference/assets/scenes/scene.scene"
    if os.path.isfile(ref):     # authored here, not the reference's file
        assert open(ref).read() != open(rt.RayTracer.REPO_DEFAULT_SCENE).read()


def test_c1_and_c5_come_from_scene_and_obj_files():
    """BASELINE.json: config 1 is an "assets/scenes single-sphere + plane", config 5 a "50k-triangle OBJ mesh": both
    workloads are read from files (assets/scenes/c1_sphere.scene; assets/scenes/c5_mesh.scene → assets/models/
    c5_sphere.obj through the OBJ reader, src/scene.cpp:192-295) and equal the generated arrays: positions and
    indices bit for bit; the uv array to its last bit but one (FlipUVs computes 1 - v in binary32, which cannot
    reproduce every v < 0.5 — C5's material is dielectric, no texel is ever fetched)."""
    a = rt._abi
    c1 = rt.workloads.get("c1").scene
    assert len(c1.spheres) == 1 and len(c1.planes) == 1 and list(c1.materials["type"]) == [a.T_DIFFUSE, a.T_LIGHT]
    assert tuple(c1.spheres["pos"][0][:3]) == (0.0, 0.0, 3.0) and c1.spheres["r"][0] == 1.0
    c5 = rt.workloads.get("c5").scene
    pos, uv, idx = rt.workloads.uv_sphere(200, 126, radius=2.5, centre=(0.0, 2.0, 0.0))
    assert len(idx) == 150000 and list(c5.meshes["face_count"]) == [50000]
    assert np.array_equal(c5.vertices[:, :3].view(np.uint32), pos.view(np.uint32))
    assert np.array_equal(c5.indices, idx)
    assert np.abs(c5.texture_uv - uv).max() <= 2.0 ** -25
    # the committed OBJ is what the generator writes
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        rt.workloads.write_c5_obj(os.path.join(d, "x.obj"))
        assert open(os.path.join(d, "x.obj")).read() == open(rt.workloads.C5_OBJ).read()


@pytest.mark.skipif(not os.path.isfile("/opt/rocm/lib/llvm/bin/clang"), reason="needs ROCm's clang (OpenCL front end + device libraries)")
def test_rocm_opencl_policy_compiles_to_the_opencl_builds_float_sequences():
    """Offline early warning (the bit-for-bit check runs on the GPU, tests/test_gpu_ref950.py): every OpenCL builtin the
    reference calls, compiled by ROCm's OpenCL tool chain, has the same floating-point opcode multiset as its
    restatement in csrc/pt_arith.hpp (policy 1) compiled by hipcc with the policy's flags — tools/arith_probe.py."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "arith_probe.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("| ok |") == 11
