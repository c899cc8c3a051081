"""The C++ host façade (host/: RayTracer / SceneCreator / Camera over the C ABI — the
reference's language) against the Python mirror.  CPU part: scene arrays and camera block
byte for byte (rt_cli --dump-scene needs no GPU).  GPU part: frames rendered by rt_cli."""
import os
import subprocess

import numpy as np
import pytest

import cases

rt = cases.rt
ROOT = cases.ROOT
CLI = os.path.join(ROOT, "host", "rt_cli")
ASSETS = os.path.join(ROOT, "assets")


def dump(scene, cam, size, tmp_path):
    out = str(tmp_path / "scene.bin")
    cmd = [CLI, "--scene", os.path.join(ASSETS, "scenes", scene), "--dump-scene", out,
           "--camera=%s" % ",".join(str(c) for c in cam), "--size", "%dx%d" % size]
    subprocess.run(cmd, check=True, cwd=ROOT)
    raw = open(out, "rb").read()
    counts = np.frombuffer(raw, np.uint32, 12)
    off = 48
    camblock = np.frombuffer(raw, np.float32, 12, off)
    off += 48
    a = rt._abi
    parts = {}
    for name, dt, n in (("materials", a.MATERIAL, counts[0]), ("spheres", a.SPHERE, counts[1]),
                        ("planes", a.PLANE, counts[2]), ("lenses", a.LENS, counts[3]),
                        ("vertices", np.dtype(("<f4", 4)), counts[4]), ("uvs", np.dtype(("<f4", 2)), counts[5]),
                        ("indices", np.dtype("<u4"), counts[6]), ("meshes", a.MESH, counts[7]),
                        ("models", a.MODEL, counts[8])):
        parts[name] = np.frombuffer(raw, dt, int(n), off)
        off += dt.itemsize * int(n)
    tex = None
    if counts[11]:
        tex = np.frombuffer(raw, np.float32, int(counts[9] * counts[10] * counts[11] * 4), off).reshape(
            counts[11], counts[10], counts[9], 4)
    return counts, camblock, parts, tex


@pytest.mark.parametrize("scene", ["c2_cornell.scene", "c3_cube.scene", "all_kinds.scene"])
def test_cpp_scene_arrays_match_python(built, scene, tmp_path):
    cam = (-8.0, -1.0, -8.0, 45.0, 0.0)
    counts, camblock, parts, tex = dump(scene, cam, (1200, 800), tmp_path)
    s = rt.SceneCreator()
    s.loadScene(os.path.join(ASSETS, "scenes", scene), base_dir=ASSETS)
    if len(s.models):
        s.loadTextures(search_dirs=[os.path.join(ASSETS, "textures")])
    d = s.desc()
    for name, arr in (("materials", s.materials), ("spheres", s.spheres), ("planes", s.planes),
                      ("lenses", s.lenses), ("vertices", s.vertices), ("uvs", s.texture_uv),
                      ("indices", s.indices), ("meshes", s.meshes), ("models", s.models)):
        assert parts[name].tobytes() == np.ascontiguousarray(arr).tobytes(), name
    assert d.material_count == counts[0] and d.model_count == counts[8]
    pycam = rt.Camera(60, np.float32(1200) / np.float32(800), cam[:3], cam[3], cam[4]).transferData()
    assert np.array_equal(camblock.view(np.uint32), pycam.view(np.uint32))
    if tex is not None:
        assert tex.shape == s.textures.shape
        # PNG via PIL (RGBA) vs PPM via the C++ reader: same 8-bit texels, same RGB^2.2 rule
        assert np.allclose(tex, s.textures, rtol=2e-7, atol=0)


def test_cpp_scene_errors(built, tmp_path):
    bad = tmp_path / "bad.scene"
    for text, msg in (("SPHERES:\n(0, 0, 3), 1, 12\n", "IMPROPER UNSIGNED INT"),
                      ("SPHERES:\n(0, 0, 3), 1e3, 0\n", "IMPROPER FLOAT"),
                      ("SPHERES:\n(0, 0), 1, 0\n", "IMPROPER VECTOR"),
                      ("MATERIALS:\nshiny, (1, 1, 1), 1\n", "DOES NOT EXIST"),
                      ("(0, 0, 3), 1, 0\n", "OPERATION NOT SPECIFIED")):
        bad.write_text(text)
        p = subprocess.run([CLI, "--scene", str(bad), "--dump-scene", str(tmp_path / "o.bin")], capture_output=True,
                           text=True)
        assert p.returncode != 0 and msg in p.stderr, (text, p.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,progressive", [("c2_cornell.scene", False), ("c2_cornell.scene", True),
                                               ("all_kinds.scene", False)])
def test_cli_frames_match_python_path(built, scene, progressive, tmp_path):
    w, h, spp = 320, 200, 8
    raw = str(tmp_path / "f.f32")
    cmd = [CLI, "--scene", os.path.join(ASSETS, "scenes", scene), "--size", "%dx%d" % (w, h), "--spp", str(spp),
           "--camera=-8,-1,-8,45,0", "--raw", raw, "--out", str(tmp_path / "f.tga"), "--pfm", str(tmp_path / "f.pfm")]
    if progressive:
        cmd.append("--progressive")
    subprocess.run(cmd, check=True, cwd=ROOT)
    got = np.fromfile(raw, np.float32).reshape(h, w, 4)
    s = rt.SceneCreator()
    s.loadScene(os.path.join(ASSETS, "scenes", scene), base_dir=ASSETS)
    if len(s.models):
        s.loadTextures(search_dirs=[os.path.join(ASSETS, "textures")])
    t = rt.RayTracer(w, h, scene=s)
    cam = rt.Camera(60, np.float32(w) / np.float32(h), (-8, -1, -8), 45.0, 0.0)
    if progressive:
        t.render(cam)
        for _ in range(spp - 1):
            t.renderAgain(cam)
        exp = t.transferImage()
    else:
        exp = t.renderFrame(cam, spp)
    t.close()
    if len(s.models):   # texels may differ in the last bit between the two decoders
        assert (np.abs(got - exp) <= 1e-4 * np.maximum(np.abs(exp), 1e-6) + 1e-6).mean() > 0.999
    else:
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    tga = open(str(tmp_path / "f.tga"), "rb").read()
    assert len(tga) == 18 + 3 * w * h and tga[2] == 2 and tga[16] == 24
    assert int.from_bytes(tga[12:14], "little") == w and int.from_bytes(tga[14:16], "little") == h
    pfm = open(str(tmp_path / "f.pfm"), "rb").read()
    assert pfm.startswith(b"PF\n%d %d\n-1.0\n" % (w, h))
