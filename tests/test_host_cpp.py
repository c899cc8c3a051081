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


@pytest.mark.parametrize("scene", ["c2_cornell.scene", "c3_cube.scene", "all_kinds.scene", "scene.scene", "c1_sphere.scene",
                                   "c5_mesh.scene"])
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
        # PNG decoded by PIL vs by the C++ reader (zlib + scanline filters): same 8-bit texels, and the same
        # stb_image rule (float)pow(byte / 255.0f, 2.2f) through libm's pow in both mirrors → same bits
        assert np.array_equal(tex.view(np.uint32), s.textures.view(np.uint32))


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "assets", "scenes", "scene.scene")),
                    reason="the reference tree exists in the build container only")
def test_reference_scene_tree_loads_unchanged(built, tmp_path):
    """The reference's own scene file, OBJ/MTL cubes and RGBA PNGs, read where they lie with the working
    directory at the reference's root (its `load:` paths are CWD-relative, src/scene.cpp:355 → :195; its .mtl
    files name an absolute path on the author's machine, assets/cube/cube.mtl:13): both mirrors must accept
    the tree as it is and produce byte-identical arrays.  Nothing is copied into the repo."""
    out = str(tmp_path / "ref.bin")
    subprocess.run([CLI, "--scene", "assets/scenes/scene.scene", "--dump-scene", out, "--size", "1200x800"],
                   check=True, cwd=REF)
    raw = open(out, "rb").read()
    counts = np.frombuffer(raw, np.uint32, 12)
    # materials, spheres, planes, lenses, vertices, uvs, indices, meshes, models, tex w, h, layers
    assert list(counts) == [9, 8, 1, 1, 48, 48, 72, 2, 2, 1024, 1024, 2]
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        s = rt.SceneCreator()
        s.loadScene("assets/scenes/scene.scene")
        s.loadTextures()
    finally:
        os.chdir(cwd)
    a = rt._abi
    off = 96
    for name, arr, dt in (("materials", s.materials, a.MATERIAL), ("spheres", s.spheres, a.SPHERE),
                          ("planes", s.planes, a.PLANE), ("lenses", s.lenses, a.LENS),
                          ("vertices", s.vertices, np.dtype(("<f4", 4))), ("uvs", s.texture_uv, np.dtype(("<f4", 2))),
                          ("indices", s.indices, np.dtype("<u4")), ("meshes", s.meshes, a.MESH),
                          ("models", s.models, a.MODEL)):
        n = len(arr) * dt.itemsize
        assert raw[off:off + n] == np.ascontiguousarray(arr).tobytes(), name
        off += n
    tex = np.frombuffer(raw, np.float32, 2 * 1024 * 1024 * 4, off).reshape(2, 1024, 1024, 4)
    assert np.array_equal(tex.view(np.uint32), s.textures.view(np.uint32))
    assert [os.path.basename(p) for p in s.texture_paths] == ["die.png", "die2.png"]
    # what the importer must yield for the cubes: one vertex per face corner, 12 triangles each
    assert list(s.meshes["face_count"]) == [12, 12] and list(s.meshes["texture_ID"]) == [0, 1]
    assert 0.0 <= float(s.textures.min()) and float(s.textures.max()) <= 1.0


def test_png_reader_rejects_non_rgba(built, tmp_path):
    """src/scene.cpp:165-179: a texture must have 4 channels."""
    from PIL import Image
    os.makedirs(tmp_path / "m")
    Image.fromarray(np.full((4, 4, 3), 128, np.uint8), "RGB").save(str(tmp_path / "m" / "t.png"))
    (tmp_path / "m" / "b.mtl").write_text("newmtl s\nmap_Kd t.png\n")
    (tmp_path / "m" / "b.obj").write_text("mtllib b.mtl\nusemtl s\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\n"
                                           "f 1/1 2/2 3/3\n")
    (tmp_path / "s.scene").write_text('MATERIALS:\ntextured, (1, 1, 1), 1\nMODELS:\nload: "m/b.obj", 0\n')
    p = subprocess.run([CLI, "--scene", str(tmp_path / "s.scene"), "--dump-scene", str(tmp_path / "o.bin")],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert p.returncode != 0 and "3 INSTEAD OF 4 (RGBA)" in p.stderr, p.stderr
    s = rt.SceneCreator()
    s.loadScene(str(tmp_path / "s.scene"))
    with pytest.raises(rt.SceneError, match="3 INSTEAD OF 4"):
        s.loadTextures()


def test_png_reader_all_filters(built, tmp_path):
    """The C++ PNG reader against PIL on an image that makes the encoder use every scanline filter."""
    from PIL import Image
    rng = np.random.RandomState(5)
    img = np.zeros((64, 48, 4), np.uint8)
    img[:16] = rng.randint(0, 256, (16, 48, 4))                                   # noise
    img[16:32] = np.linspace(0, 255, 48).astype(np.uint8)[None, :, None]          # horizontal ramp (Sub)
    img[32:48] = np.linspace(0, 255, 16).astype(np.uint8)[:, None, None]          # vertical ramp (Up)
    img[48:] = (np.add.outer(np.arange(16), np.arange(48)) * 3 % 256).astype(np.uint8)[..., None]  # diagonal (Paeth/Avg)
    os.makedirs(tmp_path / "m")
    Image.fromarray(img, "RGBA").save(str(tmp_path / "m" / "t.png"), optimize=True)
    (tmp_path / "m" / "b.mtl").write_text("newmtl s\nmap_Kd t.png\n")
    (tmp_path / "m" / "b.obj").write_text("mtllib b.mtl\nusemtl s\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\n"
                                           "f 1/1 2/2 3/3\n")
    (tmp_path / "s.scene").write_text('MATERIALS:\ntextured, (1, 1, 1), 1\nMODELS:\nload: "m/b.obj", 0\n')
    out = str(tmp_path / "o.bin")
    subprocess.run([CLI, "--scene", str(tmp_path / "s.scene"), "--dump-scene", out], check=True, cwd=str(tmp_path))
    raw = open(out, "rb").read()
    counts = np.frombuffer(raw, np.uint32, 12)
    assert list(counts[9:12]) == [48, 64, 1]
    tex = np.frombuffer(raw, np.float32, 48 * 64 * 4, len(raw) - 48 * 64 * 16).reshape(64, 48, 4)
    s = rt.SceneCreator()
    s.loadScene(str(tmp_path / "s.scene"))
    s.loadTextures()
    assert np.array_equal(tex.view(np.uint32), s.textures[0].view(np.uint32))
    # the rule itself: colour = (float)pow(byte / 255.0f, 2.2f), alpha = byte / 255.0f
    q = (img.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    assert np.array_equal(tex[..., 3], q[..., 3])
    exp = np.power(q[..., :3].astype(np.float64), float(np.float32(2.2))).astype(np.float32)
    assert np.abs(tex[..., :3].view(np.int32) - exp.view(np.int32)).max() <= 1


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
