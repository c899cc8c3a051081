"""GPU (-m gpu): size-independent properties at BASELINE.json's full sizes, edge cases
and error behaviour of the C ABI."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
rt = cases.rt


@pytest.fixture(scope="module")
def c2_full():
    wl = cases.workload("c2")
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=cases.SEED)
    yield wl, t
    t.close()


def test_c2_full_size_deterministic_and_probes(c2_full, oracle, table):
    wl, t = c2_full
    a = t.renderFrame(wl.camera, wl.spp)
    b = t.renderFrame(wl.camera, wl.spp)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))          # run-to-run bit-identical
    assert (a[..., 3] == 1.0).all() and np.isfinite(a).all()
    assert 0.05 < (a[..., :3].sum(-1) > 0).mean() < 1.0
    rng = np.random.RandomState(3)
    n = 6000
    xs, ys, ss = rng.randint(0, wl.width, n), rng.randint(0, wl.height, n), rng.randint(0, wl.spp, n)
    got = t.traceSamples(wl.camera, xs, ys, ss)
    exp, _ = oracle.samples(wl.scene, wl.camera, table, wl.width, wl.height, xs, ys, ss)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_c2_full_size_sharded_sum_equals_unsharded(c2_full):
    """Tile sharding (what the N-GPU path does) on one GPU with virtual ranks: the
    element-wise sum of the ranks' buffers is the unsharded frame, bit for bit."""
    wl, t = c2_full
    t.setShard(0, 1, 8, 8)
    whole_img = t.renderFrame(wl.camera, 16)
    whole_lin = t.readLinear()
    for world in (2, 8):
        acc_img = np.zeros_like(whole_img)
        acc_lin = np.zeros_like(whole_lin)
        owned = np.zeros(whole_img.shape[:2], np.int32)
        for rank in range(world):
            t.setShard(rank, world, 8, 8)
            img = t.renderFrame(wl.camera, 16)
            owned += (img[..., 3] > 0)
            acc_img += img
            acc_lin += t.readLinear()
        assert (owned == 1).all()                       # every pixel owned exactly once
        assert np.array_equal(acc_img.view(np.uint32), whole_img.view(np.uint32))
        assert np.array_equal(acc_lin.view(np.uint32), whole_lin.view(np.uint32))
    t.setShard(0, 1, 8, 8)


def test_c2_sample_ranges_compose(c2_full):
    """Linearity: samples [0,32) then [32,64) accumulate to the same mean as [0,64) (within
    float summation order), and a second accumulation of the same range doubles the sum."""
    wl, t = c2_full
    t.clear(); t.renderSamples(wl.camera, 0, 64); one = t.readLinear()
    t.clear(); t.renderSamples(wl.camera, 0, 32); t.renderSamples(wl.camera, 32, 32); two = t.readLinear()
    dev = np.abs(one - two) / np.maximum(np.maximum(np.abs(one), np.abs(two)), 1e-6)
    assert dev.max() <= 1e-5
    t.clear(); t.renderSamples(wl.camera, 0, 64); t.renderSamples(wl.camera, 0, 64); twice = t.readLinear()
    assert np.array_equal(twice.view(np.uint32), one.view(np.uint32))    # (2·sum)/(2·n) == sum/n exactly


def test_c3_full_size_256spp(oracle, table):
    wl = cases.workload("c3")
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=cases.SEED)
    img = t.renderFrame(wl.camera, wl.spp)
    assert np.isfinite(img).all() and (img[..., :3] <= 1.0 + 1e-6).all()
    # a 32×24 window through the textured cube against the oracle's progressive result
    region = (940, 310, 32, 24)
    exp, _ = oracle.render(wl.scene, wl.camera, table, wl.width, wl.height, 2, count=wl.spp, region=region, threads=8)
    x0, y0, cw, ch = region
    a, b = img[y0:y0 + ch, x0:x0 + cw], exp[y0:y0 + ch, x0:x0 + cw]
    assert (np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)).max() <= 1e-4
    t.close()


def test_edge_cases(oracle, table):
    s = rt.SceneCreator()                                   # empty scene: everything misses → black
    s.addMaterial(rt._abi.T_DIFFUSE, (1, 1, 1), 1)
    t = rt.RayTracer(33, 17, scene=s)
    cam = rt.Camera(60, 33 / 17).transferData()
    img = t.renderFrame(cam, 3)
    assert (img[..., :3] == 0).all() and (img[..., 3] == 1).all()
    t.resize(1, 1)                                          # 1×1 frame
    t.render(cam); t.renderAgain(cam)
    assert t.transferImage().shape == (1, 1, 4)
    t.close()
    wl = rt.workloads.get("all_kinds", width=40, height=24)
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=cases.SEED)
    # largest sample index the API admits
    xs, ys = np.arange(40) % 40, np.arange(40) % 24
    ss = np.full(40, 65535)
    got = t.traceSamples(wl.camera, xs, ys, ss)
    exp, _ = oracle.samples(wl.scene, wl.camera, table, 40, 24, xs, ys, ss)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    for spp in (1, 2, 63, 65, 130, 600):                    # non-power-of-two counts; > 512 leaves the sample queue
        img = t.renderFrame(wl.camera, spp)
        ref, _ = oracle.render(wl.scene, wl.camera, table, 40, 24, 2, count=spp, threads=8)
        assert (np.abs(img - ref) / np.maximum(np.maximum(np.abs(img), np.abs(ref)), 1e-6)).max() <= 1e-4, spp
    # another seed really changes the picture, and an injected table is honoured
    base = t.renderFrame(wl.camera, 4)
    t.setSeed(99)
    other = t.renderFrame(wl.camera, 4)
    assert not np.array_equal(base, other)
    t.setRandomTable(table)
    assert np.array_equal(t.renderFrame(wl.camera, 4).view(np.uint32), base.view(np.uint32))
    t.close()


def test_error_behaviour():
    lib = rt.load_library()
    ctx = C.c_void_p()
    assert lib.rt_create(0, 0, 10, C.byref(ctx)) == -1 and b"frame size" in lib.rt_last_error(None)
    assert lib.rt_create(99, 8, 8, C.byref(ctx)) == -2
    s = rt.SceneCreator()
    s.addMaterial(rt._abi.T_DIFFUSE, (1, 1, 1), 1)
    s.addSphere((0, 0, 3), 1, 5)                            # material 5 does not exist
    with pytest.raises(rt.RtError) as e:
        rt.RayTracer(8, 8, scene=s)
    assert e.value.code == -5 and "material 5" in str(e.value)
    wl = rt.workloads.get("c3", width=16, height=16, tex_size=8)
    wl.scene.textures = None                                # textured model without textures
    t = rt.RayTracer(16, 16, scene=wl.scene)
    with pytest.raises(rt.RtError) as e:
        t.render(wl.camera)
    assert e.value.code == -5
    with pytest.raises(rt.RtError):
        t.setShard(2, 2)                                    # rank out of range
    with pytest.raises(rt.RtError):
        t.setShard(0, 1, 6, 8)                              # tile side not a power of two
    with pytest.raises(rt.RtError):
        t.renderSamples(wl.camera, 65530, 10)               # beyond RT_MAX_SAMPLE
    t.close()
    assert rt.RtError  # render before any scene
    ctx = C.c_void_p()
    assert lib.rt_create(0, 8, 8, C.byref(ctx)) == 0
    cam = (C.c_float * 12)()
    assert lib.rt_render(ctx, cam) == -4 and b"no scene" in lib.rt_last_error(ctx)
    lib.rt_destroy(ctx)


@pytest.mark.parametrize("name,kw,spps", [
    ("all_kinds", dict(width=160, height=96), (1, 3, 64, 100, 128)),
    ("c2", dict(width=320, height=180), (64,)),
    ("c3", dict(width=160, height=90, tex_size=64), (16, 256)),
])
def test_prefix_sharing_and_sample_queue_change_no_bit(name, kw, spps):
    """The fused path (shared per-pixel prefix, in-wave sample queue) against the same path
    without the queue and against the direct path that traces every sample from the camera:
    identical accumulators and counters."""
    wl = rt.workloads.get(name, **kw)
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=cases.SEED)
    for spp in spps:
        out = []
        for share, queue in ((1, 1), (1, 0), (0, 0)):
            t.setOption(t.OPT_PREFIX_SHARING, share)
            t.setOption(t.OPT_SAMPLE_QUEUE, queue)
            t.enableCounters(True)
            t.resetCounters()
            t.clear()
            t.renderSamples(wl.camera, 5, spp)
            out.append((t.readLinear(), t.counters().as_dict()))
            t.enableCounters(False)
        for o in out[1:]:
            assert np.array_equal(out[0][0].view(np.uint32), o[0].view(np.uint32)), spp
            assert out[0][1] == o[1], spp
    t.close()


def test_pack_unpack_exchange_with_virtual_ranks():
    """The compact exchange of the N-GPU path on one GPU: every virtual rank renders its tiles and
    packs them (rt_pack_accum); a receiver context unpacks all of them (rt_unpack_accum) and must
    hold the unsharded accumulator bit for bit.  The packed layout equals the host restatement."""
    import torch
    from importlib import import_module
    d = import_module("opencl-raytracing_amd.distributed")
    wl = rt.workloads.get("all_kinds", width=173, height=99)
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=cases.SEED)
    t.clear(); t.renderSamples(wl.camera, 0, 8); whole = t.readLinear()
    world = 3
    recv = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=cases.SEED)
    recv.clear()
    n = t.shardSlots(world)
    assert n == d.shard_slots(wl.width, wl.height, world)
    acc_t = torch.as_tensor(t.deviceAccum(), device="cuda")
    for r in range(world):
        t.setShard(r, world, 8, 8)
        t.clear(); t.renderSamples(wl.camera, 0, 8)
        packed = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        t.packAccum(packed.data_ptr(), packed.numel() * 4)
        t.sync()
        valid, y, x = d.slot_pixels(wl.width, wl.height, r, world)
        host = acc_t.cpu().numpy()[y, x]
        host[~valid] = 0
        assert np.array_equal(packed.cpu().numpy().view(np.uint32), host.view(np.uint32))
        recv.unpackAccum(packed.data_ptr(), packed.numel() * 4, r, world)
        recv.sync()
    assert np.array_equal(recv.readLinear().view(np.uint32), whole.view(np.uint32))
    t.close(); recv.close()


@pytest.mark.parametrize("name,kw,spp", [
    ("c4", dict(width=160, height=90, n_spheres=20000), 4),
    ("c4", dict(width=64, height=36, n_spheres=100000), 2),
    ("c2", dict(width=160, height=90), 16),
    ("all_kinds", dict(width=160, height=96), 8),
])
def test_sphere_bvh_changes_no_bit(name, kw, spp):
    """RT_OPT_ACCEL: the conservative sphere BVH against the reference's brute-force loop —
    identical accumulators, probes and counters (the counters price the reference's logical
    work, so they do not depend on the search structure)."""
    wl = rt.workloads.get(name, **kw)
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=cases.SEED)
    rng = np.random.RandomState(11)
    n = 4000
    xs, ys, ss = rng.randint(0, wl.width, n), rng.randint(0, wl.height, n), rng.randint(0, 4096, n)
    out = []
    for accel in (0, 2):
        t.setOption(t.OPT_ACCEL, accel)
        t.enableCounters(True)
        t.resetCounters()
        t.clear()
        t.renderSamples(wl.camera, 0, spp)
        lin = t.readLinear()
        cn = t.counters().as_dict()
        t.enableCounters(False)
        t.render(wl.camera)
        t.renderAgain(wl.camera)
        out.append((lin, cn, t.transferImage(), t.traceSamples(wl.camera, xs, ys, ss)))
    for k in (0, 2, 3):
        assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), k
    assert out[0][1] == out[1][1]
    t.close()


def test_sphere_bvh_adversarial_scene(oracle, table):
    """Spheres that stress the conservative culling: tiny and huge radii, concentric and touching
    spheres, duplicates (index tie-break), centres far from the camera, glass (non-unit refracted
    directions).  BVH result == brute force == oracle probes."""
    s = rt.SceneCreator()
    s.addMaterial(rt._abi.T_DIFFUSE, (0.9, 0.9, 0.9), 1)
    s.addMaterial(rt._abi.T_REFRACTIVE, (1, 1, 1), 1.5)
    s.addMaterial(rt._abi.T_REFLECTIVE, (1, 1, 1), 0.9)
    s.addMaterial(rt._abi.T_LIGHT, (1, 1, 1), 0)
    s.addMaterial(rt._abi.T_DIELECTRIC, (1, 1, 1), 1.3)
    u = rt.workloads.uniforms(3000, 5)
    pos = np.stack([u[:, 0] * 40 - 20, u[:, 1] * 8 - 4, u[:, 2] * 40 - 10], 1).astype(np.float32)
    rad = (0.02 + 0.6 * u[:, 3] ** 3).astype(np.float32)
    mat = (np.arange(3000) % 5).astype(np.uint32)
    s.addSpheres(pos, rad, mat)
    s.addSpheres(pos[:200], rad[:200], (mat[:200] + 1) % 5)                 # exact duplicates: lower index must win
    s.addSpheres(pos[200:400], rad[200:400] * np.float32(1.5), mat[200:400])  # concentric shells
    s.addSphere((0, -500, 0), 400, 3)                                        # huge light
    s.addSphere((900, 0, 900), 50, 0)                                        # far away
    s.addSphere((0, 0, 5), 1e-4, 2)                                          # below MIN_DISTANCE scale
    s.addPlane((0, 5, 0), (0, 1, 0), 0)
    w, h = 192, 108
    cam = rt.Camera(70, w / h, (0.3, -1.0, -14.0), 3.0, 4.0).transferData()
    t = rt.RayTracer(w, h, scene=s, seed=cases.SEED)
    frames = []
    for accel in (0, 2):
        t.setOption(t.OPT_ACCEL, accel)
        t.clear(); t.renderSamples(cam, 0, 16)
        frames.append(t.readLinear())
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    rng = np.random.RandomState(5)
    n = 3000
    xs, ys, ss = rng.randint(0, w, n), rng.randint(0, h, n), rng.randint(0, 500, n)
    got = t.traceSamples(cam, xs, ys, ss)
    exp, _ = oracle.samples(s, cam, table, w, h, xs, ys, ss)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    t.close()


def _mesh_scene(segments, rings, camera_inside=False):
    """Two uv-sphere meshes (one dielectric, one textured, in one model each) that overlap, a mirror
    sphere, a plane and a light: rays enter, leave, graze and miss meshes of 2·segments·(rings−1) faces."""
    s = rt.SceneCreator()
    s.addMaterial(rt._abi.T_DIELECTRIC, (1, 1, 1), 1.3)   # 0
    s.addMaterial(rt._abi.T_DIFFUSE, (0.8, 0.8, 0.8), 1)  # 1
    s.addMaterial(rt._abi.T_LIGHT, (1, 1, 1), 0)          # 2
    s.addMaterial(rt._abi.T_TEXTURED, (1, 1, 1), 1)       # 3
    s.addMaterial(rt._abi.T_REFLECTIVE, (1, 1, 1), 0.9)   # 4
    pos, uv, idx = rt.workloads.uv_sphere(segments, rings, radius=2.5, centre=(0.0, 2.0, 0.0))
    s.addMesh(pos, uv, idx)
    s.addModel(1, 0)
    pos, uv, idx = rt.workloads.uv_sphere(segments // 2 + 3, rings // 2 + 3, radius=1.7, centre=(2.6, 2.6, 0.8))
    s.addMesh(pos, uv, idx, texture_ID=0)
    s.addModel(1, 3)
    s.setTextures(rt.workloads.checker_texture(32, 4))
    s.addSphere((1, -200, 0), 100, 2)
    s.addSphere((-4.5, 3.5, 1.0), 1.5, 4)
    s.addPlane((0, 5, 0), (0, 1, 0), 1)
    cam = rt.Camera(60, 16 / 9, (0.2, 2.1, 0.3) if camera_inside else (-7, 0, -7), 45.0, 8.0).transferData()
    return s, cam


@pytest.mark.parametrize("segments,rings,inside", [(24, 16, False), (100, 60, False), (60, 40, True)])
def test_mesh_bvh_changes_no_bit(segments, rings, inside, oracle, table):
    """RT_OPT_ACCEL for meshes: the conservative per-mesh BVH (first front-facing hit in face order)
    against the reference's face-by-face scan — identical accumulators, compat images, probes and
    counters; and the probes against the oracle."""
    s, cam = _mesh_scene(segments, rings, inside)
    w, h = 128, 72
    t = rt.RayTracer(w, h, scene=s, seed=cases.SEED)
    rng = np.random.RandomState(17)
    n = 3000
    xs, ys, ss = rng.randint(0, w, n), rng.randint(0, h, n), rng.randint(0, 2000, n)
    out = []
    for accel in (0, 1):
        t.setOption(t.OPT_ACCEL, accel)
        t.enableCounters(True)
        t.resetCounters()
        t.clear()
        t.renderSamples(cam, 0, 8)
        lin = t.readLinear()
        cn = t.counters().as_dict()
        t.enableCounters(False)
        t.render(cam)
        t.renderAgain(cam)
        out.append((lin, cn, t.transferImage(), t.traceSamples(cam, xs, ys, ss)))
    for k in (0, 2, 3):
        assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), k
    assert out[0][1] == out[1][1]
    exp, _ = oracle.samples(s, cam, table, w, h, xs[:600], ys[:600], ss[:600])
    assert np.array_equal(out[1][3][:600].view(np.uint32), exp.view(np.uint32))
    assert (out[1][0][..., :3].sum(-1) > 0).mean() > 0.2
    t.close()
