"""Random scenes through every instantiation of the trace kernels, bit for bit against the oracle.

Each seed draws a scene of random size and make-up — spheres (a handful, or enough for the sphere BVH), planes,
lenses, uv-sphere meshes below and above the mesh-BVH threshold, all six material kinds, a random camera — and
checks (a) 1 200 random pixel-samples of the probe kernel, (b) the fused frame (pt_prefix + pt_samples_q / _w at 8 or 64 spp)
against the oracle's per-sample values summed in kernel order, (c) the acceleration structures on against off.
The scenes are what the golden vectors are not: arbitrary."""
import numpy as np
import pytest

import cases
from test_gpu_parity import fused_sum_in_kernel_order

pytestmark = pytest.mark.gpu
rt = cases.rt
A = rt._abi


def random_scene(seed):
    g = np.random.RandomState(seed)
    s = rt.SceneCreator()
    kinds = [A.T_DIFFUSE, A.T_REFLECTIVE, A.T_DIELECTRIC, A.T_LIGHT, A.T_DIFFUSE, A.T_REFRACTIVE]
    for k in kinds:                                        # 0..5 plain materials
        extra = {A.T_REFLECTIVE: 0.5 + 0.5 * g.rand(), A.T_DIELECTRIC: 1.1 + 0.6 * g.rand(), A.T_REFRACTIVE: 1.1 + 0.6 * g.rand()}.get(k, 1.0)
        s.addMaterial(k, tuple(0.3 + 0.7 * g.rand(3)), extra)
    s.addMaterial(A.T_TEXTURED, (1, 1, 1), 1)              # 6: only ever on meshes with uv
    n_sph = int(g.choice([3, 12, 40, 90, 400]))            # >= 64: sphere BVH
    pos = np.stack([g.uniform(-9, 9, n_sph), g.uniform(-2.5, 4.0, n_sph), g.uniform(-9, 9, n_sph)], 1).astype(np.float32)
    rad = (0.15 + 1.2 * g.rand(n_sph) ** 3).astype(np.float32)
    s.addSpheres(pos, rad, g.randint(0, 6, n_sph).astype(np.uint32))
    s.addSphere((2.0, -60.0, 1.0), 40.0, 3)                # a light overhead
    for _ in range(int(g.randint(1, 3))):
        n = g.normal(size=3)
        n[1] = abs(n[1]) + 1.5
        s.addPlane((0.0, 5.0 + 2.0 * g.rand(), 0.0), tuple(n / np.linalg.norm(n)), int(g.choice([0, 4, 1])))
    for _ in range(int(g.randint(0, 3))):
        h = 0.4 + 0.6 * g.rand()
        n = g.normal(size=3)
        s.addLens(tuple(g.uniform(-4, 4, 3)), tuple(n / np.linalg.norm(n)), h + 0.2 + g.rand(), h + 0.2 + g.rand(), h, int(g.choice([2, 5])))
    n_models = int(g.randint(0, 3))
    textured = False
    for _ in range(n_models):
        seg, rings = [(5, 4), (12, 8), (30, 20)][int(g.randint(0, 3))]     # 30, 168 and 1 140 faces: below / above the BVH threshold
        p, uv, idx = rt.workloads.uv_sphere(seg, rings, radius=float(0.8 + 1.5 * g.rand()), centre=tuple(g.uniform(-5, 5, 3) * (1, 0.5, 1)))
        tex = bool(g.rand() < 0.4)
        s.addMesh(p, uv, idx, texture_ID=0 if tex else 0xFFFFFFFF)
        s.addModel(1, 6 if tex else int(g.choice([0, 2, 1, 5])))
        textured |= tex
    if textured:
        s.setTextures(rt.workloads.checker_texture(16, 4))
    eye = tuple(g.uniform(-7, 7, 3) * (1, 0.3, 1) + (0, 0.5, 0))
    cam = rt.Camera(float(g.uniform(40, 80)), 16 / 9, eye, float(g.uniform(0, 360)), float(g.uniform(-10, 15))).transferData()
    return s, cam


@pytest.mark.parametrize("seed", list(range(101, 125)))
def test_random_scene_bit_exact(seed, oracle, table):
    s, cam = random_scene(seed)
    W, H, spp = 64, 36, (64 if seed % 3 == 0 else 8)   # 64: six pixels per wave, full waves of the sample queue
    t = rt.RayTracer(W, H, scene=s, seed=cases.SEED)
    t.setOption(t.OPT_PREFIX_TREE, 2 if seed % 2 else 1)   # odd seeds: decision trees in the 8-sample calls too (default: from 24 on)
    t.setOption(t.OPT_WAVE_FILL, 0 if seed % 3 == 1 else 1)  # a third of the seeds: as many pixels per wave as the LDS share holds
    g = np.random.RandomState(seed + 5000)
    n = 1200
    xs, ys, ss = g.randint(0, W, n), g.randint(0, H, n), g.randint(0, 3000, n)
    exp, _ = oracle.samples(s, cam, table, W, H, xs, ys, ss)
    frames = []
    for accel in (1, 0):
        t.setOption(t.OPT_ACCEL, accel)
        got = t.traceSamples(cam, xs, ys, ss)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), "probes, accel %d" % accel
        t.clear()
        t.renderSamples(cam, 0, spp)
        t.sync()
        frames.append(t.readLinear())
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    yy, xx, sm = np.meshgrid(np.arange(H), np.arange(W), np.arange(spp), indexing="ij")
    per, _ = oracle.samples(s, cam, table, W, H, xx.ravel(), yy.ravel(), sm.ravel())
    exp_sum = fused_sum_in_kernel_order(per.reshape(H * W, spp, 3), spp).reshape(H, W, 3)
    exp_lin = (exp_sum / np.float32(spp)).astype(np.float32)
    assert np.array_equal(frames[0][..., :3].view(np.uint32), exp_lin.view(np.uint32))
    t.close()


# ---- the same random scenes under the ROCm-OpenCL arithmetic policies, against the reference's gfx950 OpenCL code objects
import os  # noqa: E402
import sys  # noqa: E402

sys.path.insert(0, os.path.join(cases.ROOT, "oracle"))
import oracle as orc  # noqa: E402

_HSACO = {1: orc.REF950_HSACO_NOCONTRACT, 2: orc.REF950_HSACO}


@pytest.mark.skipif(not (orc.ReferenceGfx950.available() and orc.ReferenceGfx950.available(orc.REF950_HSACO_NOCONTRACT)),
                    reason="oracle/_ref_gfx950 (gfx950 builds of the reference) not present")
@pytest.mark.parametrize("seed", list(range(201, 217)))
def test_random_scene_bit_exact_against_the_rocm_opencl_build(seed):
    """Arbitrary scenes (spheres below and above the BVH threshold, planes, lenses, meshes below and above theirs, every
    material kind — textured turned diffuse: no OpenCL image object from HIP), policies 1 and 2 alternating: every pixel
    of samples 0-2 of the frame bit-identical to the reference kernel file as ROCm's OpenCL tool chain builds it, with
    the acceleration structures on and off; the fused frame within 1e-4."""
    policy = 1 + seed % 2
    s, cam = random_scene(seed)
    s.materials["type"][s.materials["type"] == A.T_TEXTURED] = A.T_DIFFUSE
    s.textures = None
    W, H, spp = 96, 54, (64 if seed % 3 == 0 else 8)
    ref = orc.ReferenceGfx950(_HSACO[policy])
    t = rt.RayTracer(W, H, scene=s, seed=cases.SEED)
    t.setArith(policy)
    t.setOption(t.OPT_PREFIX_TREE, 2 if seed % 4 < 2 else 1)   # the decision trees in the one-sample calls too, for half the seeds
    t.setOption(t.OPT_WAVE_FILL, 0 if seed % 3 == 1 else 1)
    t.resetCounters()
    table = t.getRandomTable()
    for k in range(3):
        _, last = ref.render(s, cam, table, W, H, k, 1, want_last=True)
        for accel in (1, 0):
            t.setOption(t.OPT_ACCEL, accel)
            t.clear()
            t.renderSamples(cam, k, 1)
            t.sync()
            mine = t.readLinear()[..., :3]
            same = (mine.view(np.uint32) == last[..., :3].view(np.uint32)).all(axis=2)
            assert same.all(), (seed, policy, k, accel, int((~same).sum()))
    t.setOption(t.OPT_ACCEL, 1)
    a = ref.render(s, cam, table, W, H, 0, spp)[..., :3].astype(np.float64) / spp
    t.clear()
    t.renderSamples(cam, 0, spp)
    t.sync()
    b = t.readLinear()[..., :3].astype(np.float64)
    rel = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)
    assert rel.max() <= 1e-4, rel.max()
    assert t.walkOverflow() == 0
    t.close()
