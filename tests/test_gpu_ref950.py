"""GPU (-m gpu), where oracle/_ref_gfx950/ travelled: the HIP path against the reference's kernel file as ROCm's OWN
OpenCL tool chain builds it for this chip (real ROCm builtin library — no stand-ins; oracle/Makefile ref_gfx950),
executed through the HIP module API.  This is the pin of parity to a REAL OpenCL build of the reference:

  * arithmetic policy 1 (RT_ARITH_ROCM_OCL_NOCONTRACT) must equal ref950_nocontract.hsaco, and
  * arithmetic policy 2 (RT_ARITH_ROCM_OCL) must equal ref950.hsaco (OpenCL's default flags: what an unmodified
    KernelGL::buildProgram, src/kernelgl.cpp:95-106, gets)

BIT FOR BIT PER PIXEL-SAMPLE — every builtin on a million operand sets, every pixel of samples 0-3 of C1 (full
frame), C2 (1920x1080), C3 / all-kinds (lens, small meshes by face scan; their t_textured materials turned diffuse:
an OpenCL image object cannot be made from a HIP process, so the texel fetch raytracer.cl:105-107 stays unpinned), C4
with 2 000 and with 100 000 spheres (sphere BVH), C5 with 720 and with 50 000 faces (mesh BVH, walk slices) — and
the fused frames within north_star's 1e-4.  The default policy 0 (IEEE-plain, the CPU oracle's contract) cannot agree
bit for bit with either build (v_rsq / v_rcp / fused dot re-route paths through the hash-indexed table); its
measured distance is asserted too, so that a regression in distance shows."""
import os
import sys

import numpy as np
import pytest

import cases

sys.path.insert(0, os.path.join(cases.ROOT, "oracle"))
import oracle as orc  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (orc.ReferenceGfx950.available() and orc.ReferenceGfx950.available(orc.REF950_HSACO_NOCONTRACT)),
                                                  reason="oracle/_ref_gfx950 (gfx950 builds of the reference) not present")]
rt = cases.rt
HSACO = {1: orc.REF950_HSACO_NOCONTRACT, 2: orc.REF950_HSACO}
BUILTINS = ["dot", "cross", "normalize", "divide", "sqrt", "mix", "min", "sign", "pow5", "hash"]


def _operands(n, rng, name):
    v = rng.standard_normal((n, 8)).astype(np.float32)
    v[:, :7] *= np.exp2(rng.integers(-12, 12, size=(n, 1))).astype(np.float32)
    v[::97, 0] = 0.0
    v[::101, :3] = 0.0                       # normalize(0) returns 0 in ROCm's library, NaN in the IEEE-plain contract
    v[::103, 1] = np.float32(1e-30)
    v[::107, 0] = np.float32(3e38)           # dot(v, v) overflows: normalize's rescaling branch
    v[::109, :3] *= np.float32(1e-25)        # ... and its denormal branch
    v[::113, 6] = np.float32(1e-41)          # division by a denormal
    if name in ("sqrt", "pow5"):
        v[:, 0] = np.abs(v[:, 0])
    if name == "pow5":
        v[:, 0] = rng.random(n).astype(np.float32) * 1.2     # 1 - cos of the incident angle (schlick, :404)
    if name == "hash":
        v[:, :3] = rng.standard_normal((n, 3)).astype(np.float32)
    return v


@pytest.mark.parametrize("policy", [1, 2])
def test_every_builtin_of_the_policy_is_rocm_opencls_bit_for_bit(policy):
    ref = orc.ReferenceGfx950(HSACO[policy])
    t = rt.RayTracer(8, 8, scene=rt.workloads.get("c1", width=8, height=8).scene)
    t.setArith(policy)
    rng = np.random.default_rng(11)
    for op, name in enumerate(BUILTINS):
        v = _operands(1 << 18, rng, name)
        a, b = t.debugBuiltin(op, v), ref.builtin(op, v)
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), (name, int((~same).any(axis=1).sum()), v[(~same).any(axis=1)][:3], a[(~same).any(axis=1)][:3],
                            b[(~same).any(axis=1)][:3])
    t.close()


# name → (workload, generator arguments, corner compared (None = the whole frame), spp of the fused comparison)
FRAMES = {
    "c1": ("c1", dict(width=256, height=256), None, 16),
    "c2_1080p": ("c2", dict(width=1920, height=1080), None, 64),
    "c3_untextured": ("c3", dict(width=480, height=270), None, 16),
    "all_kinds_untextured": ("all_kinds", dict(width=320, height=200), None, 16),
    "c4_2000_spheres": ("c4", dict(width=480, height=270, n_spheres=2000), None, 16),
    "c4_100000_spheres": ("c4", dict(width=160, height=90), None, 2),
    "c4_100000_spheres_corner_of_1080p": ("c4", dict(width=1920, height=1080), (128, 32), 2),
    "c5_720_faces": ("c5", dict(width=480, height=270, segments=24, rings=16), None, 16),
    "c5_50000_faces": ("c5", dict(width=160, height=90), None, 4),
    "c5_50000_faces_corner_of_1080p": ("c5", dict(width=1920, height=1080), (128, 32), 4),
}


def _ours(t, wl, first, count, gw, gh):
    t.clear()
    t.renderSamples(wl.camera, first, count)
    t.sync()
    return t.readLinear()[:gh, :gw, :3].astype(np.float64) * count      # mean → sum


@pytest.mark.parametrize("policy", [1, 2])
@pytest.mark.parametrize("name", sorted(FRAMES))
def test_every_pixel_sample_is_bit_identical_to_the_rocm_opencl_build(name, policy):
    wname, kw, grid, spp = FRAMES[name]
    wl = rt.workloads.get(wname, **kw)
    if wl.scene.texture_args()[3]:
        wl = rt.workloads.untextured(wl)
    W, H = wl.width, wl.height
    gw, gh = grid if grid else (W, H)
    ref = orc.ReferenceGfx950(HSACO[policy])
    t = rt.RayTracer(W, H, scene=wl.scene, seed=cases.SEED)
    t.setArith(policy)
    t.setOption(t.OPT_PREFIX_TREE, 2)   # the decision trees in the one-sample calls below too (default: from 24 samples per call on)
    t.resetCounters()
    table = t.getRandomTable()
    lit = 0.0
    for k in range(4 if spp >= 4 else spp):
        _, last = ref.render(wl.scene, wl.camera, table, W, H, k, 1, want_last=True, grid=grid)
        mine = _ours(t, wl, k, 1, gw, gh).astype(np.float32)     # through pt_prefix + pt_samples_q / pt_samples_w
        theirs = last[:gh, :gw, :3]
        same = (mine.view(np.uint32) == theirs.view(np.uint32)).all(axis=2)
        assert same.all(), (name, policy, k, int((~same).sum()))
        lit = max(lit, float((theirs.sum(axis=2) > 0).mean()))
    assert lit > 0.02, lit                                       # the compared region is not just sky
    # the fused frame (all spp in one call, our summation order) against the reference's samples summed one by one
    a = ref.render(wl.scene, wl.camera, table, W, H, 0, spp, grid=grid)[:gh, :gw, :3].astype(np.float64) / spp
    b = _ours(t, wl, 0, spp, gw, gh) / spp
    rel = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)
    assert rel.max() <= 1e-4, rel.max()
    # the individual-sample probe path (pt_probe → trace_from, no prefix sharing) computes the same bits
    rng = np.random.RandomState(5)
    n = 256
    xs, ys, ss = rng.randint(0, gw, n), rng.randint(0, gh, n), rng.randint(0, 4 if spp >= 4 else spp, n)
    got = t.traceSamples(wl.camera, xs, ys, ss)
    frames = {}
    for k in sorted(set(ss.tolist())):
        frames[k] = ref.render(wl.scene, wl.camera, table, W, H, k, 1, want_last=True, grid=grid)[1]
    exp = np.stack([frames[int(s)][y, x, :3] for x, y, s in zip(xs, ys, ss)])
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    assert t.walkOverflow() == 0
    t.close()


@pytest.mark.parametrize("policy", [1, 2])
def test_the_brute_force_search_and_the_bvhs_agree_under_the_policy(policy):
    """RT_OPT_ACCEL 0 (the reference's loops) against the default (sphere BVH, mesh BVH, walk slices) with the policy's
    arithmetic: the culling margins were derived for policy 0 (see pt_arith.hpp for why they hold a fortiori)."""
    for wname, kw, spp in (("c4", dict(width=240, height=136, n_spheres=20000), 4),
                           ("c5", dict(width=240, height=136, segments=60, rings=40), 8)):
        wl = rt.workloads.get(wname, **kw)
        t = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=cases.SEED)
        t.setArith(policy)
        t.resetCounters()
        frames = []
        for accel in (0, 1):
            t.setOption(t.OPT_ACCEL, accel)
            t.clear()
            t.renderSamples(wl.camera, 0, spp)
            t.sync()
            frames.append(t.readLinear().copy())
        assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)), wname
        assert t.walkOverflow() == 0
        t.close()


@pytest.mark.parametrize("policy", [1, 2])
def test_compat_path_runs_under_the_policy(policy):
    """rt_render / rt_render_again (kernels `trace` / `retrace`) under policies 1 / 2: the image after `trace` is the
    policy's 3-ulp sqrt of the very sample the reference build computes.  (The reference's own trace / retrace kernels
    take OpenCL image objects and cannot be launched from HIP: their gamma arithmetic, raytracer.cl:488-494,:526-531,
    is built from the same policy builtins but is NOT compared with a real OpenCL run.)"""
    wl = rt.workloads.get("c2", width=240, height=136)
    ref = orc.ReferenceGfx950(HSACO[policy])
    t = rt.RayTracer(wl.width, wl.height, scene=wl.scene, seed=cases.SEED)
    t.setArith(policy)
    table = t.getRandomTable()
    t.render(wl.camera)
    img = t.transferImage()
    _, s0 = ref.render(wl.scene, wl.camera, table, wl.width, wl.height, 0, 1, want_last=True)
    exp = np.sqrt(s0[..., :3].astype(np.float64))
    assert np.abs(img[..., :3] - exp).max() <= 4e-7 * max(1.0, exp.max())       # 3 ulp
    t.renderAgain(wl.camera)
    img1 = t.transferImage()
    _, s1 = ref.render(wl.scene, wl.camera, table, wl.width, wl.height, 1, 1, want_last=True)
    mean = (s0[..., :3].astype(np.float64) + s1[..., :3]) / 2
    assert np.abs(img1[..., :3].astype(np.float64) ** 2 - mean).max() <= 2e-6 * max(1.0, mean.max())
    assert t.sample_counter == 1
    t.close()


# policy 0 (IEEE-plain: the CPU oracle's contract) vs the default ROCm-OpenCL build — measured distance, asserted with a
# small slack so that a change of distance shows (profiles/r02_ref_distance_default.json: C2 97.7 % of pixel-samples
# identical, C4 with 2 000 spheres 95.8 %, C5 with 720 faces 99.2 %)
@pytest.mark.parametrize("name,kw,spp,min_same,min_within", [
    ("c2", dict(width=480, height=270), 64, 0.970, 0.80),
    ("c4", dict(width=480, height=270, n_spheres=2000), 16, 0.950, 0.72),
    ("c5", dict(width=480, height=270, segments=24, rings=16), 16, 0.985, 0.93)])
def test_ieee_policy_distance_to_the_rocm_opencl_build(name, kw, spp, min_same, min_within):
    wl = rt.workloads.get(name, **kw)
    W, H = wl.width, wl.height
    t = rt.RayTracer(W, H, scene=wl.scene, seed=cases.SEED)     # default policy: IEEE
    table = t.getRandomTable()
    ref = orc.ReferenceGfx950(HSACO[2])
    _, last = ref.render(wl.scene, wl.camera, table, W, H, 3, 1, want_last=True)
    mine = _ours(t, wl, 3, 1, W, H).astype(np.float32)
    same = (mine.view(np.uint32) == last[..., :3].view(np.uint32)).all(axis=2).mean()
    assert min_same <= same < 1.0, same      # (== 1.0 would mean the policies are not what they say)
    a = ref.render(wl.scene, wl.camera, table, W, H, 0, spp)[..., :3].astype(np.float64) / spp
    b, b2 = _ours(t, wl, 0, spp, W, H) / spp, _ours(t, wl, spp, spp, W, H) / spp
    within = (np.abs(a - b) <= 1e-4 * np.maximum(np.abs(a), 1e-6)).mean()
    assert within >= min_within, within
    # agreement in distribution: closer to the OpenCL build than two sample sets of one renderer are to each other
    assert np.abs(a - b).mean() < np.abs(b2 - b).mean()
    t.close()
