"""GPU (-m gpu): the device routines one by one, through the C ABI's unit-probe entry points
(rt_debug_hit / rt_debug_material / rt_debug_div3), against

(a) tests/golden/units.npz — 2 000 vectors per routine made by the compiled reference kernel
    (hitSphere, hitPlane, hitLens, hitTriangle, hitScene; rayReflect, rayRefract, rayScatter,
    rayRefractDielectric — raytracer.cl:149-435), and
(b) the oracle on 10 000 fresh vectors per routine.

Bar: every float of every record bit for bit (uint32 compare)."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
rt = cases.rt


@pytest.fixture(scope="module")
def setup():
    wl = cases.workload("all_kinds")
    t = rt.RayTracer(64, 64, scene=wl.scene, device=0, seed=cases.SEED)
    yield wl, t
    t.close()


def golden():
    return np.load(os.path.join(cases.GOLDEN_DIR, "units.npz"))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("accel", [0, 2])
def test_intersection_routines_match_the_reference_vectors(setup, accel):
    wl, t = setup
    t.setOption(t.OPT_ACCEL, accel)   # hitScene once by brute force, once through the BVHs
    g = golden()
    n = g["sphere"].shape[0]
    for i, kind in enumerate(("sphere", "plane", "lens")):
        rays, prim, _ = cases.unit_rays(kind, wl.scene, n, 100 + 10 * i)
        assert np.array_equal(bits(t.debugHit(i, rays, prim)), g[kind]), kind
    rays, mesh, face = cases.unit_rays("triangle", wl.scene, n, 140)
    assert np.array_equal(bits(t.debugHit(4, rays, mesh, face)), g["triangle"])
    rays, prim, _ = cases.unit_rays("scene", wl.scene, n, 150)
    hs = t.debugHit(3, rays)
    exp = g["scene"].view(np.float32)
    nonmesh = exp[:, 0] > 0
    hs[:, 8:11] = 0    # uv / texture id of non-mesh hits are indeterminate in the reference (see gen_golden.py)
    assert np.array_equal(bits(hs)[nonmesh], g["scene"][nonmesh])
    assert np.array_equal(hs[:, 0], exp[:, 0])
    t.setOption(t.OPT_ACCEL, 1)


def test_material_routines_match_the_reference_vectors(setup):
    wl, t = setup
    g = golden()
    n = g["mat_reflect"].shape[0]
    for i, routine in enumerate(cases.MATERIAL_ROUTINES):
        vec = cases.material_vectors(wl.scene, n, 200 + 10 * i, routine)
        assert np.array_equal(bits(t.debugMaterial(i, vec)), g["mat_" + routine]), routine


def test_ten_thousand_fresh_vectors_per_routine_vs_oracle(setup, oracle):
    wl, t = setup
    table = t.getRandomTable()
    n = 10000
    for i, kind in enumerate(("sphere", "plane", "lens")):
        rays, prim, _ = cases.unit_rays(kind, wl.scene, n, 500 + 10 * i)
        assert np.array_equal(bits(t.debugHit(i, rays, prim)), bits(oracle.hit(i, wl.scene, rays, prim))), kind
    rays, mesh, face = cases.unit_rays("triangle", wl.scene, n, 540)
    assert np.array_equal(bits(t.debugHit(4, rays, mesh, face)), bits(oracle.hit_triangle(wl.scene, rays, mesh, face)))
    rays, prim, _ = cases.unit_rays("scene", wl.scene, n, 550)
    a, b = t.debugHit(3, rays), oracle.hit(3, wl.scene, rays, prim)
    mesh_hit = (b[:, 0] > 0) & (b[:, 10].view(np.uint32) != 0) | (b[:, 8] != 0) | (b[:, 9] != 0)
    a[~mesh_hit, 8:11] = 0
    b[~mesh_hit, 8:11] = 0
    assert np.array_equal(bits(a), bits(b))
    for i, routine in enumerate(cases.MATERIAL_ROUTINES):
        vec = cases.material_vectors(wl.scene, n, 600 + 10 * i, routine)
        assert np.array_equal(bits(t.debugMaterial(i, vec)), bits(oracle.material(i, wl.scene, table, vec))), routine


def test_unit_probe_argument_checks(setup):
    wl, t = setup
    rays = np.zeros((2, 6), np.float32)
    with pytest.raises(rt.RtError):
        t.debugHit(0, rays, np.array([0, 10 ** 6], np.uint32))        # sphere that does not exist
    with pytest.raises(rt.RtError):
        t.debugHit(7, rays)                                           # unknown routine
    vec = cases.material_vectors(wl.scene, 2, 1)
    vec.view(np.uint32)[1, 12] = 999
    with pytest.raises(rt.RtError):
        t.debugMaterial(2, vec)                                       # material that does not exist


def test_div3_is_three_ieee_divisions(setup):
    """The shared-reciprocal form of x/d, y/d, z/d (pt_device.hpp div3) against the compiler's three IEEE
    divisions, bit for bit: operands across the whole exponent range (the in-range test decides which
    records take the short sequence; out-of-range records must fall back), signed zeros, infinities, NaN."""
    _, t = setup
    rng = np.random.RandomState(3)
    n = 1 << 22
    total_short = 0
    for rnd in range(4):
        m = rng.randint(0, 1 << 23, (n, 4)).astype(np.uint32)
        sign = rng.randint(0, 2, (n, 4)).astype(np.uint32) << 31
        if rnd == 0:      # everything near 1: the regime of normalize() and of the sphere normal
            e = rng.randint(117, 137, (n, 4)).astype(np.uint32)
        elif rnd == 1:    # the guard's edges
            e = np.stack([rng.randint(30, 45, n), rng.randint(180, 195, n), rng.randint(30, 195, n),
                          rng.randint(90, 164, n)], axis=1).astype(np.uint32)
        else:             # all exponents, denormals, inf / NaN included
            e = rng.randint(0, 256, (n, 4)).astype(np.uint32)
        v = (sign | (e << 23) | m).view(np.float32)
        if rnd == 3:
            v[::7, 0] = 0.0
            v[3::11, 1] = -0.0
            v[5::13, 3] = 1.0
        out = t.debugDiv3(v)
        a, b = out[:, :3].view(np.uint32), out[:, 3:].view(np.uint32)
        nan = np.isnan(out[:, :3]) & np.isnan(out[:, 3:])
        assert ((a == b) | nan).all(), "round %d: %d differ" % (rnd, int((~((a == b) | nan)).sum()))
        av = np.abs(v.view(np.uint32) & 0x7FFFFFFF)
        short = (av[:, :3].min(1) >= 0x12800000) & (av[:, :3].max(1) <= 0x5D800000) & (av[:, 3] >= 0x30800000) & \
                (av[:, 3] <= 0x4E800000)
        total_short += int(short.sum())
    assert total_short > n     # the short sequence was actually exercised on millions of operand sets
