"""CPU, build container only: the oracle (C restatement) directly against the compiled
reference kernel (oracle/_ref) on fresh random inputs — skipped where oracle/_ref does
not exist (the GPU box, unless the prebuilt library travelled)."""
import numpy as np
import pytest

import cases
from oracle import Reference

pytestmark = pytest.mark.skipif(not Reference.available(), reason="oracle/_ref/libref.so not built here")


@pytest.fixture(scope="module")
def ref():
    return Reference()


@pytest.mark.parametrize("name,kw", [
    ("all_kinds", dict(width=200, height=120)),
    ("c2", dict(width=160, height=90)),
    ("c3", dict(width=160, height=90, tex_size=32)),
    ("c4", dict(width=64, height=36, n_spheres=500)),
    ("c5", dict(width=64, height=36, segments=12, rings=8)),
])
def test_progressive_frames_bit_exact(name, kw, oracle, ref, table):
    wl = cases.rt.workloads.get(name, **kw)
    a = ref.progressive(wl.scene, wl.camera, table, wl.width, wl.height, 6, threads=4)
    b, _ = oracle.render(wl.scene, wl.camera, table, wl.width, wl.height, 2, count=6, threads=4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_random_probes_other_seed_and_camera(oracle, ref):
    table = oracle.make_random_table(12345)
    wl = cases.rt.workloads.get("all_kinds", width=333, height=211)
    cam = cases.rt.Camera(75, 333 / 211, (-5.0, -2.5, -9.0), 30.0, 12.0).transferData()
    rng = np.random.RandomState(7)
    n = 20000
    xs, ys, ss = rng.randint(0, 333, n), rng.randint(0, 211, n), rng.randint(0, 65535, n)
    a = ref.samples(wl.scene, cam, table, 333, 211, xs, ys, ss)
    b, _ = oracle.samples(wl.scene, cam, table, 333, 211, xs, ys, ss)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (a.sum(1) > 0).mean() > 0.2  # the probes are not all black


def test_camera_inside_glass_and_grazing(oracle, ref, table):
    """Origins inside the refractive sphere / lens and a camera skimming the floor."""
    wl = cases.rt.workloads.get("all_kinds", width=96, height=64)
    for pos, yaw, pitch in (((0.25, 3.1, 0.1), 10.0, 5.0), ((5.2, 0.4, 0.3), 200.0, -20.0), ((0, 4.99, -9), 0.0, 0.5)):
        cam = cases.rt.Camera(60, 1.5, pos, yaw, pitch).transferData()
        a = ref.progressive(wl.scene, cam, table, 96, 64, 3, threads=4)
        b, _ = oracle.render(wl.scene, cam, table, 96, 64, 2, count=3, threads=4)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_unit_vectors_ten_thousand_per_routine(oracle, ref, table):
    """10^4 random (ray, primitive) pairs per intersection routine and 10^4 records per material
    routine, oracle against the compiled reference kernel, bit for bit (fresh streams, not the committed ones)."""
    wl = cases.workload("all_kinds")
    n = 10000
    for i, kind in enumerate(("sphere", "plane", "lens")):
        rays, prim, _ = cases.unit_rays(kind, wl.scene, n, 300 + 10 * i)
        assert np.array_equal(oracle.hit(i, wl.scene, rays, prim).view(np.uint32),
                              ref.hit(i, wl.scene, rays, prim).view(np.uint32)), kind
    rays, mesh, face = cases.unit_rays("triangle", wl.scene, n, 340)
    assert np.array_equal(oracle.hit_triangle(wl.scene, rays, mesh, face).view(np.uint32),
                          ref.hit_triangle(wl.scene, rays, mesh, face).view(np.uint32))
    for i, routine in enumerate(cases.MATERIAL_ROUTINES):
        vec = cases.material_vectors(wl.scene, n, 400 + 10 * i, routine)
        assert np.array_equal(oracle.material(i, wl.scene, table, vec).view(np.uint32),
                              ref.material(i, wl.scene, table, vec).view(np.uint32)), routine
