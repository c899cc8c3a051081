"""CPU: the oracle (C restatement) against the committed golden vectors, which were
produced by the unmodified reference kernel compiled for x86-64
(tests/golden/gen_golden.py).  Bit-exact."""
import os

import numpy as np
import pytest

import cases

ALL = list(cases.CASES)


def load(name):
    return np.load(os.path.join(cases.GOLDEN_DIR, name + ".npz"))


@pytest.mark.parametrize("name", ALL)
def test_fixture_inputs_unchanged(name, table):
    """The deterministic workload generators still produce the scene the fixture was made from."""
    g = load(name)
    wl = cases.workload(name)
    assert str(g["scene_hash"]) == cases.scene_hash(wl, table), \
        "workload %s drifted: regenerate tests/golden with gen_golden.py" % name
    assert np.array_equal(g["camera"], wl.camera)


@pytest.mark.parametrize("name", ALL)
def test_per_sample_radiance_bit_exact(name, oracle, table):
    g = load(name)
    wl = cases.workload(name)
    got, _ = oracle.samples(wl.scene, wl.camera, table, wl.width, wl.height, g["probe_x"], g["probe_y"], g["probe_s"])
    assert np.array_equal(got.view(np.uint32), g["probe_rgb_bits"])


@pytest.mark.parametrize("name", ALL)
def test_progressive_crop_bit_exact(name, oracle, table):
    """trace + (spp-1) retrace on the crop: the reference's gamma-space running mean."""
    g = load(name)
    wl = cases.workload(name)
    x0, y0, cw, ch = (int(v) for v in g["crop"])
    img, cn = oracle.render(wl.scene, wl.camera, table, wl.width, wl.height, 2, count=int(g["spp"]),
                            region=(x0, y0, cw, ch), threads=4)
    assert np.array_equal(img[y0:y0 + ch, x0:x0 + cw].view(np.uint32), g["crop_rgba_bits"])
    assert cn.samples == cw * ch * int(g["spp"])
    assert cn.image_reads == cw * ch * (int(g["spp"]) - 1)


@pytest.mark.parametrize("name", [n for n in ALL if cases.CASES[n]["full_frame_spp"]])
def test_full_frame_trace_checksum(name, oracle, table):
    g = load(name)
    wl = cases.workload(name)
    img, _ = oracle.render(wl.scene, wl.camera, table, wl.width, wl.height, 0, threads=8)
    assert cases.frame_checksum(img) == int(g["full_trace_checksum"])


def test_unit_intersections_bit_exact(oracle):
    g = load("units")
    wl = cases.workload("all_kinds")
    n = g["sphere"].shape[0]
    for i, kind in enumerate(("sphere", "plane", "lens")):
        rays, prim, _ = cases.unit_rays(kind, wl.scene, n, 100 + 10 * i)
        assert np.array_equal(oracle.hit(i, wl.scene, rays, prim).view(np.uint32), g[kind]), kind
    rays, mesh, face = cases.unit_rays("triangle", wl.scene, n, 140)
    assert np.array_equal(oracle.hit_triangle(wl.scene, rays, mesh, face).view(np.uint32), g["triangle"])
    rays, prim, _ = cases.unit_rays("scene", wl.scene, n, 150)
    hs = oracle.hit(3, wl.scene, rays, prim)
    exp = g["scene"].view(np.float32)
    nonmesh = exp[:, 0] > 0
    hs[:, 8:11] = 0  # uv / texture id are compared through the textured full-path probes instead
    assert np.array_equal(hs.view(np.uint32)[nonmesh], g["scene"][nonmesh])
    assert np.array_equal(hs[:, 0], exp[:, 0])


def test_unit_material_routines_bit_exact(oracle, table):
    """rayReflect / rayRefract / rayScatter / rayRefractDielectric (raytracer.cl:362-435) of the oracle against the
    vectors the compiled reference kernel produced (tests/golden/gen_golden.py)."""
    g = load("units")
    wl = cases.workload("all_kinds")
    n = g["mat_reflect"].shape[0]
    for i, routine in enumerate(cases.MATERIAL_ROUTINES):
        vec = cases.material_vectors(wl.scene, n, 200 + 10 * i, routine)
        assert np.array_equal(oracle.material(i, wl.scene, table, vec).view(np.uint32), g["mat_" + routine]), routine


def test_random_table_three_implementations_agree(oracle, table):
    """numpy twin (workloads.py) == oracle C == product host function rt_make_random_table."""
    t_oracle = oracle.make_random_table(cases.SEED)
    t_lib = cases.rt.make_random_table(cases.SEED)
    assert np.array_equal(table.view(np.uint32), t_oracle.view(np.uint32))
    assert np.array_equal(table.view(np.uint32), t_lib.view(np.uint32))
    v = table[:300000].reshape(-1, 3).astype(np.float64)
    assert (np.sum(v * v, axis=1) < 1.0).all()          # inside the unit ball
    assert abs(np.mean(np.sqrt(np.sum(v * v, axis=1))) - 0.75) < 0.01   # E|x| of a uniform ball
    u = table[300000:]
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01
    other = oracle.make_random_table(cases.SEED + 1)
    assert not np.array_equal(other, table)
