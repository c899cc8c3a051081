"""CPU: the acceleration structures are host logic (built at rt_set_scene) — their invariants
are checked without a device through rt_debug_check_accel: every primitive in exactly one leaf and
inside its boxes, nested boxes, split axes and skip links, the kernels' threaded walk visiting
every leaf exactly once in all eight direction octants, smallest-face indices, normal cones."""
import numpy as np
import pytest

import cases

rt = cases.rt


@pytest.mark.parametrize("name,kw", [("c2", {}), ("c4", dict(n_spheres=100000)), ("c4", dict(n_spheres=777)),
                                     ("c5", {}), ("c5", dict(segments=24, rings=16)), ("c3", dict(tex_size=8)),
                                     ("all_kinds", {})])
def test_workload_structures(built, name, kw):
    wl = rt.workloads.get(name, **kw)
    st = rt.check_accel(wl.scene)
    n_sph = len(wl.scene.spheres)
    if n_sph:
        assert st["sphere_leaves"] >= (n_sph + 3) // 4 and st["sphere_nodes"] == 2 * st["sphere_leaves"] - 1
        assert st["sphere_depth"] <= 2 * int(np.ceil(np.log2(max(n_sph, 2)))) + 8     # SAH trees stay shallow
    big = [int(f) for f in wl.scene.meshes["face_count"] if f >= 32]
    assert st["meshes"] == len(big)
    if big:
        assert st["mesh_leaves"] >= sum((f + 1) // 2 for f in big)
        assert st["mesh_nodes"] == 2 * st["mesh_leaves"] - len(big)


def test_adversarial_spheres(built):
    s = rt.SceneCreator()
    s.addMaterial(rt._abi.T_DIFFUSE, (1, 1, 1), 1)
    u = rt.workloads.uniforms(5000, 9)
    pos = np.stack([u[:, 0] * 40 - 20, u[:, 1] * 8 - 4, u[:, 2] * 40 - 10], 1).astype(np.float32)
    s.addSpheres(pos, (0.02 + 0.6 * u[:, 3] ** 3).astype(np.float32), np.zeros(5000, np.uint32))
    s.addSpheres(pos[:300], np.full(300, 0.25, np.float32), np.zeros(300, np.uint32))      # coincident centres
    s.addSpheres(np.zeros((40, 3), np.float32), np.linspace(0.1, 4, 40).astype(np.float32), np.zeros(40, np.uint32))
    s.addSphere((0, -5000, 0), 4000, 0)
    s.addSphere((1e5, 0, 1e5), 1e-5, 0)
    s.addSphere((3, 3, 3), -1.5, 0)                                                        # negative radius
    st = rt.check_accel(s)
    assert st["sphere_leaves"] > 1000


def test_degenerate_and_tiny_meshes(built):
    s = rt.SceneCreator()
    s.addMaterial(rt._abi.T_DIFFUSE, (1, 1, 1), 1)
    pos, uv, idx = rt.workloads.uv_sphere(16, 10)
    pos = pos.copy()
    pos[3:6] = pos[3]                       # a zero-area face
    pos[30:33, :] = pos[30:33, :] * np.float32(1e-20)   # a microscopic face at the origin
    s.addMesh(pos, uv, idx)
    s.addMesh(pos[:9] + np.float32(5), uv[:9], idx[:9])   # 3 faces: below the BVH threshold
    s.addModel(2, 0)
    st = rt.check_accel(s)
    assert st["meshes"] == 1 and st["mesh_leaves"] >= 100
    empty = rt.SceneCreator()
    empty.addMaterial(rt._abi.T_DIFFUSE, (1, 1, 1), 1)
    assert rt.check_accel(empty)["sphere_nodes"] == 0


@pytest.mark.parametrize("scale", [1e-6, 3e-3, 1.0, 4e4, 1e9])
def test_packed_mesh_nodes_at_extreme_scales(built, scale):
    """The device's 48-byte mesh node carries half extent, sin alpha, longest edge and q as binary16
    (mesh_node_pack): rt_debug_check_accel unpacks every node and checks that each field is rounded
    to the safe side — here for meshes whose extents fall below binary16's normal range (1e-6: the
    smallest normal stands in) and above its largest value (4e4, 1e9: +inf, an unbounded box)."""
    s = rt.SceneCreator()
    s.addMaterial(rt._abi.T_DIELECTRIC, (1, 1, 1), 1.3)
    pos, uv, idx = rt.workloads.uv_sphere(40, 24, radius=2.0 * scale, centre=(3.0 * scale, -1.0 * scale, 2.0 * scale))
    s.addMesh(pos, uv, idx)
    s.addModel(1, 0)
    st = rt.check_accel(s)
    assert st["meshes"] == 1 and st["mesh_leaves"] >= 40 * 23
