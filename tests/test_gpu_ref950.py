"""GPU (-m gpu), where oracle/_ref_gfx950/ travelled: the HIP path against the reference's kernel file as ROCm's own
OpenCL tool chain builds it for this chip (real ROCm builtin library, no stand-ins; oracle/Makefile ref_gfx950),
executed through the HIP module API.  That build contracts dot() into fma chains, normalizes by v_rsq_f32 and
divides by v_rcp_f32 (profiles/r02_ref_gfx950_builtins.md), so it cannot agree bit for bit with the IEEE-plain
contract the oracle is pinned to; the table index is a hash of the ray direction, so a 1-ulp difference re-routes a
path.  What must hold — and is asserted here — is agreement IN DISTRIBUTION: most pixel-samples identical, and the
frame means of the two renderers no further apart than two independent sample sets of one renderer."""
import os
import sys

import numpy as np
import pytest

import cases

sys.path.insert(0, os.path.join(cases.ROOT, "oracle"))
import oracle as orc  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not orc.ReferenceGfx950.available(),
                                                  reason="oracle/_ref_gfx950 (gfx950 build of the reference) not present")]
rt = cases.rt


@pytest.mark.parametrize("name,kw,spp", [("c2", dict(width=480, height=270), 64),
                                         ("c4", dict(width=240, height=136, n_spheres=1500), 16)])
def test_hip_path_agrees_in_distribution_with_the_rocm_opencl_build(name, kw, spp):
    wl = rt.workloads.get(name, **kw)
    W, H = wl.width, wl.height
    t = rt.RayTracer(W, H, scene=wl.scene, seed=cases.SEED)
    table = t.getRandomTable()
    ref = orc.ReferenceGfx950()

    def ours(first, count):
        t.clear()
        t.renderSamples(wl.camera, first, count)
        t.sync()
        return t.readLinear()[..., :3].astype(np.float64)

    # one sample, pixel by pixel: the same bits wherever the two arithmetic contracts round alike
    _, last = ref.render(wl.scene, wl.camera, table, W, H, 3, 1, want_last=True)
    mine = ours(3, 1).astype(np.float32)
    same = (mine.view(np.uint32) == last[..., :3].view(np.uint32)).all(axis=2).mean()
    assert same > 0.9, same
    # the spp-sample means: as close as two sample sets of ONE renderer are to each other
    a = ref.render(wl.scene, wl.camera, table, W, H, 0, spp)[..., :3].astype(np.float64) / spp
    b, b2 = ours(0, spp), ours(spp, spp)
    noise = np.abs(b2 - b).mean()
    assert np.abs(a - b).mean() < noise
    rel = abs(a.mean() - b.mean()) / b.mean()
    assert rel < 5 * max(abs(b2.mean() - b.mean()) / b.mean(), 2e-4), rel
    assert a.mean() > 0.01     # the frames are lit
    t.close()
