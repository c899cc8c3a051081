import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Everything native is built once per session (no-op when up to date)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def table(built):
    import cases
    return cases.rt.workloads.make_random_table(cases.SEED)


@pytest.fixture(scope="session", autouse=True)
def _no_walk_ever_ends_on_its_loop_bound():
    """Every RayTracer a GPU test closes must report rt_walk_overflow() == 0: a BVH walk that leaves its loop on the
    iteration bound returns a possibly wrong nearest hit (round 2 shipped such a bound for a while and found it only
    through a flaky pixel diff)."""
    import cases
    rt = cases.rt
    orig = rt.RayTracer.close

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            flags = self.walkOverflow()
            orig(self)
            assert flags == 0, "a BVH walk ended on its loop bound (PT_OVF bits %d)" % flags
        else:
            orig(self)

    rt.RayTracer.close = close
    yield
    rt.RayTracer.close = orig


@pytest.fixture(autouse=True)
def _both_wave_fill_modes(request):
    """The test frames are small, and on a small launch a sample-kernel wave owns fewer pixels (RT_OPT_WAVE_FILL 1, the
    default).  So that the regime of the full-size frames — as many pixels per wave as its LDS share holds — stays
    covered, every other test (by its name) creates its RayTracers with RT_OPT_WAVE_FILL 0.  Results never depend on it."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import zlib
    import cases
    rt = cases.rt
    fill = zlib.crc32(request.node.nodeid.encode()) & 1
    orig = rt.RayTracer.__init__

    def init(self, *a, **kw):
        orig(self, *a, **kw)
        self.setOption(self.OPT_WAVE_FILL, fill)

    rt.RayTracer.__init__ = init
    yield
    rt.RayTracer.__init__ = orig
