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
