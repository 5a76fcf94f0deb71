import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# before anything imports torch (its libgomp reads this once, and the oracle shares that runtime): the oracle's threads
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))
# the suite runs on the TESTING build of the library (the same kernel objects; the launch-form switches and the RCCL test double's seam compiled in); child processes inherit it.
# tests/test_capi_cpu.py and test_advance_gpu.py::test_release_build_ignores_the_switches load the release build next to it
os.environ.setdefault("VDN_LIB_FLAVOUR", "testing")
os.environ.setdefault("VO_POISON", "1")      # the oracle's work arrays are handed out full of NaN: a read of an unset entry shows up (oracle/vo_godunov.c)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_sessionstart(session):
    """the shared libraries are build products (git-ignored): build them when a fresh checkout has none"""
    lib = os.path.join(ROOT, "varden_amd", "csrc", "libvarden_amd_testing.so")
    olib = os.path.join(ROOT, "oracle", "libvoracle.so")
    if not (os.path.exists(lib) and os.path.exists(olib)):
        import __graft_entry__ as g
        g.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import voracle
    voracle.lib()
    return voracle


@pytest.fixture(scope="session")
def gpu():
    """initialised HIP runtime; fails loudly (no fallback) when the library or the GPU is missing"""
    from varden_amd import boxlib, capi
    capi.load()
    boxlib.initialize(capi.default_params(), 0, 1, 0)
    yield boxlib
