import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU-side libraries are cheap to (re)build; the HIP library is built by
    # __graft_entry__.build() beforehand and is never built inside a test.
    if not os.path.exists(os.path.join(REPO, "oracle", "_build", "liboracle_a.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle"), "all"], check=True)
    if not os.path.exists(os.path.join(REPO, "raytracer_amd", "lib", "librtamd_host.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "raytracer_amd", "csrc"), "host"], check=True)


@pytest.fixture(scope="session")
def oracle_a():
    from oracles import OracleA
    a = OracleA()
    a.set_flush_denormals(1)
    return a


@pytest.fixture(scope="session")
def tracer():
    """One HIP context for the whole GPU session (fails loudly without GPU / library)."""
    from raytracer_amd.renderer import HipRayTracer
    t = HipRayTracer(0)
    yield t
    t.close()
