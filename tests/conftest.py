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
def _tracer_session():
    """One HIP context for the whole GPU session (fails loudly without GPU / library)."""
    from raytracer_amd.renderer import HipRayTracer
    t = HipRayTracer(0)
    yield t
    t.close()


@pytest.fixture
def tracer(_tracer_session):
    """The session's context with the state a test may rely on without setting it: the shader's two albedo slots hold the
    white texel (a scene without texture coordinates samples texel (0,0) of whatever the previous test left there), the
    persistent kernel, float texture filtering, the context's own stream.  Scene, materials and the other options are the test's business."""
    import numpy as np
    t = _tracer_session
    white = np.full((1, 1, 3), 255, np.uint8)
    t.set_stream(0)
    t.set_kernel_variant(0)
    t.set_option("texture_filter_8bit", 0)
    t.upload_texture(0, white)
    t.upload_texture(1, white)
    return t
