"""The host layer (OBJ / MTL parsers, serial and parallel BVH builders) rebuilt from source under
AddressSanitizer + UBSan and under ThreadSanitizer (tools/sanitize/).  CPU only; the driver also
re-checks that the parallel builder's output equals the serial one's on four kinds of input."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["asan", "tsan"])
def test_host_layer_under_sanitizers(mode, tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run([os.path.join(REPO, "tools", "sanitize", "run.sh"), mode, "3000"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])
