"""__graft_entry__: build() followed by smoke() in ONE interpreter (as a driver that imports the module once would call
them).  build() checks the exported symbols by loading the library; it has to load it the way every other caller does
(isochrones_amd._cabi.lib(): torch first, so the library binds to the HIP runtime torch bundles) - a bare ctypes.CDLL
before torch binds it to /opt/rocm's copy of libamdhip64, and the smoke() that follows then finds two runtimes in the
process and "no ROCm-capable device"."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_then_smoke_in_one_process():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "smoke ok" in r.stdout


def test_smoke_alone():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "smoke ok" in r.stdout
