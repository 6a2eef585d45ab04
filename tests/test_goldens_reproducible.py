"""Wherever the reference checkout is present (/root/reference: the authoring container, not the GPU box): regenerate
every fixture with oracle/make_golden.py - i.e. run the reference itself - into a scratch directory and require the
result to be byte-identical to tests/golden/.  This is the pin that ties the committed vectors to the reference
(ISO_CHECK_GOLDENS=0 skips it)."""
import filecmp
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("ISO_CHECK_GOLDENS") == "0" or not os.path.isdir("/root/reference/isochrones"),
                    reason="needs the reference checkout at /root/reference")
def test_make_golden_reproduces_the_committed_fixtures(tmp_path):
    env = dict(os.environ, ISO_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1", PYTHONHASHSEED="12345")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stderr[-3000:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    kept = sorted(f for f in os.listdir(os.path.join(ROOT, "tests", "golden")) if f.endswith(".npz"))
    assert made == kept
    match, mismatch, errors = filecmp.cmpfiles(str(tmp_path), os.path.join(ROOT, "tests", "golden"), made, shallow=False)
    assert not mismatch and not errors, (mismatch, errors)
    # the $ISOCHRONES tree the reference's grid classes write (full_grid / array_grid caches) + the exported frames
    def walk(top):
        return sorted(os.path.relpath(os.path.join(d, f), top) for d, _, fs in os.walk(top) for f in fs)
    a, b = os.path.join(str(tmp_path), "isochrones_tree"), os.path.join(ROOT, "tests", "golden", "isochrones_tree")
    assert walk(a) == walk(b) and len(walk(a)) >= 7
    match, mismatch, errors = filecmp.cmpfiles(a, b, walk(a), shallow=False)
    assert not mismatch and not errors, (mismatch, errors)
