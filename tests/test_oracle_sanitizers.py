"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers).

oracle/iso_oracle.c is rebuilt with -fsanitize=address,undefined (oracle/Makefile: libiso_oracle_san.so) and the
golden-vector tests of the oracle - every interpolation, lnpost, tree and EEP entry point, NaN / out-of-grid /
upper-edge inputs included - are run against that build in a subprocess with libasan preloaded.  Any out-of-bounds
table read, misaligned access, signed overflow or invalid shift aborts the subprocess."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_golden_tests_pass_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan runtime here")
    lib = os.path.join(ROOT, "oracle", "libiso_oracle_san.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libiso_oracle_san.so"])
    env = dict(os.environ, ISO_ORACLE_LIB=lib, LD_PRELOAD=asan, PYTHONDONTWRITEBYTECODE="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # (the container-only tests that import the reference package pull in matplotlib & co., which do not survive an
    # LD_PRELOADed ASan runtime; they exercise the same oracle entry points as the committed cases)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-k", "not reference_itself",
           "tests/test_oracle_golden.py",
           "tests/test_tree_cpu.py::test_tree_structure_and_oracle_vs_reference",
           "tests/test_tree_cpu.py::test_keyword_tree_equals_basic_model_on_the_oracle"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "passed" in p.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # the sanitized build really was the one loaded
    probe = subprocess.run([sys.executable, "-c", "from oracle import oracle as o; o.lib(); print(o._LIBPATH); "
                            "print(open('/proc/self/maps').read().count('libiso_oracle_san.so') > 0)"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and probe.stdout.split()[-2:] == [lib, "True"], probe.stdout + probe.stderr
