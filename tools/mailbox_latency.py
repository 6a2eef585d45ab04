"""Where the time of a per-point callback goes: the Python wrapper, the ctypes call, the C entry point with the resident
mailbox wave (ISOCHRONES_AMD_MAILBOX=1, default) and with a launch per call (=0), for 1 / 5 / 64 / 128 rows."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from isochrones_amd import _cabi                # noqa: E402

ic, mod = bench.build_model()
lib = _cabi.lib()
h = mod.handle(0)
p = np.array([1.0, 355.0, 0.0, 100.0, 0.1])


def t(label, f, n=5000):
    for _ in range(50):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    us = (time.perf_counter() - t0) / n * 1e6
    print("%-58s %6.2f us" % (label, us), flush=True)
    return us


out = np.empty(1)
t("ctypes call of a trivial entry point (iso_model_n_params)", lambda: lib.iso_model_n_params(h))
for mb in ("1", "0"):
    os.environ["ISOCHRONES_AMD_MAILBOX"] = mb
    tag = "mailbox" if mb == "1" else "launch "
    t("%s  iso_lnpost_host(1 row), raw ctypes" % tag, lambda: lib.iso_lnpost_host(h, p.ctypes.data, 1, out.ctypes.data, None, None))
    t("%s  mod.lnpost(p)" % tag, lambda: mod.lnpost(p))
    for n in (5, 64, 128):
        rows = np.tile(p, (n, 1)) * (1 + 1e-3 * np.random.default_rng(n).standard_normal((n, 5)))
        o = np.empty(n)
        t("%s  iso_lnpost_host(%d rows), raw ctypes" % (tag, n), lambda: lib.iso_lnpost_host(h, rows.ctypes.data, n, o.ctypes.data, None, None), n=2000)
os.environ.pop("ISOCHRONES_AMD_MAILBOX")
for idle in ("20", "100"):
    os.environ["ISOCHRONES_AMD_MAILBOX_IDLE_US"] = idle
    mod2 = bench.build_model()[1]
    h2 = mod2.handle(0)
    # calls spaced further apart than the idle time: every call relaunches the wave
    def spaced():
        time.sleep(0.0005)
        t0 = time.perf_counter()
        lib.iso_lnpost_host(h2, p.ctypes.data, 1, out.ctypes.data, None, None)
        return time.perf_counter() - t0
    spaced()
    ts = [spaced() for _ in range(300)]
    print("idle %s us, calls 500 us apart: median call %.2f us" % (idle, 1e6 * float(np.median(ts))))
