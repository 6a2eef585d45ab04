"""Latency of the scalar call forms of the public API (host value in, host value out, one launch + one
synchronise each); the reference publishes 7-12 us (DFInterpolator), 69 us (lnpost, single star)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ic, mod = bench.build_model()
p = [1.0, 355.0, 0.0, 100.0, 0.1]
def t(label, f, n=2000):
    f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    print("%-46s %.1f us" % (label, (time.perf_counter() - t0) / n * 1e6), flush=True)
t("mod.lnpost(p)", lambda: mod.lnpost(p))
t("mod.lnprior(p)", lambda: mod.lnprior(p))
t("ic.interp_value(p[:3], ['Teff','logg','age'])", lambda: ic.interp_value(p[:3], ["Teff", "logg", "age"]))
t("ic.interp_mag(p, ['V'])", lambda: ic.interp_mag(p, ["V"]))
t("ic.model_grid.interp([0.0,1.0,355.0], ['Teff'])", lambda: ic.model_grid.interp([0.0, 1.0, 355.0], ["Teff"]))
t("ic.get_eep(1.0, 9.6, 0.0)", lambda: ic.get_eep(1.0, 9.6, 0.0))
t("ic.mass(*p[:3])", lambda: ic.mass(*p[:3]))
