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
import bench_configs
tmod, tpars = bench_configs.tree_model_and_samples(16)
tp = [float(v) for v in tpars[0]]
t("tree model (resolved binary) lnpost(p)", lambda: tmod.lnpost(tp))
assert tmod.lnpost(tp) == float(tmod.lnpost(np.array([tp, tp]))[0])
import isochrones_amd as ia
iso_ic = ia.synthetic_isochrone(bands=("V",))
itm = ia.IsoTrackModel(iso_ic, ic, Teff=(5770, 100), logg=(4.5, 0.1), feh=(0.0, 0.15), V=(10.0, 0.05))
ip = [355.0, 1.0, 9.6, 0.0, 100.0, 0.1]
t("IsoTrackModel lnpost(p)", lambda: itm.lnpost(ip))
half = np.tile(np.array(p), (128, 1)) * (1 + 1e-3 * np.random.default_rng(0).standard_normal((128, 5)))
t("mod.lnpost(half ensemble [128, 5])", lambda: mod.lnpost(half))
os.environ["ISOCHRONES_AMD_HOST_SYNC"] = "1"          # A/B: the stream-synchronise completion of round 1
t("mod.lnpost(p), hipStreamSynchronize", lambda: mod.lnpost(p))
t("mod.lnpost(half), hipStreamSynchronize", lambda: mod.lnpost(half))
os.environ.pop("ISOCHRONES_AMD_HOST_SYNC")
assert mod.lnpost(p) == float(mod.lnpost(np.array([p]))[0])
# host-array path at benchmark size: where the time goes
import torch
big = bench.make_samples(np.random.default_rng(1), 1_000_000, "prior_valid")
def tm(label, f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    print("%-46s %.3f ms" % (label, (time.perf_counter() - t0) / n * 1e3), flush=True)
tm("mod.lnpost(numpy [1e6, 5]) -> numpy", lambda: mod.lnpost(big))
dv = torch.as_tensor(big, device="cuda")
tm("  H2D 40 MB pageable (torch.as_tensor)", lambda: torch.as_tensor(big, device="cuda"))
tm("  kernel on resident rows", lambda: mod.lnpost(dv))
res = mod.lnpost(dv)
tm("  D2H 8 MB (.cpu().numpy())", lambda: res.cpu().numpy())
pin = torch.empty(big.shape, dtype=torch.float64).pin_memory()
tm("  memcpy numpy -> pinned (1 thread)", lambda: pin.numpy().__setitem__(slice(None), big))
tm("  H2D 40 MB from pinned", lambda: dv.copy_(pin, non_blocking=True))
