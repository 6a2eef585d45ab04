"""Where the time of a large catalog fit goes (bench.py's catalog leg: 4 x 10^5 stars, 32 walkers x 250 steps)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import fit_stars_gpu
bands = ["G", "BP", "RP"]
ic = ia.synthetic_track(bands=bands)
warm, _ = ia.synthetic_catalog(ic, 64, bands=bands, seed=1, mag_unc=0.01)
fit_stars_gpu(warm, ic, np.arange(64), nwalkers=32, nburn=5, niter=5)
for n in (10_000, 400_000, 400_000):
    cat, _ = ia.synthetic_catalog(ic, n, bands=bands, seed=7, mag_unc=0.01)
    tm = {}
    torch.cuda.synchronize(); t = time.perf_counter()
    rows = fit_stars_gpu(cat, ic, np.arange(n), nwalkers=32, nburn=150, niter=100, seed=11, timings=tm)
    torch.cuda.synchronize(); w = time.perf_counter() - t
    print("%7d stars: %.3f s = %.3g stars/s  %s" % (n, w, n / w, {k: round(v, 4) for k, v in tm.items()}), flush=True)
