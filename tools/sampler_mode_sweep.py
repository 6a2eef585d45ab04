"""Catalog sampler: persistent kernel (workgroups own their ensembles for all iterations; beyond the chip's capacity they
run in rounds) against the launch-per-half-step kernel, by catalog size.  32 walkers, 150 + 100 steps, 3 bands."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import fit_stars_gpu
KIND = os.environ.get("SWEEP_KIND", "track")          # track | iso
N = int(os.environ.get("SWEEP_N", 1))                   # stars per system (iso only)
W = int(os.environ.get("SWEEP_W", 32))
bands = list(ia.grids.DEFAULT_BANDS[:int(os.environ.get("SWEEP_NB", 3))]) if "SWEEP_NB" in os.environ else ["G", "BP", "RP"]
ic = ia.synthetic_track(bands=bands) if KIND == "track" else ia.synthetic_isochrone(bands=bands)
warm, _ = ia.synthetic_catalog(ic, 64, bands=bands, seed=1, mag_unc=0.01)
fit_stars_gpu(warm, ic, np.arange(64), N=N, nwalkers=W, nburn=5, niter=5)
sizes = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else "2000,5000,10000,15000,20000,30000,50000,100000,200000,400000".split(","))]
big, _ = ia.synthetic_catalog(ic, max(sizes), bands=bands, seed=7, mag_unc=0.01)
for n in sizes:
    line = "%s N=%d nb=%d W=%d %7d stars:" % (KIND, N, len(bands), W, n)
    for mode in ("stepwise", "persistent", "auto"):
        os.environ["ISOCHRONES_AMD_SAMPLER"] = mode
        best = None
        for rep in range(3):
            tm = {}
            fit_stars_gpu(big, ic, np.arange(n), N=N, nwalkers=W, nburn=150, niter=100, seed=11, timings=tm)
            t = tm["burn_in"] + tm["sampling"]
            best = t if best is None else min(best, t)
        line += "  %s %.4f s" % (mode, best)
    print(line, flush=True)
