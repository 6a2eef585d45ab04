"""Where the host time of a SMALL catalog fit goes (1 250 stars, 32 walkers x 250 steps: one GPU's share of eight of the
10^4-star leg): wall per fit and a cProfile of 40 fits.  python tools/prof_small_catalog.py [stars=1250]"""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import fit_stars_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
bands = ["G", "BP", "RP"]
ic = ia.synthetic_track(bands=bands)
cat, _ = ia.synthetic_catalog(ic, 10_000, bands=bands, seed=7, mag_unc=0.01)
idx = np.array([i for i in range(10_000) if (i + 1) % 8 == 0])[:n]
for _ in range(5):
    fit_stars_gpu(cat, ic, idx, nwalkers=32, nburn=150, niter=100, seed=11)
walls = []
for _ in range(20):
    torch.cuda.synchronize(); t = time.perf_counter()
    fit_stars_gpu(cat, ic, idx, nwalkers=32, nburn=150, niter=100, seed=11)
    torch.cuda.synchronize(); walls.append(time.perf_counter() - t)
print("stars %d: median %.3f ms, best %.3f ms" % (idx.size, 1e3 * float(np.median(walls)), 1e3 * min(walls)))
pr = cProfile.Profile(); pr.enable()
for _ in range(40):
    fit_stars_gpu(cat, ic, idx, nwalkers=32, nburn=150, niter=100, seed=11)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
