import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import synthetic_catalog, fit_stars_gpu
ic = ia.get_ichrone("mist", bands=["G", "BP", "RP"], tracks=True)
cat, _ = synthetic_catalog(ic, 10000, bands=["G", "BP", "RP"], seed=3, mag_unc=0.01)
idx = np.arange(10000)
fit_stars_gpu(cat, ic, idx[:500], nwalkers=32, nburn=20, niter=20)       # warm-up
torch.cuda.synchronize()
tm = {}
t = time.perf_counter()
fit_stars_gpu(cat, ic, idx, nwalkers=32, nburn=150, niter=100, timings=tm)
torch.cuda.synchronize()
print("wall", time.perf_counter() - t, tm)
pr = cProfile.Profile()
pr.enable()
fit_stars_gpu(cat, ic, idx, nwalkers=32, nburn=150, niter=100)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
