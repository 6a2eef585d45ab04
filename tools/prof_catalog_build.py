import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import CatalogPosterior, fit_stars_gpu
bands = ["G", "BP", "RP"]
ic = ia.synthetic_track(bands=bands)
cat, _ = ia.synthetic_catalog(ic, 400_000, bands=bands, seed=7, mag_unc=0.01)
fit_stars_gpu(cat, ic, np.arange(64), nwalkers=32, nburn=5, niter=5)
idx = np.arange(200_000)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    cols, template = CatalogPosterior.build_columns(cat, ic, indices=idx)
    t1 = time.perf_counter()
    post = CatalogPosterior(ic, _columns=cols, _template=template)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    post.close()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("build_columns %.2f ms  create %.2f ms  close %.2f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
pr = cProfile.Profile(); pr.enable()
cols, template = CatalogPosterior.build_columns(cat, ic, indices=idx)
post = CatalogPosterior(ic, _columns=cols, _template=template)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
