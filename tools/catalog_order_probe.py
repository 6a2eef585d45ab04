"""Does the order of the stars matter to the catalog sampler?  Same 4 x 10^5-star catalog, fitted in file order and in
table order (stars sorted by the cell of their true parameters) - the second keeps the stars that run at the same time
in the same part of the tables."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
cat, truth = ia.synthetic_catalog(ic, n, bands=bands, seed=7, mag_unc=0.01)
t = truth.values            # mass, eep, feh, distance, AV
orders = {"file order": np.arange(n),
          "feh, mass, eep": np.lexsort((t[:, 1], t[:, 0], np.round(t[:, 2] / 0.25))),
          "mass, eep": np.lexsort((t[:, 1], np.round(np.log(t[:, 0]) * 20))),
          "eep, mass": np.lexsort((t[:, 0], np.round(t[:, 1] / 4)))}
for rep in range(2):
    for name, idx in orders.items():
        tm = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rows = fit_stars_gpu(cat, ic, idx, nwalkers=32, nburn=150, niter=100, seed=11, timings=tm)
        torch.cuda.synchronize(); w = time.perf_counter() - t0
        print("%-16s %.3f s = %.3g stars/s  %s" % (name, w, n / w, {k: round(v, 4) for k, v in tm.items()}), flush=True)
