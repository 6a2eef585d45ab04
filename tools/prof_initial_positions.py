"""Where the time of the catalog start-point search goes (2 x 10^5 stars, 32 walkers, 256 candidates per star)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import CatalogPosterior, initial_positions
bands = ["G", "BP", "RP"]
ic = ia.synthetic_track(bands=bands)
cat, _ = ia.synthetic_catalog(ic, 200_000, bands=bands, seed=7, mag_unc=0.01)
post = CatalogPosterior.from_catalog(cat, ic)
initial_positions(post, 32, rng_seed=0)
torch.cuda.synchronize()
for rep in range(2):
    t = time.perf_counter()
    initial_positions(post, 32, rng_seed=rep)
    torch.cuda.synchronize()
    print("initial_positions: %.2f ms" % ((time.perf_counter() - t) * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    initial_positions(post, 32, rng_seed=5)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
