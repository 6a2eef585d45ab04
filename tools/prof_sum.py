import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import synthetic_catalog, CatalogPosterior, initial_positions
from isochrones_amd.sampler import FusedEnsembleSampler
ic = ia.get_ichrone("mist", bands=["G", "BP", "RP"], tracks=True)
S = 200000
cat, _ = synthetic_catalog(ic, S, bands=["G", "BP", "RP"], seed=3, mag_unc=0.01)
post = CatalogPosterior.from_catalog(cat, ic, N=1)
pos, lnp, failed = initial_positions(post, 32, rng_seed=0)
s = FusedEnsembleSampler(post, 32, seed=1)
pos, lnp = s.run_mcmc(pos, 20, lnprob0=lnp, store=False); s.reset()
def T(label, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); print("%-28s %.1f ms" % (label, (time.perf_counter() - t) * 1e3), flush=True); return r
T("run_mcmc 100 store", lambda: s.run_mcmc(pos, 100, lnprob0=lnp, store=True))
for rep in range(2):
    q = T("quantiles", lambda: s.quantiles((0.5, 0.16, 0.84)))
    T("lnprob max", lambda: s._lnprob.amax(dim=0).view(S, 32).amax(dim=1))
    T("acc frac", lambda: s.acceptance_fraction.mean(dim=1))
    rows = T("rows alloc+fill", lambda: torch.empty(S, 18, dtype=torch.float64, device="cuda").copy_(torch.cat([q.reshape(S, 15), q.reshape(S, 15)[:, :3]], dim=1)))
    T("to host", lambda: rows.cpu().numpy())
T("post.close", lambda: post.close())
T("del sampler", lambda: s.close())
