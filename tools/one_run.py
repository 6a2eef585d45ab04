#!/usr/bin/env python
"""One short persistent single-model run of the (isochrone, 3 stars, 9 bands) shape with default priors - the run the round-3
source state always gets wrong - for a debugger session (rocgdb --args python tools/one_run.py).  16 walkers: the 8 moves
of a half-step are lanes 0-7 of wave 0."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import isochrones_amd as ia
from isochrones_amd.sampler import FusedEnsembleSampler

N, nb, W = 3, 9, 16
rng = np.random.default_rng(3)
ages = ia.grids.mist_log_ages()[60::2]
bands = ia.grids.DEFAULT_BANDS[:nb]
ic = ia.synthetic_isochrone(bands=bands, ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                            eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))
truth = np.array([380.0, 330.0, 300.0, 9.6, -0.1, 300.0, 0.1])
mags = ic.interp_mag([truth[0], *truth[N:]], list(bands))[3]
obs = {b: (float(mags[j]) - 0.3, 0.02) for j, b in enumerate(bands)}
mod = ia.BasicStarModel(ic, N=N, parallax=(1000 / 300.0, 0.05), feh=(-0.1, 0.1), **obs)
p0 = truth + np.array([1.0] * N + [0.01, 0.01, 1.0, 0.01]) * rng.standard_normal((W, N + 4))
p0[:, :N] = -np.sort(-p0[:, :N], axis=1)
p0[:, -1] = np.abs(p0[:, -1])
lnp0 = mod.lnpost(torch.as_tensor(p0, device="cuda"))
print("start lnpost", lnp0[:4].tolist(), flush=True)
os.environ["ISOCHRONES_AMD_SAMPLER"] = "persistent"
fs = FusedEnsembleSampler(mod, W, seed=9)
fs.run_mcmc(p0, int(os.environ.get("ONE_RUN_STEPS", 3)), lnprob0=lnp0, store=True)
torch.cuda.synchronize()
print("acceptance", float(fs.acceptance_fraction.mean()), "lnprob last", fs._lnprob[-1, :4].tolist(), flush=True)
