"""End-to-end wall-clock of the reference's default single-star fit (300 walkers, 200 burn-in + 100 stored steps,
start points drawn from the prior) and of the pieces around the sampler."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ic, mod = bench.build_model()
mod.lnpost([1.0, 355.0, 0.0, 100.0, 0.1])
def T(label, f, n=3):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("%-52s %.1f ms" % (label, min(ts) * 1e3), flush=True)
    return r
T("fit_mcmc(300 walkers, 200 + 100 steps), p0 from prior", lambda: mod.fit_mcmc(nwalkers=300, nburn=200, niter=100, seed=1))
T("fit_mcmc(..., p0=truth)", lambda: mod.fit_mcmc(nwalkers=300, nburn=200, niter=100, seed=1, p0=[1.0, 355.0, 0.0, 100.0, 0.1]))
T("emcee_p0(300)", lambda: mod.emcee_p0(300, rng=np.random.default_rng(0)))
def samples():
    mod._samples = None
    return mod.samples
T("samples DataFrame (30 000 rows, all derived columns)", samples)
pr = cProfile.Profile(); pr.enable(); mod.fit_mcmc(nwalkers=300, nburn=200, niter=100, seed=1); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
