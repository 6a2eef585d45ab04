"""A/B of the two fused sampler kernels (step-wise launches vs one persistent launch) over
catalog sizes; prints one JSON line per case.  Run on the GPU box: python tools/sampler_modes.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import isochrones_amd as ia
    from isochrones_amd.catalog import CatalogPosterior, initial_positions, synthetic_catalog
    from isochrones_amd.sampler import FusedEnsembleSampler
    ic = ia.get_ichrone("mist", bands=["G", "BP", "RP"], tracks=True)
    cases = [(1, 256, 2000), (1, 64, 2000), (16, 32, 1000), (256, 32, 500), (1024, 32, 250), (2048, 32, 250),
             (4096, 32, 250), (6000, 32, 250), (8192, 32, 250), (10000, 32, 250), (12288, 32, 250), (16384, 32, 250),
             (1024, 128, 250)]
    cat, _ = synthetic_catalog(ic, 16384, bands=["G", "BP", "RP"], seed=3, mag_unc=0.01)
    for S, W, nsteps in cases:
        post = CatalogPosterior.from_catalog(cat, ic, N=1, indices=np.arange(S))
        pos, lnp, failed = initial_positions(post, W, rng_seed=1)
        good = ~failed
        if bool(failed.any()):
            src = int(torch.nonzero(good)[0])
            pos[failed] = pos[src]
            lnp[failed] = lnp[src]
        out = {"stars": S, "walkers": W, "nsteps": nsteps}
        for mode in ("stepwise", "persistent", "auto"):
            os.environ["ISOCHRONES_AMD_SAMPLER"] = mode
            fs = FusedEnsembleSampler(post, W, seed=2)
            fs.run_mcmc(pos, 10, lnprob0=lnp, store=False)
            torch.cuda.synchronize()
            t = time.perf_counter()
            fs.run_mcmc(pos, nsteps, lnprob0=lnp, store=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            out[mode + "_us_per_step"] = dt / nsteps * 1e6
            out[mode + "_lnpost_per_s"] = S * W * nsteps / dt
            fs.close()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
