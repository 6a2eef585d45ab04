#!/usr/bin/env python
"""Persistent single-model sampler (lane-per-sample gathers) against the step-wise kernel (cooperative gathers):
where do the stored lnprob / chain values differ, and by how much?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mod, p0, lnp0, W, n, mode):
    from isochrones_amd.sampler import FusedEnsembleSampler
    os.environ["ISOCHRONES_AMD_SAMPLER"] = mode
    fs = FusedEnsembleSampler(mod, W, seed=9)
    fs.run_mcmc(p0, n, lnprob0=lnp0, store=True)
    return fs.chain.clone().cpu().numpy(), fs.lnprobability.clone().cpu().numpy()


def main():
    import torch
    import isochrones_amd as ia
    from tests.test_gpu_catalog import _small_track, synthetic_catalog, CatalogPosterior
    from isochrones_amd.catalog import initial_positions
    for bands in (("G",), ("G", "BP", "RP")):
        ic = _small_track(bands)
        cat, truth = synthetic_catalog(ic, 5, bands=list(bands), seed=4, mag_unc=0.01)
        models = list(cat.iter_models(ic))
        post = CatalogPosterior(ic, models)
        W = 128
        pos, lnp, failed = initial_positions(post, W, rng_seed=2)
        a = run(models[2], pos[2], lnp[2], W, int(os.environ.get("PROBE_STEPS", "1")), "stepwise")
        b = run(models[2], pos[2], lnp[2], W, int(os.environ.get("PROBE_STEPS", "1")), "persistent")
        for name, x, y in (("chain", a[0], b[0]), ("lnprob", a[1], b[1])):
            d = x != y
            print(bands, name, "differing", int(d.sum()), "of", d.size)
            if d.any():
                idx = np.argwhere(d)[:6]
                for i in idx:
                    i = tuple(i)
                    print("   ", i, repr(float(x[i])), repr(float(y[i])), "rel", abs(x[i] - y[i]) / abs(x[i]))
        # the two evaluations of the same points through the batch kernel (cooperative) for reference
        pts = torch.as_tensor(b[0].reshape(-1, b[0].shape[-1]), device="cuda")
        ref = models[2].lnpost(pts).cpu().numpy().reshape(b[1].shape)
        print(bands, "persistent lnprob vs batch kernel at the stored points: differing", int((ref != b[1]).sum()),
              "; stepwise vs batch:", int((models[2].lnpost(torch.as_tensor(a[0].reshape(-1, a[0].shape[-1]), device='cuda')).cpu().numpy().reshape(a[1].shape) != a[1]).sum()))


if __name__ == "__main__":
    main()
