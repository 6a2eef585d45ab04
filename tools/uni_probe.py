#!/usr/bin/env python
"""Single-model persistent sampler (the UNI kernel) against the batch kernel at the stored points, over model shapes:
python tools/uni_probe.py  ->  one line per (kind, stars, bands): number of stored lnprob values that differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import isochrones_amd as ia
    from isochrones_amd.sampler import FusedEnsembleSampler
    rng = np.random.default_rng(3)
    ages = ia.grids.mist_log_ages()[60::2]
    for N in (1, 2, 3):
        for nb in (1, 3, 6, 8, 9, 10, 12):
            bands = ia.grids.DEFAULT_BANDS[:min(nb, len(ia.grids.DEFAULT_BANDS))]
            if len(bands) < nb:
                continue
            ic = ia.synthetic_isochrone(bands=bands, ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                                        eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))
            truth = np.array([380.0, 330.0, 300.0][:N] + [9.6, -0.1, 300.0, 0.1])
            mags = ic.interp_mag([truth[0], *truth[N:]], list(bands))[3]
            obs = {b: (float(mags[j]) - 0.3 * (N > 1), 0.02) for j, b in enumerate(bands)}
            mod = ia.BasicStarModel(ic, N=N, parallax=(1000 / 300.0, 0.05), feh=(-0.1, 0.1), **obs)
            for W in (16, 64, 256):
                p0 = truth + np.array([1.0] * N + [0.01, 0.01, 1.0, 0.01]) * rng.standard_normal((W, N + 4))
                p0[:, :N] = -np.sort(-p0[:, :N], axis=1)
                p0[:, -1] = np.abs(p0[:, -1])
                lnp0 = mod.lnpost(torch.as_tensor(p0, device="cuda"))
                if not bool(torch.isfinite(lnp0).all()):
                    print("start not finite", N, nb, W)
                    continue
                out = []
                for mode in ("stepwise", "persistent"):
                    os.environ["ISOCHRONES_AMD_SAMPLER"] = mode
                    fs = FusedEnsembleSampler(mod, W, seed=9)
                    fs.run_mcmc(p0, 12, lnprob0=lnp0, store=True)
                    ch = fs.chain.clone()
                    lp = fs.lnprobability.clone()
                    ref = mod.lnpost(ch.reshape(-1, N + 4)).reshape(lp.shape)
                    bad = int((~torch.isclose(ref, lp, rtol=1e-9, atol=1e-9)).sum())
                    out.append((mode, bad, float((ref - lp).abs().max())))
                    fs.close()
                print("iso N=%d nb=%2d W=%3d  " % (N, nb, W) + "  ".join("%s: %d differ (max %.3g)" % o for o in out), flush=True)
            ic.release()


if __name__ == "__main__":
    main()
