#!/usr/bin/env python
"""Library built -DISO_DEBUG_PROPOSALS -DISO_DEBUG_PARTS (variant source trees only): the persistent sampler stores, per
move, seven intermediates of its evaluation instead of the position - lnprior, lnlike after the spectroscopic terms, the
distance modulus, lnlike after the photometric terms, final lnlike, BC of the first band (primary), M_bol of the primary.
Prints them for runs whose lnpost came out NaN and for runs that are fine."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import isochrones_amd as ia  # noqa
from isochrones_amd._cabi import IsoError
from isochrones_amd.sampler import FusedEnsembleSampler
from tests import _fixtures as fx
from tests.soak import soak, soak_sampler

NAMES = ["lnprior", "lnl_spec", "dm", "lnl_phot", "lnl_all", "bc0_primary", "Mbol_primary"]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    os.environ["ISOCHRONES_AMD_SAMPLER"] = "auto"
    t0 = time.time()
    shown = {True: 0, False: 0}
    while time.time() - t0 < budget and min(shown.values()) < 4:
        cfg, ic, mod, axes, lo, hi = soak.build(rng)
        W = int(rng.choice([16, 64]))
        try:
            fs = FusedEnsembleSampler(mod, W, a=2.0, seed=int(rng.integers(0, 2 ** 40)))
        except IsoError:
            ic.release(); continue
        p0 = soak_sampler.start_points(rng, mod, lo, hi, W, True)
        if p0 is None:
            ic.release(); continue
        lnp0 = mod.lnpost(torch.as_tensor(p0, device="cuda")).cpu().numpy()
        fs.run_mcmc(p0, 2, lnprob0=lnp0, store=True)
        parts = fs.chain_steps.cpu().numpy()              # [T, W, 7]
        got = fs._lnprob.cpu().numpy()
        bad = bool(np.isnan(got).all())
        if shown[bad] < 4:
            shown[bad] += 1
            print("%s run  cfg %s" % ("BAD " if bad else "good", json.dumps(cfg)))
            for w in range(3):
                print("   walker %d lnpost %.10g : " % (w, got[0, w]) + ", ".join("%s=%.10g" % (n, v) for n, v in zip(NAMES, parts[0, w])))
        del fs
        ic.release()


if __name__ == "__main__":
    main()
