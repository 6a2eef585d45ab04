#!/usr/bin/env python
"""Per-run record for the hunt: random models of a pinned shape (tests/soak/soak.py's generator), each fitted for a few steps
by the persistent and by the step-wise sampler kernel from the same start (chains are bit-identical when both are right);
one JSON line per run with everything the run is made of - table sizes, observations, keywords, ensemble, start statistics -
and whether the two chains differ.  Offline: which property of a run decides?
    ISOCHRONES_AMD_LIB=variants/libs/libiso_hip_r987.so SOAK_KIND=iso SOAK_NSTARS=3 SOAK_NB=9 python tools/probe_runs.py [seconds] [seed]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import isochrones_amd as ia  # noqa
from isochrones_amd._cabi import IsoError
from isochrones_amd.sampler import FusedEnsembleSampler
from tests.soak import soak, soak_sampler


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0 = time.time()
    while time.time() - t0 < budget:
        cfg, ic, mod, axes, lo, hi = soak.build(rng)
        W = int(rng.choice([4, 8, 16, 30, 64, 100, 256]))
        a = float(rng.choice([1.3, 2.0, 3.0]))
        T = int(rng.integers(3, 10))
        ball = bool(rng.random() < 0.6)
        sseed = int(rng.integers(0, 2 ** 40))
        p0 = soak_sampler.start_points(rng, mod, lo, hi, W, ball)
        if p0 is None:
            ic.release(); continue
        lnp0 = mod.lnpost(torch.as_tensor(p0, device="cuda")).cpu().numpy()
        out = {}
        try:
            for mode in ("stepwise", "persistent"):
                os.environ["ISOCHRONES_AMD_SAMPLER"] = mode
                fs = FusedEnsembleSampler(mod, W, a=a, seed=sseed)
                fs.run_mcmc(p0, T, lnprob0=lnp0, store=True)
                out[mode] = (fs.chain_steps.cpu().numpy().copy(), fs._lnprob.cpu().numpy().copy(), float(fs.acceptance_fraction.mean()))
                fs.close()
        except IsoError:
            ic.release(); continue
        same = bool(np.array_equal(out["stepwise"][0], out["persistent"][0]) and np.array_equal(out["stepwise"][1], out["persistent"][1]))
        mi, bi = ic.model_grid.interp, ic.bc_grid.interp
        desc = mod.model_desc()
        rec = dict(bad=not same, acc_stepwise=out["stepwise"][2], acc_persistent=out["persistent"][2], W=W, a=a, T=T, ball=ball,
                   seed_lo=sseed & 0xFFFFFFFF, seed_hi=sseed >> 32,
                   n_age=int(mi.index_columns[0].size), n_feh=int(mi.index_columns[1].size), n_eep=int(mi.index_columns[2].size),
                   eep0=float(mi.index_columns[2][0]), uniform_eep=cfg["uniform_eep"],
                   has_Teff="Teff" in cfg["obs"], has_logg="logg" in cfg["obs"], has_feh="feh" in cfg["obs"],
                   has_plx="parallax" in cfg["obs"], plx=float(desc.plx_val) if desc.has_parallax else 0.0,
                   has_numax="nu_max" in cfg["obs"], has_dnu="delta_nu" in cfg["obs"],
                   kw=sorted(cfg["kw"]), priors=sorted(cfg["priors"]),
                   maxAV=float(cfg["kw"].get("maxAV", -1)), max_distance=float(cfg["kw"].get("max_distance", -1)),
                   halo=float(cfg["kw"].get("halo_fraction", -1)),
                   lnp0_min=float(lnp0.min()), lnp0_max=float(lnp0.max()),
                   p0_mean=[float(v) for v in p0.mean(axis=0)], p0_std=[float(v) for v in p0.std(axis=0)],
                   mag_val=[float(desc.mag_val[j]) for j in range(desc.n_bands)], mag_unc=[float(desc.mag_unc[j]) for j in range(desc.n_bands)],
                   bound_lo=[float(desc.bound_lo[j]) for j in range(7)], bound_hi=[float(desc.bound_hi[j]) for j in range(7)])
        print(json.dumps(rec), flush=True)
        ic.release()


if __name__ == "__main__":
    main()
