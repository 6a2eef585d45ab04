#!/usr/bin/env python
"""A/B timing of the library's main kernels in one process (pick the library with ISOCHRONES_AMD_LIB):
cfg 2 / cfg 3 batches (rotating over 8 distinct batches), the cfg 4 sampler, a catalog fit.  One JSON line.

    python tools/ab_kernels.py [--cases cfg2,cfg3,cfg4,cfg5,cfg5ref,generic] [--reps 100] [--stars 400000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="cfg2,cfg3,cfg4,cfg5")
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--stars", type=int, default=400_000)
    ap.add_argument("--label", default=os.environ.get("ISOCHRONES_AMD_LIB", "default"))
    args = ap.parse_args()
    import torch
    import bench
    import bench_configs
    import isochrones_amd as ia
    from isochrones_amd import device as dev
    cases = args.cases.split(",")
    out = {"label": args.label, "path": os.environ.get("ISOCHRONES_AMD_PATH", "auto")}
    stream = dev.stream_ptr(0)
    nb = 8

    def timed(handle, hosts):
        r = bench.Rotation(handle, hosts, stream)
        r.run(10)
        ts = [r.run(args.reps) for _ in range(3)]
        return {"ms_min": min(ts), "ms_all": ts}

    if "cfg2" in cases or "cfg4" in cases or "generic" in cases:
        ic, mod = bench.build_model()
        mod.lnpost(bench.make_samples(np.random.default_rng(0), 4096, "prior"))
        if "cfg2" in cases:
            for wl in ("prior_valid", "posterior"):
                out["cfg2/" + wl] = timed(mod.handle(0), [bench.make_samples(np.random.default_rng(12345 + b), args.n, wl) for b in range(nb)])
        if "cfg4" in cases:
            from isochrones_amd.sampler import FusedEnsembleSampler
            truth = np.array([1.0, 355.0, 0.0, 100.0, 0.1])
            p0 = truth + np.array([0.01, 2.0, 0.02, 1.0, 0.02]) * np.random.default_rng(1).standard_normal((256, 5))
            p0[:, 4] = np.abs(p0[:, 4])
            fs = FusedEnsembleSampler(mod, 256, seed=2)
            fs.run_mcmc(p0, 50, store=False)
            walls = []
            for _ in range(4):
                fs.reset()
                torch.cuda.synchronize()
                t = time.perf_counter()
                fs.run_mcmc(p0, 5000, store=True)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t)
            out["cfg4"] = {"us_per_step": min(walls) / 5000 * 1e6, "chain_digest": float(fs._chain.double().sum())}
            fs.close()
        del mod, ic
    if "cfg3" in cases:
        ic3, mod3 = bench_configs.cfg3_model()
        mod3.lnpost(bench_configs.cfg3_samples(4096, "prior"))
        for wl in ("prior_valid", "posterior"):
            out["cfg3/" + wl] = timed(mod3.handle(0), [bench_configs.cfg3_samples(args.n, wl, seed=3 + 17 * b) for b in range(nb)])
        del mod3, ic3
    if "cfg5" in cases:
        from isochrones_amd.catalog import fit_stars_gpu
        bands = ["G", "BP", "RP"]
        ic = ia.synthetic_track(bands=bands)
        warm, _ = ia.synthetic_catalog(ic, 64, bands=bands, seed=1, mag_unc=0.01)
        fit_stars_gpu(warm, ic, np.arange(64), nwalkers=32, nburn=5, niter=5)
        for n_stars in (10_000, args.stars):
            cat, _ = ia.synthetic_catalog(ic, n_stars, bands=bands, seed=7, mag_unc=0.01)
            walls, tm = [], {}
            for k in range(3):
                tm = {}
                torch.cuda.synchronize()
                t = time.perf_counter()
                rows = fit_stars_gpu(cat, ic, np.arange(n_stars), nwalkers=32, nburn=150, niter=100, seed=11, timings=tm)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t)
            out["cfg5/%d" % n_stars] = {"wall_s_min": min(walls), "walls": walls, "breakdown_last": {k: round(v, 4) for k, v in tm.items()},
                                        "rows_digest": float(np.nansum(rows[:, :15]))}
    if "cfg5ref" in cases:
        # the reference's batch_starfit workload (bench.py catalog.reference_shape): isochrone parametrisation, 300 x (200 + 100)
        from isochrones_amd.catalog import fit_stars_gpu
        bands = ["G", "BP", "RP"]
        ic = ia.synthetic_isochrone(bands=bands)
        warm, _ = ia.synthetic_catalog(ic, 64, bands=bands, seed=1, mag_unc=0.01)
        fit_stars_gpu(warm, ic, np.arange(64), nwalkers=300, nburn=5, niter=5)
        for n_stars in (1_250, 10_000):
            cat, _ = ia.synthetic_catalog(ic, n_stars, bands=bands, seed=7, mag_unc=0.01)
            walls, tm = [], {}
            for k in range(3):
                tm = {}
                torch.cuda.synchronize()
                t = time.perf_counter()
                rows = fit_stars_gpu(cat, ic, np.arange(n_stars), nwalkers=300, nburn=200, niter=100, seed=11, timings=tm)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t)
            out["cfg5ref/%d" % n_stars] = {"wall_s_min": min(walls), "walls": walls, "breakdown_last": {k: round(v, 4) for k, v in tm.items()},
                                           "rows_digest": float(np.nansum(rows[:, :15]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
