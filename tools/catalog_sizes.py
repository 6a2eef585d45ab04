#!/usr/bin/env python
"""Catalog fit (BASELINE configs[4]: 32 walkers x (150 + 100) steps, G / BP / RP) over catalog sizes, with the phases of
fit_stars_gpu timed: what one GPU of an N-GPU node sees of a 10^4-star catalog is 10^4 / N stars.
    python tools/catalog_sizes.py [--sizes 1250,2500,5000,10000] [--groups 0,4,8,16] [--start kernel,torch]
ISOCHRONES_AMD_PERSIST_GROUP (ensembles per workgroup of the persistent sampler kernel; 0 = the library's choice) and
ISOCHRONES_AMD_START are swept; one JSON line per (size, group, start method): best of 3 runs."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import fit_stars_gpu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="313,625,1250,2500,5000,10000")
    ap.add_argument("--groups", default="0")
    ap.add_argument("--start", default="kernel")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--bands", default="G,BP,RP")
    args = ap.parse_args()
    bands = args.bands.split(",")
    ic = ia.synthetic_track(bands=bands)
    warm, _ = ia.synthetic_catalog(ic, 64, bands=bands, seed=1, mag_unc=0.01)
    fit_stars_gpu(warm, ic, np.arange(64), nwalkers=32, nburn=5, niter=5)
    for n in [int(v) for v in args.sizes.split(",")]:
        cat, _ = ia.synthetic_catalog(ic, n, bands=bands, seed=7, mag_unc=0.01)
        for start in args.start.split(","):
            os.environ["ISOCHRONES_AMD_START"] = start
            for g in [int(v) for v in args.groups.split(",")]:
                if g:
                    os.environ["ISOCHRONES_AMD_PERSIST_GROUP"] = str(g)
                else:
                    os.environ.pop("ISOCHRONES_AMD_PERSIST_GROUP", None)
                best = None
                for _ in range(args.reps):
                    tm = {}
                    torch.cuda.synchronize(); t = time.perf_counter()
                    rows = fit_stars_gpu(cat, ic, np.arange(n), nwalkers=32, nburn=150, niter=100, seed=11, timings=tm)
                    torch.cuda.synchronize(); w = time.perf_counter() - t
                    if best is None or w < best[0]:
                        best = (w, tm)
                w, tm = best
                print(json.dumps({"stars": n, "start": start, "group": g, "wall_ms": round(w * 1e3, 3), "stars_per_s": round(n / w),
                                  "ok": float(np.mean(rows[:, -1] == 1)), "phases_ms": {k: round(v * 1e3, 3) for k, v in tm.items()}}), flush=True)


if __name__ == "__main__":
    main()
