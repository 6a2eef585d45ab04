#!/usr/bin/env python
"""How much of a catalog fit's sampling time is lane packing?  The same 10^4-star isochrone catalog fitted with 256 walkers
(128 moves per half-step: two ensembles fill a 256-lane workgroup exactly) and with 300 (150 moves: one ensemble per
workgroup, 150 of 192-256 lanes busy): nanoseconds per move of the burn-in + sampling phases.

    python tools/walker_packing_probe.py [--stars 10000]
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
    ap.add_argument("--stars", type=int, default=10_000)
    ap.add_argument("--walkers", default="128,256,300,384,512")
    args = ap.parse_args()
    import torch
    import isochrones_amd as ia
    from isochrones_amd.catalog import fit_stars_gpu
    bands = ["G", "BP", "RP"]
    ic = ia.synthetic_isochrone(bands=bands)
    cat, _ = ia.synthetic_catalog(ic, args.stars, bands=bands, seed=7, mag_unc=0.01)
    for W in [int(w) for w in args.walkers.split(",")]:
        fit_stars_gpu(cat, ic, np.arange(64), nwalkers=W, nburn=5, niter=5)
        best = None
        for _ in range(3):
            tm = {}
            torch.cuda.synchronize()
            t = time.perf_counter()
            fit_stars_gpu(cat, ic, np.arange(args.stars), nwalkers=W, nburn=200, niter=100, seed=11, timings=tm)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t
            if best is None or wall < best[0]:
                best = (wall, tm)
        wall, tm = best
        moves = args.stars * W * 300
        print(json.dumps({"walkers": W, "stars": args.stars, "wall_ms": wall * 1e3,
                          "ns_per_move_burn_and_sample": (tm["burn_in"] + tm["sampling"]) / moves * 1e9,
                          "breakdown": {k: round(v, 4) for k, v in tm.items()}}))


if __name__ == "__main__":
    main()
