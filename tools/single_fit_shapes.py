#!/usr/bin/env python
"""Time per step of one star's ensemble fit (persistent single-model sampler, FusedEnsembleSampler) by model shape:
(kind, stars in the system, bands).  A/B of a library variant: ISOCHRONES_AMD_LIB=... python tools/single_fit_shapes.py
    python tools/single_fit_shapes.py [walkers=256] [steps=2000] [repeats=5]
One JSON line per shape; the chain's checksum is printed so that two libraries can be compared for identical chains."""
import json, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import isochrones_amd as ia  # noqa: E402
from isochrones_amd.sampler import FusedEnsembleSampler  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
SHAPES = [tuple(int(x) if x.isdigit() else x for x in t.split(":")) for t in os.environ["SHAPES"].split(",")] if os.environ.get("SHAPES") else [("track", 1, nb) for nb in (3, 4, 5, 6, 8, 10, 12)] + [("iso", 1, nb) for nb in (3, 5, 8)] + [("iso", 2, 6), ("iso", 3, 9)]


def main():
    for kind, ns, nb in SHAPES:
        bands = list(ia.grids.KNOWN_BANDS[:nb])
        ic = ia.synthetic_track(bands=bands) if kind == "track" else ia.synthetic_isochrone(bands=bands)
        cat, truth = ia.synthetic_catalog(ic, 1, bands=bands, seed=5, mag_unc=0.02, with_parallax=True)
        mod = cat.model(0, ic, N=ns)
        from isochrones_amd.catalog import CatalogPosterior, initial_positions
        post = CatalogPosterior.from_catalog(cat, ic, N=ns)
        best, lnp, failed = initial_positions(post, W, rng_seed=3, oversample=8, max_tries=4)
        p0 = best[0].cpu().numpy()
        post.close()
        fs = FusedEnsembleSampler(mod, W, seed=11)
        times = []
        crc = None
        for r in range(R):
            fs.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pos, lnp_end = fs.run_mcmc(p0, T, store=False)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            if crc is None:
                crc = zlib.crc32(lnp_end.cpu().numpy().tobytes())
        fs.close()
        print(json.dumps(dict(kind=kind, n_stars=ns, n_bands=nb, walkers=W, steps=T, us_per_step=1e6 * float(np.median(times)) / T,
                              best_us_per_step=1e6 * min(times) / T, lnprob_crc=crc, lib=os.environ.get("ISOCHRONES_AMD_LIB", "default"))), flush=True)
        ic.release()


if __name__ == "__main__":
    main()
