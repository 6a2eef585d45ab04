#!/usr/bin/env python
"""k_chain_quantiles_big on the chains it summarises in production: N stars of the reference-shape catalog (isochrone
parametrisation, 300 walkers x (200 + 100) steps, G / BP / RP) are fitted, then the 16 / 50 / 84 % summaries of the stored
chain (30 000 values per (star, parameter) pair) are timed on their own, and compared with numpy.percentile on a few stars.
    python tools/quantile_big_timing.py [--stars 10000] [--check 24] [--reps 7] [--label NAME]
One JSON line: ms per call, chain bytes, chain-passes-per-second figure, a digest of the summaries (equal across libraries)."""
import argparse, hashlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import CatalogPosterior, initial_positions
from isochrones_amd.sampler import FusedEnsembleSampler


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stars", type=int, default=10000)
    ap.add_argument("--walkers", type=int, default=300)
    ap.add_argument("--nburn", type=int, default=200)
    ap.add_argument("--niter", type=int, default=100)
    ap.add_argument("--check", type=int, default=24)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--label", default=os.path.basename(os.environ.get("ISOCHRONES_AMD_LIB", "default")))
    args = ap.parse_args()
    bands = ("G", "BP", "RP")
    ic = ia.synthetic_isochrone(bands=bands)
    cat, _ = ia.synthetic_catalog(ic, args.stars, bands=list(bands), seed=7, mag_unc=0.01)
    post = CatalogPosterior.from_catalog(cat, ic, indices=np.arange(args.stars))
    pos, lnp, failed = initial_positions(post, args.walkers, rng_seed=3)
    assert not bool(failed.any())
    s = FusedEnsembleSampler(post, args.walkers, seed=4)
    pos, lnp = s.run_mcmc(pos, args.nburn, lnprob0=lnp, store=False)
    s.reset()
    s.run_mcmc(pos, args.niter, lnprob0=lnp, store=True)
    torch.cuda.synchronize()
    q = s.quantiles((0.5, 0.16, 0.84))
    torch.cuda.synchronize()
    times = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        q = s.quantiles((0.5, 0.16, 0.84))
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    chain_bytes = s._chain.numel() * 8
    got = q.cpu().numpy()                                            # [S, D, 3]
    # numpy on the first stars: chain storage is [step][parameter][star * W + walker]
    n_chk = min(args.check, args.stars)
    W = args.walkers
    ch = s._chain[:, :, : n_chk * W].cpu().numpy()                   # [T, D, n_chk * W]
    T, D = ch.shape[0], ch.shape[1]
    flat = ch.reshape(T, D, n_chk, W).transpose(2, 1, 0, 3).reshape(n_chk, D, T * W)
    want = np.moveaxis(np.quantile(flat, [0.5, 0.16, 0.84], axis=2), 0, 2)
    same = bool(np.array_equal(got[:n_chk], want))
    ms = float(np.median(times))
    print(json.dumps({"label": args.label, "stars": args.stars, "values_per_pair": T * W, "ms": round(ms, 4), "ms_min": round(min(times), 4),
                      "chain_GB": round(chain_bytes / 1e9, 3), "chain_passes_TBs": round(chain_bytes / (ms * 1e-3) / 1e12, 3),
                      "bit_for_bit_numpy_first_%d" % n_chk: same, "digest": hashlib.sha256(got.tobytes()).hexdigest()[:16]}), flush=True)
    post.close()


if __name__ == "__main__":
    main()
