#!/usr/bin/env python
"""Launch the library kernels of the benchmark workloads in a fixed order, `--launches` times each, and write
the order as a manifest - so that a rocprofv3 --pmc pass over this script can be split per workload even though
several workloads run the same kernel (tools/pmc_summarize.py walks the dispatches of a kernel in order).

    python tools/pmc_workload.py --manifest out.json [--launches 6] [--cases cfg2,cfg3,...]

cases: catalog_ref (reference-shape catalog sampler, 10^4 stars x 300 walkers), cfg2 (single star, 1 band: prior / prior_valid / posterior samples), cfg3 (binary, 6 bands + parallax: the same
three), generic (cfg2 model on the generic kernel), astero, tree (resolved binary, fast tree kernel), tree_generic (the same
tree on the generic tree kernel),
quantiles (chain summaries of a 10^4-star catalog, 32 walkers x 100 steps), sampler (the catalog sampler on 2 x 10^5 stars: step-wise and
persistent kernels).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--manifest", required=True)
    ap.add_argument("--launches", type=int, default=18)
    ap.add_argument("--cases", default="cfg2,cfg3,generic,astero,tree,quantiles,primitives,sampler")
    ap.add_argument("--n", type=int, default=1_000_000)
    args = ap.parse_args()
    import torch
    import bench
    import bench_configs
    import isochrones_amd as ia
    from isochrones_amd import _cabi, device as dev
    lib = _cabi.lib()
    manifest = []
    L, n = args.launches, args.n
    cases = args.cases.split(",")

    NB = 8      # distinct sample batches a workload's launches rotate over (no launch finds lines of the previous ones in the
                # 256 MiB Infinity Cache: a batch touches ~0.5-1.2 GB of table lines), as bench.py's timed steps do

    def lnpost_launches(label, kernel, mod, batches, bytes_per_eval):
        """`batches`: list of [n, D] host arrays; launch r evaluates batch r % len(batches)."""
        n_rows = batches[0].shape[0]
        pts = [torch.as_tensor(np.ascontiguousarray(b.T), device="cuda") for b in batches]
        outs = [torch.empty(n_rows, dtype=torch.float64, device="cuda") for _ in batches]
        p_arr = (C.c_void_p * len(pts))(*[t.data_ptr() for t in pts])
        o_arr = (C.c_void_p * len(pts))(*[t.data_ptr() for t in outs])
        ms = C.c_double()
        h = mod.handle(torch.cuda.current_device())
        torch.cuda.synchronize()
        _cabi.check(lib.iso_time_lnpost_rotating(h, p_arr, o_arr, len(pts), 1, n_rows, n_rows, L, dev.stream_ptr(), C.byref(ms)))
        torch.cuda.synchronize()
        manifest.append(dict(label=label, kernel=kernel, launches=L, skip=2, n=int(n_rows), ms_per_launch_profiled=ms.value,
                             distinct_batches=len(pts), algorithmic_bytes_per_launch=float(bytes_per_eval) * n_rows,
                             finite_fraction=float(torch.isfinite(outs[0]).double().mean())))

    if "cfg2" in cases or "generic" in cases:
        ic, mod = bench.build_model()
        samples = {wl: [bench.make_samples(np.random.default_rng((12345 if wl == "prior_valid" else 999) + b), n, wl) for b in range(NB)]
                   for wl in ("prior", "prior_valid", "posterior")}
        if "cfg2" in cases:
            mod.lnpost(samples["prior"][0][:4096])                      # builds the packs outside the counted launches
            for wl, batches in samples.items():
                lnpost_launches("cfg2/" + wl, "k_lnpost_fast<0, 1, 1, false, false>", mod, batches, 560)
            # the one-batch-repeated form of rounds 1-2 next to it (what the Infinity Cache hides)
            lnpost_launches("cfg2/prior_valid_single_batch", "k_lnpost_fast<0, 1, 1, false, false>", mod,
                            samples["prior_valid"][:1], 560)
        if "generic" in cases:
            os.environ["ISOCHRONES_AMD_PATH"] = "generic"
            ic_g, mod_g = bench.build_model()
            mod_g.lnpost(samples["prior"][0][:4096])
            for wl in ("prior_valid", "posterior"):
                lnpost_launches("generic/" + wl, "k_lnpost<0, 1, 0", mod_g, samples[wl], 560)
            os.environ.pop("ISOCHRONES_AMD_PATH")
            del mod_g, ic_g
        del mod, ic, samples
    if "cfg3" in cases:
        ic3, mod3 = bench_configs.cfg3_model()
        first = True
        for wl in ("prior", "prior_valid", "posterior"):
            batches = [bench_configs.cfg3_samples(n, wl, seed=3 + 17 * b) for b in range(NB)]
            if first:
                mod3.lnpost(batches[0][:4096])
                first = False
            lnpost_launches("cfg3/" + wl, "k_lnpost_fast<1, 2, 6, false, false>", mod3, batches, 2360)
            del batches
        del mod3, ic3
    if "astero" in cases:
        ic = ia.synthetic_track(bands=("V",))
        mod = ia.SingleStarModel(ic, Teff=(5770, 100), logg=(4.5, 0.1), feh=(0.0, 0.15), V=(10.0, 0.05),
                                 nu_max=(3000.0, 100.0), delta_nu=(135.0, 3.0))
        batches = [bench.make_samples(np.random.default_rng(12345 + b), n, "prior_valid") for b in range(NB)]
        mod.lnpost(batches[0][:4096])
        lnpost_launches("astero/prior_valid", "k_lnpost_fast<0, 1, 1, false, true>", mod, batches, 688)
        del batches
        del mod, ic
    if "tree" in cases:
        mod, pars = bench_configs.tree_model_and_samples(n)
        pt = torch.as_tensor(pars, device="cuda")
        mod.lnpost(pt[:4096])
        torch.cuda.synchronize()
        for _ in range(L):
            mod.lnpost(pt)
        torch.cuda.synchronize()
        manifest.append(dict(label="tree/posterior", kernel="k_lnpost_tree_fast<3, 2>", launches=L + 1, skip=2, n=n,
                             algorithmic_bytes_per_launch=float(2 * 384 + 2 * 3 * 128 + 56) * n))
        # the runtime-leaf-count form of the same kernel (what trees of 5-8 stars run): per-leaf values in LDS
        os.environ["ISOCHRONES_AMD_TREE_RUNTIME_LEAVES"] = "1"
        mod.lnpost(pt[:4096])
        torch.cuda.synchronize()
        for _ in range(L):
            mod.lnpost(pt)
        torch.cuda.synchronize()
        os.environ.pop("ISOCHRONES_AMD_TREE_RUNTIME_LEAVES")
        manifest.append(dict(label="tree_runtime_leaves/posterior", kernel="k_lnpost_tree_fast<3, 0>", launches=L + 1, skip=2, n=n,
                             algorithmic_bytes_per_launch=float(2 * 384 + 2 * 3 * 128 + 56) * n))
        del mod
    if "tree_generic" in cases:
        # the generic tree kernel (any tree, tables off the corner-packed path; per-leaf values in LDS since round 5)
        os.environ["ISOCHRONES_AMD_PATH"] = "generic"
        mod, pars = bench_configs.tree_model_and_samples(n)
        pt = torch.as_tensor(pars, device="cuda")
        mod.lnpost(pt[:4096])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(L):
            mod.lnpost(pt)
        e1.record()
        torch.cuda.synchronize()
        os.environ.pop("ISOCHRONES_AMD_PATH")
        manifest.append(dict(label="tree_generic/posterior", kernel="k_lnpost_tree(", launches=L + 1, skip=2, n=n,
                             ms_per_launch_profiled=e0.elapsed_time(e1) / L,
                             algorithmic_bytes_per_launch=float(2 * 384 + 2 * 3 * 128 + 56) * n))
        mod.ic.release()
        del mod
    if "quantiles" in cases:
        from isochrones_amd.catalog import CatalogPosterior, initial_positions
        from isochrones_amd.sampler import FusedEnsembleSampler
        bands = ["G", "BP", "RP"]
        ic = ia.synthetic_track(bands=bands)
        S, W, T = 100_000, 32, 100
        cat, _ = ia.synthetic_catalog(ic, S, bands=bands, seed=7, mag_unc=0.01)
        post = CatalogPosterior.from_catalog(cat, ic)
        pos, lnp, failed = initial_positions(post, W, rng_seed=0)
        if bool(failed.any()):
            good = int(torch.nonzero(~failed)[0])
            pos[failed] = pos[good]
            lnp[failed] = 0.0
        fs = FusedEnsembleSampler(post, W, seed=1)
        fs.run_mcmc(pos, T, lnprob0=lnp, store=True)
        fs.quantiles()
        torch.cuda.synchronize()
        for _ in range(L):
            fs.quantiles()
        torch.cuda.synchronize()
        manifest.append(dict(label="quantiles/100000x32x100", kernel="k_chain_quantiles_exact", launches=L + 1, skip=1,
                             n=S * 5, algorithmic_bytes_per_launch=float(S) * W * T * 5 * 8))
    if "primitives" in cases:
        # the batch API primitives (rows a3-a8): interp_mag on the corner-packed tables, interp_value on the wide pack
        ic = ia.synthetic_track()                                    # 11 default bands
        rng = np.random.default_rng(5)
        lo = np.array([0.1, 1.0, -4.0, 1.0, 0.0]); hi = np.array([10.0, 1710.0, 0.5, 3000.0, 1.0])
        pars = torch.as_tensor(np.ascontiguousarray(rng.uniform(lo, hi, size=(n, 5)).T), device="cuda")
        ci = ic.model_grid.interp.column_index
        todo = [("interp_mag/1_band", "k_interp_mag_fast<0, 1>", lambda: ic.interp_mag_device(pars, list(ic.bands)[:1]),
                 8 * 4 * 8 + 16 * 1 * 8 + 40 + 8 * 4),
                ("interp_mag/11_bands", "k_interp_mag_fast<0, 11>", lambda: ic.interp_mag_device(pars, list(ic.bands)),
                 8 * 4 * 8 + 16 * 11 * 8 + 40 + 8 * 14),
                ("interp_value/18_cols", "k_interp3_wide", lambda: ic.model_grid.interp.interp_device([pars[2], pars[0], pars[1]], np.arange(18)),
                 8 * 18 * 8 + 24 + 18 * 8)]
        for label, kernel, fn, nbytes in todo:
            fn(); fn()
            torch.cuda.synchronize()
            for _ in range(L):
                fn()
            torch.cuda.synchronize()
            manifest.append(dict(label=label, kernel=kernel, launches=L + 2, skip=2, n=n, algorithmic_bytes_per_launch=float(nbytes) * n))
        # the single-column interp_value also runs k_interp3_wide: after the 18-column launches in dispatch order
        one = np.array([ci["radius"]])
        f1 = lambda: ic.model_grid.interp.interp_device([pars[2], pars[0], pars[1]], one)
        f1(); torch.cuda.synchronize()
        for _ in range(L):
            f1()
        torch.cuda.synchronize()
        manifest.append(dict(label="interp_value/1_col", kernel="k_interp3_wide", launches=L + 1, skip=1, n=n,
                             algorithmic_bytes_per_launch=float(8 * 8 + 24 + 8) * n))
        del ic
    if "sampler" in cases:
        # the catalog sampler in its throughput form: one launch per half-step over 2 x 10^5 stars x 16 moves
        from isochrones_amd.catalog import CatalogPosterior, initial_positions
        from isochrones_amd.sampler import FusedEnsembleSampler
        bands = ["G", "BP", "RP"]
        ic = ia.synthetic_track(bands=bands)
        S, W = 200_000, 32
        cat, _ = ia.synthetic_catalog(ic, S, bands=bands, seed=7, mag_unc=0.01)
        post = CatalogPosterior.from_catalog(cat, ic)
        pos, lnp, failed = initial_positions(post, W, rng_seed=0)
        if bool(failed.any()):
            good = int(torch.nonzero(~failed)[0])
            pos[failed] = pos[good]
            lnp[failed] = 0.0
        os.environ["ISOCHRONES_AMD_SAMPLER"] = "stepwise"
        fs = FusedEnsembleSampler(post, W, seed=1)
        pos, lnp = fs.run_mcmc(pos, 100, lnprob0=lnp, store=False)          # burn in: walkers settle on their posteriors
        torch.cuda.synchronize()
        fs.run_mcmc(pos, L // 2, lnprob0=lnp, store=False)
        torch.cuda.synchronize()
        os.environ.pop("ISOCHRONES_AMD_SAMPLER")
        # 816 B per move: 384 (model cell) + 3 x 128 (BC) + position in/out
        manifest.append(dict(label="sampler_stepwise/200000x32", kernel="k_stretch_half<0, 1, 3", launches=2 * (100 + L // 2),
                             skip=200, n=S * W // 2, algorithmic_bytes_per_launch=float(S * W // 2) * (384 + 3 * 128 + 2 * 48)))
        # ... and as the library runs a catalog of this size since the auto rule changed: the persistent kernel, whose
        # workgroups own 16 stars each for all iterations of the call and run in rounds (one launch = L // 2 iterations)
        os.environ["ISOCHRONES_AMD_SAMPLER"] = "persistent"
        fs2 = FusedEnsembleSampler(post, W, seed=1)
        pos2, lnp2 = fs2.run_mcmc(pos, 10, lnprob0=lnp, store=False)
        torch.cuda.synchronize()
        fs2.run_mcmc(pos2, L // 2, lnprob0=lnp2, store=False)
        torch.cuda.synchronize()
        os.environ.pop("ISOCHRONES_AMD_SAMPLER")
        manifest.append(dict(label="sampler_persistent/200000x32", kernel="k_stretch_persist<0, 1, 3", launches=2, skip=1,
                             n=S * W * (L // 2), iterations=L // 2,
                             algorithmic_bytes_per_launch=float(S * W * (L // 2)) * (384 + 3 * 128) + float(S * W) * 2 * 48))
    if "catalog_ref" in cases:
        # the reference's own catalog shape in rounds: isochrones, 300 walkers (one ensemble per workgroup: the register-capped
        # kernel reads its star's block through scalar loads - DENSE + UNI), 10^4 stars, L // 2 iterations per launch
        from isochrones_amd.catalog import CatalogPosterior, initial_positions
        from isochrones_amd.sampler import FusedEnsembleSampler
        bands = ["G", "BP", "RP"]
        ic = ia.synthetic_isochrone(bands=bands)
        S, W = 10_000, 300
        cat, _ = ia.synthetic_catalog(ic, S, bands=bands, seed=7, mag_unc=0.01)
        post = CatalogPosterior.from_catalog(cat, ic)
        pos, lnp, failed = initial_positions(post, W, rng_seed=0)
        if bool(failed.any()):
            good = int(torch.nonzero(~failed)[0])
            pos[failed] = pos[good]
            lnp[failed] = 0.0
        fs = FusedEnsembleSampler(post, W, seed=1)
        pos, lnp = fs.run_mcmc(pos, 60, lnprob0=lnp, store=False)           # burn in
        torch.cuda.synchronize()
        it = 20
        fs.run_mcmc(pos, it, lnprob0=lnp, store=False)
        torch.cuda.synchronize()
        manifest.append(dict(label="catalog_reference_shape/10000x300", kernel="k_stretch_persist<1, 1, 3, true, false, true", launches=2, skip=1,
                             n=S * W * it, iterations=it,
                             algorithmic_bytes_per_launch=float(S * W * it) * (384 + 3 * 128) + float(S * W) * 2 * 48))
    json.dump(manifest, open(args.manifest, "w"), indent=1)
    print("manifest:", args.manifest, [m["label"] for m in manifest])


if __name__ == "__main__":
    main()
