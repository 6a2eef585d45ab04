#!/usr/bin/env python
"""Secondary measurements for the other BASELINE.json configs (the driver's contract line is
bench.py = configs[1]).  One JSON line per config:

  cfg1  Sun-like single star, 1e4 lnpost samples on the CPU path (oracle, scalar-call / batch)
  cfg3  binary (two-component flux sum), 6 bands + parallax, 1e6-sample batch, 1 GPU
  cfg4  ensemble MCMC 256 walkers x 5000 steps with the GPU lnpost, wall-clock (+ CPU estimate)
  cfg5  synthetic catalog, stars/s on the GPUs of this process group (weak scaling)

    python bench_configs.py [--configs cfg1,cfg3,cfg4,cfg5] [--stars 10000]
    python -m torch.distributed.run --nproc-per-node N bench_configs.py --configs cfg5
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def _oracle_ic(ic):
    import bench
    return bench.oracle_view(ic)[1]          # CPU-baseline legs only (never the measured GPU path)


def _time_kernel(mod, pars_t, reps, warm=10):
    import torch
    from isochrones_amd import _cabi, device as dev
    n = pars_t.shape[1]
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    h = mod.handle(torch.cuda.current_device())
    ms = C.c_double()
    lib = _cabi.lib()
    for r in (warm, reps):
        _cabi.check(lib.iso_time_lnpost(h, dev.ptr(pars_t), 1, n, n, dev.ptr(out), r, dev.stream_ptr(), C.byref(ms)))
    return ms.value, out


def cfg1():
    """The reference's own CPU-runnable case: 1e4 samples, Sun-like star, CPU path."""
    import isochrones_amd as ia
    import bench
    ic, mod = bench.build_model()
    rng = np.random.default_rng(12345)
    pars = bench.make_samples(rng, 10_000, "posterior")
    oic = _oracle_ic(ic)
    desc = mod.model_desc()
    soa = np.ascontiguousarray(pars.T)
    oic.lnpost(desc, soa, parts=False)
    t = time.perf_counter(); oic.lnpost(desc, soa, nthreads=1, parts=False); batch1 = time.perf_counter() - t
    t = time.perf_counter()
    for i in range(2000):                      # scalar-call mode: how the reference is really driven
        oic.lnpost(desc, soa[:, i:i + 1], parts=False)
    scalar = (time.perf_counter() - t) / 2000
    return {"config": "cfg1", "metric": "CPU lnpost evals/s, 1e4 samples (C port of the reference, 1 thread)",
            "batch_evals_per_s": 1e4 / batch1, "scalar_call_us": scalar * 1e6,
            "reference_published_us_per_call": 69.0, "note": "published: notebooks/Overview.ipynb:738 (laptop, numba)"}


def cfg3_model():
    """BASELINE configs[2]: binary, 6 bands + parallax on the full-size isochrone table."""
    import isochrones_amd as ia
    bands = ("J", "H", "K", "BP", "RP", "G")
    ic = ia.synthetic_isochrone(bands=bands)
    mod = ia.BinaryStarModel(ic, J=(9.3, 0.02), H=(9.0, 0.02), K=(8.95, 0.02), BP=(10.7, 0.002), RP=(9.8, 0.002),
                             G=(10.3, 0.001), parallax=(2.0, 0.05))
    return ic, mod


def cfg3_samples(n, workload, seed=3, rng=None):
    """[n, 6] rows (eep_0, eep_1, age, feh, distance, AV) of one of cfg 3's three sample distributions."""
    import isochrones_amd as ia
    rng = np.random.default_rng(seed) if rng is None else rng
    if workload == "prior_valid":
        # uniform over the populated part of the isochrone table: both components always reach
        # the BC gather (uncorrelated 384 B + 768 B reads per component)
        age = rng.uniform(6.0, 10.25, n)
        first = ia.grids.iso_eep_range(age + 0.05)[0] + 1.0
        last = ia.grids.iso_eep_range(age - 0.05)[1] - 1.0
        e = first[:, None] + rng.uniform(0, 1, (n, 2)) * (last - first)[:, None]
        return np.column_stack([e.max(axis=1), e.min(axis=1), age, rng.uniform(-4.0, 0.5, n),
                                rng.uniform(1.0, 1000.0, n), rng.uniform(0.0, 1.0, n)])
    if workload == "prior":
        lo = np.array([1.0, 1.0, 5.0, -4.0, 1.0, 0.0])
        hi = np.array([1710.0, 1710.0, 10.3, 0.5, 1000.0, 1.0])
        pars = rng.uniform(lo, hi, size=(n, 6))
        pars[:, :2] = -np.sort(-pars[:, :2], axis=1)
        return pars
    if workload == "posterior":
        c = np.array([350.0, 300.0, 9.7, 0.0, 500.0, 0.2])
        w = np.array([10.0, 10.0, 0.1, 0.1, 10.0, 0.05])
        pars = c + w * rng.standard_normal((n, 6))
        pars[:, 5] = np.abs(pars[:, 5])
        return pars
    raise ValueError(workload)


def cfg3_model_and_samples(n=1_000_000, seed=3):
    """The model and one batch of each distribution, drawn from one generator in the order (prior, prior_valid,
    posterior) - the batches rounds 1-2 measured."""
    ic, mod = cfg3_model()
    rng = np.random.default_rng(seed)
    sets = {wl: cfg3_samples(n, wl, rng=rng) for wl in ("prior", "prior_valid", "posterior")}
    return ic, mod, sets


def cfg3(n=1_000_000, reps=100):
    import torch
    ic, mod, sets = cfg3_model_and_samples(n)
    out = {}
    for workload, pars in sets.items():
        pt = torch.as_tensor(np.ascontiguousarray(pars.T), device="cuda")
        ms, res = _time_kernel(mod, pt, reps)
        oic = _oracle_ic(ic)
        ns = 50_000
        t = time.perf_counter()
        ref = oic.lnpost(mod.model_desc(), np.ascontiguousarray(pars[:ns].T), nthreads=os.cpu_count(), parts=False)
        cpu = ns / (time.perf_counter() - t)
        got = res[:ns].cpu().numpy()
        fin = np.isfinite(ref)
        rel = float(np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1, np.abs(ref[fin])))) if fin.any() else 0.0
        bytes_eval = 2 * 384 + 2 * 768 + 56
        # bytes_eval x n / t is SURVEY 8d's algorithmic rate, not an HBM rate: half of cfg 3's BC bytes are served by L2 /
        # Infinity Cache, so it exceeds the HBM peak; the memory-side bound is bench.py's counter-based `roofline.hbm`
        import bench
        out[workload] = {"kernel_ms": ms, "evals_per_s": n / (ms * 1e-3), "algorithmic_GBs": bytes_eval * n / (ms * 1e-3) / 1e9,
                         "roofline": bench.bounds("cfg3/" + workload, n, ms, float(bytes_eval) * n), "finite_fraction": float(fin.mean()),
                         "cpu_all_cores_evals_per_s_one_cold_pass": cpu, "parity_max_rel_err": rel,
                         "pattern_ok": bool(np.array_equal(np.isnan(got), np.isnan(ref)) and
                                            np.array_equal(np.isneginf(got), np.isneginf(ref)))}
    return {"config": "cfg3", "metric": "lnpost evals/s, binary 6 bands + parallax, 1e6 batch, 1 GPU",
            "bytes_per_eval": 2360, "reference_published_us_per_call": 719.0, **out}


def cfg4(nwalkers=256, nsteps=5000):
    import torch
    import bench
    ic, mod = bench.build_model()
    truth = np.array([1.0, 355.0, 0.0, 100.0, 0.1])
    rng = np.random.default_rng(1)
    p0 = truth + np.array([0.01, 2.0, 0.02, 1.0, 0.02]) * rng.standard_normal((nwalkers, 5))
    p0[:, 4] = np.abs(p0[:, 4])
    assert np.isfinite(mod.lnpost(p0)).all()
    from isochrones_amd.sampler import EnsembleSampler, FusedEnsembleSampler
    # (a) framework-op sampler: ~25 small launches per half-step around the batched lnpost kernel
    s = EnsembleSampler(nwalkers, 5, mod.lnpost, seed=2, device=torch.device("cuda"))
    s.run_mcmc(p0, 20, store=False)                   # warm-up
    torch.cuda.synchronize()
    t = time.perf_counter()
    s.run_mcmc(p0, 500, store=True)
    torch.cuda.synchronize()
    wall_ops = (time.perf_counter() - t) * (nsteps / 500.0)
    # (b) fused sampler: proposal + lnpost + accept in one kernel; "stepwise" = one launch per
    # half-step, "auto" picks the persistent one-launch kernel for an ensemble this small
    import os
    walls = {}
    for mode in ("stepwise", "auto"):
        os.environ["ISOCHRONES_AMD_SAMPLER"] = mode
        fsamp = FusedEnsembleSampler(mod, nwalkers, seed=2)
        fsamp.run_mcmc(p0, 20, store=False)
        fsamp.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        pos, lp = fsamp.run_mcmc(p0, nsteps, store=True)
        torch.cuda.synchronize()
        walls[mode] = time.perf_counter() - t
    os.environ.pop("ISOCHRONES_AMD_SAMPLER", None)
    wall = walls["auto"]
    acc = float(fsamp.acceptance_fraction.mean())
    # sampler-callback forms (what emcee does with lnpost as its log-probability function)
    one = truth.copy()
    mod.lnpost(one)
    t = time.perf_counter()
    for _ in range(2000):
        mod.lnpost(one)
    host_scalar_us = (time.perf_counter() - t) / 2000 * 1e6
    halfens = np.ascontiguousarray(p0[: nwalkers // 2])
    t = time.perf_counter()
    for _ in range(1000):
        mod.lnpost(halfens)
    host_vec_us = (time.perf_counter() - t) / 1000 * 1e6
    # CPU side: the same number of lnpost calls one at a time through the C port
    oic = _oracle_ic(ic)
    desc = mod.model_desc()
    soa = np.ascontiguousarray(p0.T)
    t = time.perf_counter()
    for i in range(2000):
        oic.lnpost(desc, soa[:, i % nwalkers:i % nwalkers + 1], parts=False)
    per_call = (time.perf_counter() - t) / 2000
    calls = nwalkers * nsteps
    return {"config": "cfg4", "metric": "wall-clock of a %d-walker x %d-step ensemble fit, GPU lnpost" % (nwalkers, nsteps),
            "gpu_wall_s": wall, "gpu_wall_s_stepwise_kernel": walls["stepwise"], "gpu_wall_s_framework_op_sampler": wall_ops, "lnpost_calls": calls, "us_per_step": wall / nsteps * 1e6, "acceptance": acc,
            "host_callback_scalar_us": host_scalar_us, "host_callback_half_ensemble_us": host_vec_us,
            "host_callback_estimated_wall_s": {"one_walker_per_call": host_scalar_us * 1e-6 * nwalkers * nsteps,
                                               "vectorized_half_ensembles": host_vec_us * 1e-6 * 2 * nsteps},
            "cpu_scalar_call_us": per_call * 1e6, "cpu_estimated_wall_s": per_call * calls,
            "reference_published_estimate_s": [69e-6 * calls, 719e-6 * calls]}


def astero(n=1_000_000, reps=50):
    """cfg-2 star plus asteroseismic constraints (nu_max, delta_nu; reference starmodel.py:1603-1612):
    fast kernel with the extra 128-B (nu_max, delta_nu) cell vs the generic kernel."""
    import torch
    import bench
    import isochrones_amd as ia
    out = {}
    for path in ("auto", "generic"):
        os.environ["ISOCHRONES_AMD_PATH"] = path
        ic = ia.synthetic_track(bands=("V",))
        mod = ia.SingleStarModel(ic, Teff=(5770, 100), logg=(4.5, 0.1), feh=(0.0, 0.15), V=(10.0, 0.05),
                                 nu_max=(3000.0, 100.0), delta_nu=(135.0, 3.0))
        pars = bench.make_samples(np.random.default_rng(12345), n, "prior_valid")
        pt = torch.as_tensor(np.ascontiguousarray(pars.T), device="cuda")
        ms, res = _time_kernel(mod, pt, reps)
        out[path] = {"kernel_ms": ms, "evals_per_s": n / (ms * 1e-3), "finite_fraction": float(torch.isfinite(res).double().mean())}
        del mod, ic
    os.environ.pop("ISOCHRONES_AMD_PATH", None)
    return {"config": "astero", "metric": "lnpost evals/s with nu_max + delta_nu terms, 1e6 batch", "bytes_per_eval": 560 + 128, **out}


def published():
    """The timings the reference records in its notebooks (laptop CPU, numba), repeated here through the same
    Python calls on the MIST-shaped synthetic tables."""
    import isochrones_amd as ia
    from isochrones_amd.mist import MIST_EvolutionTrack, MIST_Isochrone
    out = {}

    def tm(f, n=1, warm=1):
        for _ in range(warm):
            f()
        t = time.perf_counter()
        for _ in range(n):
            r = f()
        return (time.perf_counter() - t) / n, r

    trk = MIST_EvolutionTrack()
    iso = MIST_Isochrone(bands=["J", "H", "K", "BP", "RP", "G"])
    rng = np.random.default_rng(0)
    x = [rng.uniform(-1, 0.4, 10000), rng.uniform(0.5, 2.0, 10000), rng.uniform(200, 600, 10000)]
    out["track_grid.interp 10000 3-D points, 1 column (host arrays) [s]"] = \
        {"here": tm(lambda: trk.model_grid.interp(x, ["radius"]), 50)[0], "reference": 4.01e-3}
    out["track_grid.interp single 3-D point, 1 column [s]"] = \
        {"here": tm(lambda: trk.model_grid.interp([-0.12, 1.01, 353.1], ["radius"]), 2000)[0], "reference": 12.5e-6}
    N = 10000
    mass, age, feh = np.ones(N) * 1.01, np.ones(N) * 9.82, np.ones(N) * 0.02
    out["mist_track.generate(mass, age, feh), N = 10000 [s]"] = {"here": tm(lambda: trk.generate(mass, age, feh), 10)[0],
                                                                "reference": 112e-3}
    out["get_eep single [s]"] = {"here": tm(lambda: trk.get_eep(1.01, 9.51, 0.01), 2000)[0], "reference": 4.26e-6}
    out["get_eep single, accurate=True [s]"] = {"here": tm(lambda: trk.get_eep(1.01, 9.51, 0.01, accurate=True), 20)[0],
                                               "reference": 4.56e-3}
    single = ia.SingleStarModel(iso, Teff=(5770, 80), feh=(0.0, 0.1), logg=(4.44, 0.08), J=(9.0, 0.02), H=(8.7, 0.02), K=(8.6, 0.02))
    out["StarModel.lnpost(p), single star, spectroscopy + J, H, K [s]"] = \
        {"here": tm(lambda: single.lnpost([355.0, 9.6, 0.0, 100.0, 0.1]), 2000)[0], "reference": 69e-6}
    binary = ia.BinaryStarModel(iso, J=(9.3, 0.02), H=(9.0, 0.02), K=(8.95, 0.02), BP=(10.7, 0.002), RP=(9.8, 0.002),
                                G=(10.3, 0.001), parallax=(2.0, 0.05))
    pb = [350.0, 300.0, 9.7, 0.0, 500.0, 0.2]
    out["BinaryStarModel.lnpost(pars), 6 bands + parallax [s]"] = {"here": tm(lambda: binary.lnpost(pb), 2000)[0],
                                                                  "reference": 719e-6}
    dt, res = tm(lambda: binary.fit_multinest(n_live_points=2000, seed=1), 1, warm=0)
    out["binary nested fit, 2000 live points [s]"] = {"here": dt, "reference": 14 * 60.0, "lnpost_evaluations": res.ncall,
                                                     "logz": res.logz}
    return {"config": "published", "metric": "the reference's notebook timings (SURVEY 6) vs the same calls here", **out}


def nested(nlive=1000):
    """fit_multinest on the cfg-2 star (full-size tables): batched nested sampling, one fused lnpost
    launch per proposal batch.  The reference runs MultiNest with one Python lnpost call per point."""
    import bench
    ic, mod = bench.build_model()
    mod.fit_multinest(n_live_points=200, seed=0)          # warm-up (tables, pinned staging)
    t = time.perf_counter()
    res = mod.fit_multinest(n_live_points=nlive, seed=1)
    wall = time.perf_counter() - t
    t = time.perf_counter()
    res_c = mod.fit_multinest(n_live_points=nlive, seed=1, batched=False)
    wall_classic = time.perf_counter() - t
    return {"config": "nested", "wall_s_classic_loop": wall_classic, "logz_classic_loop": res_c.logz, "metric": "wall-clock of fit_multinest, %d live points, cfg-2 star" % nlive,
            "wall_s": wall, "lnpost_evaluations": res.ncall, "iterations": res.niter, "efficiency": res.efficiency,
            "logz": res.logz, "logz_err": res.logz_err, "prior_fraction_with_support": res.prior_fraction,
            "reference_published_estimate_s": [69e-6 * res.ncall, 719e-6 * res.ncall]}


def primitives(n=1_000_000, reps=20):
    """The batch primitives behind the public API: interp_mag with the 11 default bands
    (1 816 algorithmic B/sample) and interp_value of all 18 columns."""
    import torch
    import isochrones_amd as ia
    ic = ia.synthetic_track()                                    # 11 default bands
    rng = np.random.default_rng(5)
    lo = np.array([0.1, 1.0, -4.0, 1.0, 0.0]); hi = np.array([10.0, 1710.0, 0.5, 3000.0, 1.0])
    pars = torch.as_tensor(np.ascontiguousarray(rng.uniform(lo, hi, size=(n, 5)).T), device="cuda")
    out = {}
    cases = []
    for nb in (1, 3, 11):
        bands = list(ic.bands)[:nb]
        # algorithmic bytes: 8 corners x 4 columns + 16 corners x nb bands + 5 parameters in + 3 + nb out
        cases.append(("interp_mag_%d_bands" % nb, (lambda b=bands: ic.interp_mag_device(pars, b)),
                      8 * 4 * 8 + 16 * nb * 8 + 40 + 8 * (3 + nb)))
    ci = ic.model_grid.interp.column_index
    for label, cols in (("18_cols", np.arange(18)), ("3_cols", np.array([ci["Teff"], ci["logg"], ci["age"]])),
                        ("1_col", np.array([ci["radius"]]))):
        kk = len(cols)
        cases.append(("interp_value_" + label,
                      (lambda c=cols: ic.model_grid.interp.interp_device([pars[2], pars[0], pars[1]], c)),
                      8 * kk * 8 + 24 + kk * 8))
    for path in ("auto", "generic"):
        os.environ["ISOCHRONES_AMD_PATH"] = path
        for name, fn, nbytes in cases:
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out[name + ("" if path == "auto" else "_generic_kernel")] = {
                "ms": ms, "samples_per_s": n / (ms * 1e-3), "algorithmic_GBs": nbytes * n / (ms * 1e-3) / 1e9,
                "bytes_per_sample": nbytes}
    os.environ.pop("ISOCHRONES_AMD_PATH", None)
    return {"config": "primitives", "metric": "batch API primitives, 1e6 samples (includes output allocation)", **out}


def tree_model_and_samples(n=1_000_000):
    """Generic observation-tree model (docs/multiple.ipynb resolved binary: 3 unresolved bands + a resolved relative
    K image, 2 stars in one system) and a posterior-like [n, 6] sample batch."""
    import isochrones_amd as ia
    from isochrones_amd.observation import Observation, ObservationTree, Source
    ic = ia.synthetic_isochrone(bands=("J", "H", "K"))
    obs = ObservationTree(name="resolved")
    for band, m in zip("JHK", (12.11, 11.74, 11.68)):
        o = Observation("2MASS", band, 4)
        o.add_source(Source(m, 0.02))
        obs.add_observation(o)
    o = Observation("AO", "K", 0.1)
    o.add_source(Source(0.0, 0.02, separation=0, pa=0, relative=True, is_reference=True))
    o.add_source(Source(2.43, 0.02, separation=0.2, pa=100, relative=True, is_reference=False))
    obs.add_observation(o)
    mod = ia.StarModel(ic, obs=obs, parallax=(2.0, 0.05), Teff=(5834.0, 100), logg=(4.43, 0.15), feh=(-0.01, 0.1))
    rng = np.random.default_rng(8)
    c = np.array([300.0, 280.0, 9.6, 0.0, 400.0, 0.1]); w = np.array([10.0, 10.0, 0.1, 0.1, 20.0, 0.05])
    pars = c + w * rng.standard_normal((n, 6))
    pars[:, :2] = -np.sort(-pars[:, :2], axis=1)
    pars[:, 5] = np.abs(pars[:, 5])
    return mod, pars


def tree(n=1_000_000, reps=20):
    """Generic observation-tree model (docs/multiple.ipynb resolved binary); reference: 1.23 ms per lnpost call."""
    import torch
    mod, pars = tree_model_and_samples(n)
    ic = mod.ic
    pt = torch.as_tensor(pars, device="cuda")
    out = mod.lnpost(pt); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = mod.lnpost(pt)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    import bench
    orc = bench.oracle_view(ic)[0]
    ns = 20_000
    ref = orc.tree_lnpost(_oracle_ic(ic), mod.tree_desc(), np.ascontiguousarray(pars[:ns].T), nthreads=os.cpu_count())[0]
    got = out[:ns].cpu().numpy()
    fin = np.isfinite(ref)
    rel = float(np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1, np.abs(ref[fin])))) if fin.any() else 0.0
    return {"config": "tree", "metric": "generic observation-tree lnpost, resolved binary, 1e6 batch",
            "kernel_ms": ms, "evals_per_s": n / (ms * 1e-3), "finite_fraction": float(fin.mean()),
            "parity_max_rel_err": rel, "reference_published_us_per_call": 1230.0}


def fits(nwalkers=256, nsteps=5000, ops_steps=100):
    """cfg 4's fit shape (256 walkers x 5000 iterations) for the model classes that are not a BasicStarModel with at most
    12 bands: the resolved binary of docs/multiple.ipynb (an observation tree; reference: 1.23 ms per lnpost call, so
    1 570 s for this fit on its emcee path), IsoTrackModel, a 24-band single star.  Device-resident sampler (one persistent
    launch: k_stretch_tree / k_stretch_isotrack / k_stretch_wide) against the framework-op sampler (~25 launches per
    half-step around the batch kernel; timed on `ops_steps` iterations and scaled) and the reference's default fit shape
    (300 walkers, 200 + 100 iterations)."""
    import torch
    import isochrones_amd as ia
    from isochrones_amd.sampler import EnsembleSampler, FusedEnsembleSampler
    out = {"config": "fits", "metric": "wall-clock of a %d-walker x %d-step ensemble fit, device-resident sampler" % (nwalkers, nsteps),
           "walkers": nwalkers, "steps": nsteps}

    def start(mod, centre, width, W, order=None):
        rng = np.random.default_rng(3)
        got = []
        for _ in range(50):
            x = centre + width * rng.standard_normal((4 * W, len(centre)))
            x[:, -1] = np.abs(x[:, -1])
            if order:
                x[:, :order] = -np.sort(-x[:, :order], axis=1)
            ok = np.isfinite(mod.lnpost(x))
            got.extend(x[ok])
            if len(got) >= W:
                return np.array(got[:W])
        raise RuntimeError("no start points")

    def run(name, mod, centre, width, order=None):
        leg = {"n_params": int(mod.n_params)}
        p0 = start(mod, centre, width, nwalkers, order)
        fs = FusedEnsembleSampler(mod, nwalkers, seed=2)
        fs.run_mcmc(p0, 20, store=False)
        fs.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        fs.run_mcmc(p0, nsteps, store=True)
        torch.cuda.synchronize()
        leg["gpu_wall_s"] = time.perf_counter() - t
        leg["us_per_step"] = leg["gpu_wall_s"] / nsteps * 1e6
        leg["acceptance"] = float(fs.acceptance_fraction.mean())
        leg["finite_chain"] = bool(torch.isfinite(fs._lnprob).all())
        fs.close()
        es = EnsembleSampler(nwalkers, mod.n_params, mod.lnpost, seed=2, device=torch.device("cuda"))
        es.run_mcmc(p0, 5, store=False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        es.run_mcmc(p0, ops_steps, store=True)
        torch.cuda.synchronize()
        leg["framework_op_sampler_wall_s_scaled"] = (time.perf_counter() - t) * nsteps / ops_steps
        leg["resident_over_framework_ops"] = leg["framework_op_sampler_wall_s_scaled"] / leg["gpu_wall_s"]
        # the reference's default fit shape through the model's own fit_mcmc (start points included)
        torch.cuda.synchronize()
        t = time.perf_counter()
        mod.fit_mcmc(nwalkers=300, nburn=200, niter=100, p0=p0[0], seed=4)
        torch.cuda.synchronize()
        leg["fit_mcmc_300x300_wall_s"] = time.perf_counter() - t
        out[name] = leg

    mod, _ = tree_model_and_samples(16)
    run("tree_resolved_binary", mod, np.array([300.0, 280.0, 9.6, 0.0, 400.0, 0.1]), np.array([3.0, 3.0, 0.02, 0.02, 5.0, 0.01]), order=2)
    out["tree_resolved_binary"]["reference_published_us_per_call"] = 1230.0
    out["tree_resolved_binary"]["reference_fit_estimate_s"] = 1230e-6 * nwalkers * nsteps
    iso = ia.synthetic_isochrone(bands=("G", "BP", "RP"))
    track = ia.synthetic_track(bands=("G", "BP", "RP"))
    truth = np.array([355.0, 1.0, 9.6, 0.0, 300.0, 0.1])
    mags = track.interp_mag([truth[1], truth[0], truth[3], truth[4], truth[5]], ["G", "BP", "RP"])[3]
    it = ia.IsoTrackModel(iso, track, Teff=(5700.0, 150.0), feh=(0.0, 0.15), parallax=(1000.0 / 300.0, 0.05),
                          **{b: (float(m), 0.05) for b, m in zip(("G", "BP", "RP"), mags)})
    run("isotrack", it, truth, np.array([1.0, 0.005, 0.01, 0.01, 2.0, 0.01]))
    bands24 = tuple(list(ia.grids.KNOWN_BANDS) + ["X%02d" % j for j in range(16)])[:24]
    ic24 = ia.synthetic_track(bands=bands24)
    t24 = np.array([1.0, 355.0, 0.0, 300.0, 0.1])
    m24 = ic24.interp_mag(list(t24), list(bands24))[3]
    mod24 = ia.SingleStarModel(ic24, Teff=(5700.0, 120.0), parallax=(1000.0 / 300.0, 0.05), **{b: (float(m), 0.03) for b, m in zip(bands24, m24)})
    run("single_star_24_bands", mod24, t24, np.array([0.01, 1.0, 0.01, 1.0, 0.01]))
    return out


def cfg5(n_stars=10_000, nwalkers=32, nburn=150, niter=100):
    import torch
    import torch.distributed as dist
    import isochrones_amd as ia
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("ISO_BENCH_BACKEND", "nccl")       # test hooks, as in bench.py
    if os.environ.get("ISO_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    bands = ["G", "BP", "RP"]
    ic = ia.synthetic_track(bands=bands)
    cat, truth = ia.synthetic_catalog(ic, n_stars, bands=bands, seed=7, mag_unc=0.01)
    warm, _ = ia.synthetic_catalog(ic, 64, bands=bands, seed=1, mag_unc=0.01)
    ia.catalog.fit_stars_gpu(warm, ic, np.arange(64), nwalkers=nwalkers, nburn=5, niter=5)   # framework-kernel warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t = time.perf_counter()
    timings = {}
    res = ia.fit_catalog(cat, ic, nwalkers=nwalkers, nburn=nburn, niter=niter, seed=11 + rank, timings=timings)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t
    if world > 1:
        tm = torch.tensor([wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        wall = float(tm[0])
    ok = res["ok"].values == 1
    rel_d = np.abs(res.loc[ok, "distance_median"].values - truth.loc[ok, "distance"].values) / truth.loc[ok, "distance"].values
    out = {"config": "cfg5", "metric": "stars/s, synthetic catalog sharded over GPUs (batch_starfit path)",
           "n_gpus": world, "n_stars": n_stars, "wall_s": wall, "stars_per_s": n_stars / wall,
           "lnpost_evals": n_stars * nwalkers * (nburn + niter), "walkers": nwalkers, "steps": nburn + niter,
           "rank0_breakdown_s": {k: round(v, 4) for k, v in timings.items()}, "ok_fraction": float(ok.mean()), "median_rel_distance_err": float(np.median(rel_d)),
           "scaling": "weak/strong: fixed catalog split over ranks, no collective in the sampling loop"}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg1,cfg3,cfg4,cfg5")
    ap.add_argument("--stars", type=int, default=10_000)
    args = ap.parse_args()
    for name in args.configs.split(","):
        fn = {"cfg1": cfg1, "cfg3": cfg3, "cfg4": cfg4, "cfg5": lambda: cfg5(args.stars),
              "primitives": primitives, "tree": tree, "fits": fits, "nested": nested, "astero": astero, "published": published}[name.strip()]
        r = fn()
        if r is not None:
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
