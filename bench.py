#!/usr/bin/env python
"""Benchmark of the isochrones hot path on MI355X.

Metric (BASELINE.json): lnpost evaluations / second on a 10^6-sample batch over the MIST-shaped
grid.  A "step" is one pass of the fused lnpost kernel over one batch of synthetic samples that
are already resident in HBM.  Workload = BASELINE configs[1]: one Sun-like star
(Teff/logg/feh + V magnitude), evolution-track parametrisation (mass, eep, feh, distance, AV),
full-size synthetic MIST track table [15,196,1710,18] + BC table [70,26,18,13,nb] (the model's packed BC
holds its one band).  Default sample distribution: uniform over the populated part of the table
("prior_valid": uncorrelated gathers over the whole 1.9 GB packed table, ~98 % of the samples evaluate the
complete path).

The timed steps ROTATE over `--batches` (default 8) distinct seeded batches, step k evaluating batch k mod 8:
a launch touches ~0.48 GB of distinct table lines, eight of them ~3.8 GB - far more than the 256 MiB Infinity
Cache - so no launch finds lines a previous launch left in a cache.  The one-batch-repeated figure of rounds 1-2 is
reported beside it (`roofline.single_batch`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 1000000] [--workload prior|posterior]

`value` = evaluations of all ranks / the slowest rank's DEVICE time for its K launches (HIP events on the launch
stream, inside the barrier + synchronize bracket); the wall-clock of the same bracket is reported as
`wall_ms_per_step` / `value_wall` (with K = 20 a bracket lasts 1.5 ms, of which an RCCL barrier is several per cent).

N > 1 (launched by torch.distributed.run, one rank per GPU): rank 0 builds the tables and broadcasts them over RCCL
(`startup`), every rank evaluates its own star's batches — independent posteriors, no data-path collective (weak
scaling).

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` and `cpu_baseline`.
After the timed region (never part of `value`) the same ranks run BASELINE configs[4], the catalog path: a fixed
synthetic catalog fitted by `fit_catalog` (stars sharded with batch_starfit's rule, device-resident sampler, one RCCL
all-gather of the result rows), reported as `catalog` (stars/s at this N; strong scaling); at N = 1 also configs[2]
and [3] and the CPU baselines of every config.  `--no-catalog` / `--no-extras` skip them.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
#: algorithmic bytes per lnpost evaluation, single star / 1 band:
#: 8 corners x (4 likelihood + 2 prior columns) x 8 B + 16 BC corners x 8 B + (5 params + 1 out) x 8 B
BYTES_PER_EVAL_SINGLE_1BAND = 8 * 6 * 8 + 16 * 1 * 8 + 6 * 8


_PMC = None


def pmc_record(label, n):
    """Per-launch PMC figures of a workload from profiles/pmc_traffic.json (measured beforehand with
    tools/pmc_collect.sh; None if absent or taken at another batch size)."""
    global _PMC
    if _PMC is None:
        try:
            _PMC = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            _PMC = {}
    rec = _PMC.get(label)
    # only counters taken on launches that rotate over distinct batches describe the rotating launches timed here
    return rec if isinstance(rec, dict) and rec.get("n") == n and rec.get("distinct_batches", 1) > 1 else None


def bounds(label, n, kernel_ms, algorithmic_bytes):
    """What bounds a launch: the larger of (a) the memory side - L2-miss ("fabric") bytes per launch from the PMC passes
    over the HBM peak - and (b) VALU issue - busy cycles per SIMD from the PMC passes over the cycles of this launch.
    Kernel time is measured in this run; the counter values are static (separate rocprofv3 --pmc passes cannot run
    inside a timed region) and say so.  `frac` is at most 1 by construction for the VALU bound (busy cycles cannot
    exceed elapsed cycles) and for the memory bound as long as the misses are served by HBM; traffic served by the
    Infinity Cache can exceed the HBM peak, which would show as frac > 1 and mean "cache-resident"."""
    t = kernel_ms * 1e-3
    out = {"algorithmic_GBs": algorithmic_bytes / t / 1e9, "kernel_ms": kernel_ms}
    rec = pmc_record(label, n)
    if rec is None:
        out.update(bound=None, frac=None, note="no PMC record for this workload / batch size")
        return out
    hbm = rec["fabric_bytes"] / t / 1e9
    valu = rec["valu_busy_cycles_per_simd"] / (t * rec["effective_clock_GHz"] * 1e9)
    out.update(hbm={"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                    "traffic": rec["fabric_bytes"], "traffic_over_algorithmic": rec["fabric_bytes"] / algorithmic_bytes,
                    "l2_hit_rate": rec.get("l2_hit_rate")},
               valu={"busy_cycles_per_simd": rec["valu_busy_cycles_per_simd"], "clock_GHz": rec["effective_clock_GHz"],
                     "frac": valu, "insts_per_wave": rec.get("valu_insts_per_wave")},
               source="static: profiles/pmc_traffic.json <- %s (separate rocprofv3 --pmc passes over "
                      "tools/pmc_workload.py%s; counter -> byte factors: %s; SQ_ACTIVE_INST_VALU x4 / 1024 SIMDs); "
                      "kernel time: this run" % ((_PMC or {}).get("source", "?"),
                                                 ", launches rotating over %d batches" % rec["distinct_batches"] if rec.get("distinct_batches") else "",
                                                 (_PMC or {}).get("byte_factors", "FETCH_SIZE x 1024 x 2 (gfx950) + WRITE_SIZE x 1024")))
    out["bound"] = "hbm" if hbm / HBM_PEAK_GBS >= valu else "valu"
    out["frac"] = max(hbm / HBM_PEAK_GBS, valu)
    return out


def make_samples(rng, n, workload):
    """[n, 5] float64 (mass, eep, feh, distance, AV) rows."""
    if workload == "prior":
        # prior-wide: uniform over the table's extent -> uncorrelated gathers over the whole table
        lo = np.array([0.1, 1.0, -4.0, 1.0, 0.0])
        hi = np.array([10.0, 1710.0, 0.5, 3000.0, 1.0])
        return rng.uniform(lo, hi, size=(n, 5))
    if workload == "prior_valid":
        # as "prior", but EEPs only where the (mass, feh) track is populated: ~every sample takes
        # the full path (model gather + BC gather), none is cut short by NaN padding
        from isochrones_amd import grids as G
        x = make_samples(rng, n, "prior")
        last = G.track_max_eep(x[:, 0], G.MIST_FEHS[np.clip(np.searchsorted(G.MIST_FEHS, x[:, 2]), 0, 14)])
        last = np.minimum(last, G.track_max_eep(x[:, 0], G.MIST_FEHS[np.clip(np.searchsorted(G.MIST_FEHS, x[:, 2]) - 1, 0, 14)]))
        mlo = G.mist_masses()[np.clip(np.searchsorted(G.mist_masses(), x[:, 0]) - 1, 0, 195)]
        last = np.minimum(last, G.track_max_eep(mlo, -1.0 * np.ones(n)))
        x[:, 1] = 1.0 + (x[:, 1] - 1.0) / 1709.0 * (last - 2.0)
        return x
    if workload == "posterior":
        # MCMC-like: a Gaussian ball around a Sun-like solution -> cache-resident gathers
        c = np.array([1.0, 355.0, 0.0, 100.0, 0.1])
        w = np.array([0.05, 15.0, 0.1, 5.0, 0.05])
        x = c + w * rng.standard_normal((n, 5))
        x[:, 4] = np.abs(x[:, 4])
        return x
    raise ValueError(workload)


def build_model(bands=("V",)):
    import isochrones_amd as ia
    ic = ia.synthetic_track(bands=bands)          # full MIST-shaped tables
    mod = ia.SingleStarModel(ic, Teff=(5770, 100), logg=(4.5, 0.1), feh=(0.0, 0.15), V=(10.0, 0.05))
    return ic, mod


def cpu_quota_cores():
    """CFS bandwidth quota of this container in CPUs (cgroup v2 cpu.max / v1 cfs_quota_us), None if unlimited."""
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            return None if q == "max" else float(q) / float(per)
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def oracle_view(ic):
    """The CPU checker's view of an interpolator's tables — the only place outside tests/ and
    __graft_entry__.smoke() that touches oracle/ (the cpu_baseline legs here and in bench_configs.py)."""
    from oracle import oracle as orc
    m, b = ic.model_grid.interp, ic.bc_grid.interp
    return orc, orc.OracleIC(ic.kind, orc.OracleTable(m.grid, m.index_columns), orc.OracleTable(b.grid, b.index_columns),
                             ic._cols, ic._prior_cols, ic._astero_cols)


def cpu_baseline(ic, mod, pars_host, wall_budget_s=4.0, full_passes=3, scalar_calls=10_000):
    """Time the C oracle (the reference's algorithm restated, oracle/iso_oracle.c) on this host:
    repeated passes over the same batch, all host cores (OpenMP static) for ~wall_budget_s of wall
    time, then one thread for about the same."""
    orc, oic = oracle_view(ic)
    desc = mod.model_desc()
    cores = max(1, min(orc.max_threads(), os.cpu_count() or 1))
    quota = cpu_quota_cores()
    soa = np.ascontiguousarray(pars_host.T)
    n = soa.shape[1]
    out = oic.lnpost(desc, soa, nthreads=cores, parts=False)          # warm-up (threads, page faults)
    # Containers often carry a CFS quota far below the host's core count (this pool: 16 of 256 CPUs): a pass
    # that starts in a fresh quota period runs unthrottled, sustained passes are throttled.  `value` is the
    # best pass (what the host cores can do - the conservative figure for any GPU/CPU ratio), the sustained
    # rate under the quota is reported next to it.
    times, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < wall_budget_s:
        t = time.perf_counter()
        oic.lnpost(desc, soa, nthreads=cores, parts=False)
        times.append(time.perf_counter() - t)
        if quota and quota < cores:
            time.sleep(0.2)                                            # let the quota period roll over
    dt = time.perf_counter() - t0
    passes = len(times)
    # the same passes with as many threads as the quota grants (threads = min(cores, quota)): no throttling, so the passes
    # agree with each other - `value_quota` is their median, the stable figure beside the best-of `value`
    quota_threads = max(1, min(cores, int(quota))) if quota else cores
    times_q = []
    if quota_threads != cores:
        oic.lnpost(desc, soa, nthreads=quota_threads, parts=False)
        tq = time.perf_counter()
        while time.perf_counter() - tq < min(wall_budget_s, 3.0) or len(times_q) < 3:
            t = time.perf_counter()
            oic.lnpost(desc, soa, nthreads=quota_threads, parts=False)
            times_q.append(time.perf_counter() - t)
    else:
        times_q = list(times)
    # B2: one thread, the reference's serial interp_mags-style loop without Python overhead, at 10^4
    # samples (BASELINE configs[0]'s size) and over the whole batch; median of 5 / 3 passes
    def one_thread(m, reps):
        sm = np.ascontiguousarray(soa[:, :m])
        oic.lnpost(desc, sm, nthreads=1, parts=False)
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            oic.lnpost(desc, sm, nthreads=1, parts=False)
            ts.append(time.perf_counter() - t)
        return m / float(np.median(ts))

    t1 = time.perf_counter()
    b2_small = one_thread(min(n, 10_000), 5)
    b2_full = one_thread(n, full_passes)
    dt1 = time.perf_counter() - t1
    # B1, scalar-call mode: one sample per call through the oracle's C ABI from a Python loop - how the reference
    # is actually driven by its samplers (published there: 69 us per call single star, numba); 10^4 calls
    one = [np.ascontiguousarray(soa[:, k:k + 1]) for k in range(min(n, scalar_calls))]
    for k in range(200):
        oic.lnpost(desc, one[k % len(one)], nthreads=1, parts=False)
    tc = time.perf_counter()
    for col in one:
        oic.lnpost(desc, col, nthreads=1, parts=False)
    scalar_us = (time.perf_counter() - tc) / len(one) * 1e6
    p1, n1 = full_passes, n
    # `value` is the REPRODUCIBLE figure: the median pass with as many threads as the container's CPU quota grants (passes
    # agree to ~10 %); the best pass with every hardware thread - a pass that happened to start in a fresh quota period, 5-6 x
    # the median one on this pool - is kept beside it as `value_best_pass` (the conservative figure for a GPU / CPU ratio)
    return dict(value=n / float(np.median(times_q)), unit="evals/s", cores=quota_threads, kind="port", scalar_call_us=scalar_us,
                value_best_pass=n / min(times), best_pass_threads=cores,
                sample="median of %d passes over the same %d-sample batch with %d threads (= the container's CPU quota: %s), "
                       "C restatement of the reference (oracle/iso_oracle.c), OpenMP static; value_best_pass: best of %d "
                       "passes with all %d hardware threads (%.1f s wall); 1-thread figure: %d passes over the first %d "
                       "samples (%.1f s)"
                       % (len(times_q), n, quota_threads, ("%.1f CPUs" % quota) if quota else "none", passes, cores, dt, p1, n1, dt1),
                value_median_pass=n / float(np.median(times)), cpu_quota_cores=quota,
                value_quota=n / float(np.median(times_q)), quota_threads=quota_threads, quota_passes=len(times_q),
                quota_pass_spread=float((max(times_q) - min(times_q)) / np.median(times_q)),
                value_1thread=b2_full,
                modes={"B1_scalar_call": {"us_per_call": scalar_us, "calls": len(one), "evals_per_s": 1e6 / scalar_us},
                       "B2_one_thread_1e4": {"evals_per_s": b2_small}, "B2_one_thread_full_batch": {"evals_per_s": b2_full},
                       "B3_all_cores_full_batch": {"evals_per_s": n / min(times), "median_pass": n / float(np.median(times)),
                                                   "threads": cores},
                       "B3_quota_threads_full_batch": {"evals_per_s": n / float(np.median(times_q)), "threads": quota_threads,
                                                       "passes": len(times_q)}}), out


def cpu_mcmc_baseline(ic, mod, p0, nsteps, seed, gpu_chain=None, gpu_lnp=None):
    """BASELINE configs[3] on the host: the same 256-walker stretch-move fit (same start points, same Philox stream as
    the device sampler: oracle/cpu_sampler.py) with the oracle's lnpost as the log-probability function - once with
    one call per walker and half-step (how emcee drives the reference's lnpost, starmodel.py:951-969) and once with one
    call per half-ensemble (emcee's vectorize=True).  With the GPU's stored chain the two fits are compared step by
    step (same moves, so the chains agree to rounding until - if ever - a tie flips a decision)."""
    from oracle import cpu_sampler
    orc, oic = oracle_view(ic)
    desc = mod.model_desc()
    spent = [0.0, 0]

    def lnpost_rows(rows):
        t = time.perf_counter()
        r = oic.lnpost(desc, np.ascontiguousarray(rows.T), nthreads=1, parts=False)
        spent[0] += time.perf_counter() - t
        spent[1] += rows.shape[0]
        return r

    lnp0 = lnpost_rows(p0)
    out = {}
    for mode, scalar in (("vectorized_half_ensembles", False), ("one_walker_per_call", True)):
        spent[:] = [0.0, 0]
        t = time.perf_counter()
        pos, lnp, chain, clnp, nacc = cpu_sampler.stretch_fit(lnpost_rows, p0, lnp0, nsteps, a=2.0, seed=seed,
                                                              scalar_calls=scalar)
        wall = time.perf_counter() - t
        out[mode] = {"wall_s": wall, "lnpost_s": spent[0], "lnpost_evals": spent[1],
                     "acceptance": float(nacc.mean() / nsteps)}
    cmp_ = None
    if gpu_chain is not None:
        same = np.all(np.abs(chain - gpu_chain) <= 1e-9 * (1.0 + np.abs(chain)), axis=(1, 2))
        first_diff = int(np.argmin(same)) if not same.all() else None
        upto = nsteps if first_diff is None else first_diff
        cmp_ = {"steps_identical_to_1e-9": int(upto), "of_steps": int(nsteps),
                "max_rel_position_diff": float(np.max(np.abs(chain[:upto] - gpu_chain[:upto]) / (1.0 + np.abs(chain[:upto])))) if upto else None,
                "max_abs_lnpost_diff": float(np.max(np.abs(clnp[:upto] - gpu_lnp[:upto]))) if upto else None,
                "note": "free-running CPU fit vs the GPU's stored chain, same start points and random numbers: the two make the "
                        "same moves; their positions differ in the last bits from the first step (fused multiply-adds on the "
                        "device, two roundings in numpy) and a stretch move multiplies a difference by z up to 2, so the gap "
                        "grows geometrically until an accept/reject decision flips - which is why the move-by-move check of "
                        "the GPU chain (tests/test_gpu_sampler_oracle.py, all 1.28e6 moves) is teacher-forced"}
    return out, cmp_


def cpu_catalog_baseline(ic, cat, stars, nwalkers, nburn, niter, seed=5):
    """BASELINE configs[4] on the host for a SUBSAMPLE of the catalog's stars, one after another as a batch_starfit
    worker does (scripts/batch_starfit: one starfit per line): per star a start-point search (8 x W candidates in the
    parameter bounds, keep the best W), burn-in + sampling with the stretch move of oracle/cpu_sampler.py and the
    oracle's lnpost (one call per walker and half-step), 16/50/84 % summaries with numpy.percentile."""
    from oracle import cpu_sampler
    orc, oic = oracle_view(ic)
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    ok, evals = 0, 0
    for i in stars:
        m = cat.model(int(i), ic)
        desc = m.model_desc()
        D = m.n_params
        lo = np.array([desc.bound_lo[j] for j in range(D)])
        hi = np.array([desc.bound_hi[j] for j in range(D)])
        K = 8 * nwalkers
        cand = lo + (hi - lo) * rng.uniform(size=(K, D))
        if ic.eep_replaces == "age":              # evolution-track parametrisation (mass, eep, feh, distance, AV)
            cand[:, 0] = np.exp(rng.uniform(np.log(lo[0]), np.log(hi[0]), K))               # mass: log-uniform
        if desc.has_parallax and desc.plx_val > 0:
            d0, rel = 1000.0 / desc.plx_val, min(max(desc.plx_unc / desc.plx_val, 1e-3), 0.3)
            cand[:, 3] = d0 * (1.0 + 4.0 * rel * (2.0 * rng.uniform(size=K) - 1.0))
        f = lambda rows: oic.lnpost(desc, np.ascontiguousarray(rows.T), nthreads=1, parts=False)
        lnp = f(cand)
        evals += K
        lnp[~np.isfinite(lnp)] = -np.inf
        best = np.argsort(-lnp)[:nwalkers]
        if not np.isfinite(lnp[best]).all():
            continue
        pos, lp, _, _, _ = cpu_sampler.stretch_fit(f, cand[best], lnp[best], nburn, seed=seed, row0=int(i) * nwalkers, store=False)
        pos, lp, chain, clnp, nacc = cpu_sampler.stretch_fit(f, pos, lp, niter, seed=seed, step0=nburn, row0=int(i) * nwalkers)
        np.percentile(chain.reshape(-1, D), [50, 16, 84], axis=0)
        evals += nwalkers * (nburn + niter)
        ok += 1
    wall = time.perf_counter() - t0
    return {"stars": int(len(stars)), "fitted": ok, "wall_s": wall, "stars_per_s": len(stars) / wall, "cores": 1, "kind": "port",
            "lnpost_evals": evals,
            "sample": "%d stars of the 10^4-star catalog fitted one after another on one host thread: start-point search "
                      "(%d candidates), %d + %d steps x %d walkers, one oracle lnpost call per walker and half-step, "
                      "numpy.percentile summaries" % (len(stars), 8 * nwalkers, nburn, niter, nwalkers)}


def catalog_leg(ic, rank, world, barrier, dist, reduce_device, sizes=(10_000, 400_000), nwalkers=32, nburn=150, niter=100,
                cpu_subsample=0, timed_passes=None):
    """The catalog path (BASELINE configs[4]) as `fit_catalog` runs it: the whole catalog has a fixed size (strong
    scaling), rank r fits the stars scripts/batch_starfit would give task r (NR % P) with the device-resident sampler,
    and ONE all-gather (RCCL on GPUs) hands every rank all result rows.  Wall-clock of the slowest rank, bracketed by
    barriers; the fit / gather split comes from fit_catalog itself.  Every rank reaches every collective whatever happens
    to its own work (fit_catalog's failure isolation: a failed shard contributes NaN rows and an error text)."""
    import torch
    import isochrones_amd as ia
    from isochrones_amd.catalog import fit_stars_gpu
    bands = ["G", "BP", "RP"]
    err = None
    try:
        warm, _ = ia.synthetic_catalog(ic, 64, bands=bands, seed=1, mag_unc=0.01)
        fit_stars_gpu(warm, ic, np.arange(64), nwalkers=nwalkers, nburn=5, niter=5)      # framework-kernel warm-up
    except Exception as e:           # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    out = {"rule": "fit_catalog: star i -> rank (i + 1) % P, no collective in the fit, one all-gather of the result rows",
           "parametrisation": list(ic.param_names), "walkers": nwalkers, "steps": nburn + niter, "nburn": nburn, "niter": niter,
           "bands": bands, "world": world,
           "backend": (dist.get_backend() if dist is not None else None)}
    for n_stars in sizes:
        cat, res, tm = None, None, {}
        try:
            # every rank holds the star list (the reference's workers all read the same list file); deterministic
            cat, _truth = ia.synthetic_catalog(ic, n_stars, bands=bands, seed=7, mag_unc=0.01)
        except Exception as e:       # noqa: BLE001
            err = err or "%s: %s" % (type(e).__name__, e)
        # one untimed pass first, as the W warm-up steps of the metric: the first fit of a size pays for the allocator's
        # first 26 GB of chain storage (hipMalloc + page tables: 0.5 s -> 1.0-1.4 s for the 4 x 10^5-star shard)
        # a small catalog's fit is a few milliseconds: one timed pass is a coin toss (host jitter of the ~40 framework calls
        # around the kernels), so sizes up to 20 000 stars take five timed passes and report the median one (all walls listed)
        first = None
        passes = []
        n_timed = timed_passes or (5 if n_stars <= 20000 else 1)
        for timed in (False,) + (True,) * n_timed:
            barrier()
            t0 = time.perf_counter()
            try:
                if cat is not None:
                    res = ia.fit_catalog(cat, ic, strict=False, nwalkers=nwalkers, nburn=nburn, niter=niter, seed=11 + rank)
                    tm = dict(res.attrs.get("timings", {}))
                    if res.attrs.get("shard_errors"):
                        err = err or "; ".join("rank %d: %s" % kv for kv in sorted(res.attrs["shard_errors"].items()))
            except Exception as e:       # noqa: BLE001
                err = err or "%s: %s" % (type(e).__name__, e)
            torch.cuda.synchronize()
            barrier()
            wall = time.perf_counter() - t0
            if not timed:
                first = wall
            else:
                passes.append((wall, dict(tm)))
        if passes:
            passes.sort(key=lambda p: p[0])
            wall, tm = passes[len(passes) // 2]
        ok = float(np.mean(res["ok"].values == 1)) if res is not None else 0.0
        stats = torch.tensor([wall, tm.get("fit_s", 0.0), tm.get("gather_s", 0.0), 1.0 if err is not None else 0.0],
                             dtype=torch.float64, device=reduce_device)
        share = torch.zeros(world, dtype=torch.float64, device=reduce_device)
        share[rank] = float(tm.get("stars_of_this_rank", 0))
        if dist is not None:
            dist.all_reduce(stats, op=dist.ReduceOp.MAX)
            dist.all_reduce(share, op=dist.ReduceOp.SUM)
        wall, fit_s, gather_s, failed = float(stats[0]), float(stats[1]), float(stats[2]), bool(stats[3] > 0)
        if failed:
            out["%d_stars" % n_stars] = {"error": err or "a rank other than 0 failed"}
        else:
            out["%d_stars" % n_stars] = {"wall_s": wall, "stars_per_s": n_stars / wall, "fit_s": fit_s, "gather_s": gather_s,
                                         # rank 0's shard by phase (seconds): per-star blocks, start points (one kernel), sampler, summaries
                                         **{k: float(v) for k, v in (tm.get("phases") or {}).items()},
                                         "first_call_wall_s": first, "timed_passes": len(passes),
                                         "wall_s_all_passes": [float(p[0]) for p in passes],
                                         "stars_per_rank_all": [int(x) for x in share.tolist()],
                                         "rows_gathered_on_rank0": int(np.isfinite(res.iloc[:, -1].values).sum()),
                                         "lnpost_evals": int(n_stars) * nwalkers * (nburn + niter), "ok_fraction": ok}
            if world == 1 and n_stars == sizes[0] and n_stars >= 8:
                # what eight GPUs could show on this catalog: the share one of them gets - star i -> rank (i + 1) % 8, n / 8 stars -
                # fitted here on one GPU (median of five passes), against the whole catalog's fit above.  A projection from two
                # measurements, not a scaling curve: tables broadcast, rendezvous and the gather of the rows are not in it.
                try:
                    share = np.array([i for i in range(n_stars) if (i + 1) % 8 == 0])
                    walls8 = []
                    for _ in range(6):
                        torch.cuda.synchronize()
                        t8 = time.perf_counter()
                        fit_stars_gpu(cat, ic, share, nwalkers=nwalkers, nburn=nburn, niter=niter, seed=11)
                        torch.cuda.synchronize()
                        walls8.append(time.perf_counter() - t8)
                    w8 = float(np.median(walls8[1:]))
                    out["%d_stars" % n_stars]["projection_8gpu"] = {
                        "stars_of_one_rank": int(share.size), "fit_s_of_one_rank": w8, "fit_s_whole_catalog_one_gpu": fit_s,
                        "speedup_upper_bound": fit_s / w8,
                        "note": "fit of the n / 8 stars one GPU of eight gets, measured on this GPU, against the fit of the whole "
                                "catalog; no broadcast / gather / rendezvous in either number"}
                except Exception as e:       # noqa: BLE001
                    out["%d_stars" % n_stars]["projection_8gpu"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if cpu_subsample and rank == 0 and n_stars == sizes[0]:
                try:
                    stars = np.linspace(0, n_stars - 1, cpu_subsample).astype(int)
                    base = cpu_catalog_baseline(ic, cat, stars, nwalkers, nburn, niter)
                    base["gpu_over_cpu_one_thread"] = out["%d_stars" % n_stars]["stars_per_s"] / base["stars_per_s"]
                    out["%d_stars" % n_stars]["cpu_baseline"] = base
                except Exception as e:       # noqa: BLE001
                    out["%d_stars" % n_stars]["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        cat, res = None, None
    return out


class Rotation:
    """`nb` distinct seeded sample batches of one model resident in HBM and the timing call that rotates over them."""

    def __init__(self, handle, batches_host, stream):
        import ctypes as C
        import torch
        from isochrones_amd import _cabi, device as dev
        self.C, self.lib, self.check, self.handle, self.stream = C, _cabi.lib(), _cabi.check, handle, stream
        self.n = batches_host[0].shape[0]
        self.pars = [torch.as_tensor(np.ascontiguousarray(b.T), device="cuda") for b in batches_host]     # SoA [D, n]
        self.outs = [torch.empty(self.n, dtype=torch.float64, device="cuda") for _ in batches_host]
        nb = len(batches_host)
        self.p_arr = (C.c_void_p * nb)(*[t.data_ptr() for t in self.pars])
        self.o_arr = (C.c_void_p * nb)(*[t.data_ptr() for t in self.outs])
        self.nb = nb

    def run(self, reps):
        """Mean milliseconds per launch over `reps` launches, launch r on batch r % nb (HIP events on the stream)."""
        ms = self.C.c_double()
        self.check(self.lib.iso_time_lnpost_rotating(self.handle, self.p_arr, self.o_arr, self.nb, 1, self.n, self.n,
                                                     int(reps), self.stream, self.C.byref(ms)))
        return ms.value

    def run_single(self, reps, b=0):
        ms = self.C.c_double()
        self.check(self.lib.iso_time_lnpost(self.handle, self.C.c_void_p(self.pars[b].data_ptr()), 1, self.n, self.n,
                                            self.C.c_void_p(self.outs[b].data_ptr()), int(reps), self.stream,
                                            self.C.byref(ms)))
        return ms.value


def spawn_ranks(n_ranks, argv=None):
    """Re-run this script under torch.distributed.run with one rank per GPU (a bare `python bench.py --gpus N`).
    Returns the launcher's exit code; the ranks inherit stdout / stderr, so rank 0's JSON line is this process's."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    cmd += list(sys.argv[1:] if argv is None else argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--batches", type=int, default=8, help="distinct seeded sample batches the timed steps rotate over")
    ap.add_argument("--preroll", type=int, default=400,
                    help="untimed launches before the W warm-up steps (the GPU's clock ramp after set-up lasts ~10 ms)")
    ap.add_argument("--workload", default="prior_valid", choices=["prior", "prior_valid", "posterior"],
                    help="prior_valid (default): uniform over the populated part of the table, ~98 %% of the samples "
                         "take the full path (every evaluation moves its 560 algorithmic bytes); prior: uniform over "
                         "the table's bounding box (SURVEY 8d (i): ~34 %% fall on NaN padding / are cut by the prior); "
                         "posterior: MCMC-like Gaussian ball (cache resident)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-catalog", action="store_true", help="skip the catalog leg (BASELINE configs[4])")
    ap.add_argument("--extras-timeout", type=float, default=300.0,
                    help="seconds after which the line is printed with whatever the secondary legs have produced")
    ap.add_argument("--path", default=None, choices=["auto", "compact", "generic"],
                    help="kernel/table-layout selection (default: library default = auto)")
    args = ap.parse_args()

    if args.path:
        os.environ["ISOCHRONES_AMD_PATH"] = args.path
    # the host driver only supports dmabuf IPC: RCCL between the ranks of one node needs this (already exported on the
    # GPU boxes; set here as well so that a bare launch works)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import threading
    import torch
    import ctypes as C
    import isochrones_amd as ia
    from isochrones_amd import _cabi, device as dev

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver's
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
        # would, and hand their output (rank 0's one JSON line) through
        raise SystemExit(spawn_ranks(args.gpus))
    # test hooks (single-GPU boxes): ISO_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # ISO_BENCH_BACKEND=gloo replaces RCCL for the collectives
    backend = os.environ.get("ISO_BENCH_BACKEND", "nccl")
    if os.environ.get("ISO_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    t_launch = time.perf_counter()
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    reduce_device = "cuda" if backend == "nccl" else "cpu"

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- start-up: the tables exist once (rank 0 builds them; a real run reads them from disk there) and travel to the
    # other ranks in one broadcast; every rank then builds its own device-resident interpolator
    startup = {"world": world, "backend": (backend if distributed else None)}
    bands = ("V", "G", "BP", "RP")               # V: the cfg-2 star; G, BP, RP: the catalog
    t0 = time.perf_counter()
    ic = ia.synthetic_track(bands=bands) if rank == 0 else None
    startup["table_build_rank0_s"] = time.perf_counter() - t0
    if distributed:
        tb = {}
        t0 = time.perf_counter()
        try:
            ic = ia.broadcast_interpolator(ic, src=0, timings=tb)
            startup.update(tables="broadcast from rank 0", **tb)
        except Exception as e:       # noqa: BLE001 - a rank that could not receive builds its own tables
            startup.update(tables="broadcast failed (%s: %s); built locally" % (type(e).__name__, e))
            ic = ia.synthetic_track(bands=bands)
        startup["broadcast_total_s"] = time.perf_counter() - t0
    mod = ia.SingleStarModel(ic, Teff=(5770, 100), logg=(4.5, 0.1), feh=(0.0, 0.15), V=(10.0, 0.05))
    handle = mod.handle(local_rank)
    lib = _cabi.lib()
    stream = dev.stream_ptr(local_rank)

    # rank r evaluates star r of the catalog: same observables, its own seeded sample batches
    nb = max(1, args.batches)
    batches_host = [make_samples(np.random.default_rng(12345 + 1000 * rank + b), args.n, args.workload) for b in range(nb)]
    rot = Rotation(handle, batches_host, stream)
    mod.lnpost(batches_host[0][:4096])           # first call builds the model's packs (outside every timed region)
    torch.cuda.synchronize()
    startup["ready_s"] = time.perf_counter() - t_launch

    # The first ~10 ms of kernels after a process has set its tables up run 4-8 % slower than the steady state (clock ramp;
    # profiles/r03/launch_duration_trend.txt: 79-83 us per launch falling to 74.7 us over the first ~120 launches).  With
    # the driver's W = 5, K = 20 the whole timed window would sit inside that transient, so a fixed pre-roll of the same
    # rotating launches precedes the W warm-up steps; it is reported (config.preroll_launches) and never timed.
    if args.preroll > 0:
        rot.run(args.preroll)
    if args.warmup > 0:
        rot.run(args.warmup)
    if distributed:
        # untimed, with the warm-up steps: the first collectives of a communicator build its channels (RCCL: lazily,
        # per operation kind), which would otherwise land inside the K-step bracket's closing barrier
        warm = torch.zeros(2, dtype=torch.float64, device=reduce_device)
        for _ in range(2):
            barrier()
            dist.all_reduce(warm, op=dist.ReduceOp.MAX)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = rot.run(args.steps)        # K launches bracketed by HIP events on this stream; returns after the last one
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tmax[0]), float(tmax[1])

    total_evals = float(args.n) * args.steps * world
    device_s = kernel_ms * 1e-3 * args.steps
    value = total_evals / device_s
    bytes_per_launch = BYTES_PER_EVAL_SINGLE_1BAND * args.n
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    label = "cfg2/%s" % args.workload
    rec = pmc_record(label, args.n)
    traffic = rec["fabric_bytes"] if rec else None

    result = {
        "metric": "lnpost evals/sec over MIST grid (10^6-sample batch)",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": kernel_ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "timing": "value = evaluations of all ranks / (K x the slowest rank's mean device time per launch: HIP events "
                  "around the K launches on the launch stream); wall-clock of the barrier + synchronize bracket around "
                  "the same K launches: wall_ms_per_step, value_wall",
        "wall_ms_per_step": elapsed / args.steps * 1e3,
        "value_wall": total_evals / elapsed,
        "config": {"workload": "cfg2: single Sun-like star (Teff/logg/feh + V), track parametrisation, "
                               "synthetic MIST-shaped tables [15,196,1710,18]+[70,26,18,13,%d] (model BC pack: its 1 band), "
                               "%d-sample lnpost batch per GPU, samples '%s', steps rotate over %d distinct batches, "
                               "fused interp+prior+likelihood kernel" % (len(bands), args.n, args.workload, nb),
                   "samples": args.workload, "batch": args.n, "distinct_batches": nb, "preroll_launches": args.preroll,
                   "parallelism": "independent stars per GPU",
                   "kernel_path": os.environ.get("ISOCHRONES_AMD_PATH", "auto")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": (("static: profiles/pmc_traffic.json <- %s (rocprofv3 --pmc passes of this workload, launches "
                                         "rotating over %s batches; not measured in this run)"
                                         % ((_PMC or {}).get("source", "?"), rec.get("distinct_batches", 1))) if rec else None),
                     "kernel_ms": kernel_ms, "bytes_per_eval": BYTES_PER_EVAL_SINGLE_1BAND,
                     "launches_rotate_over_batches": nb,
                     "distinct_table_bytes_touched_per_rotation": (rec.get("fabric_read_bytes") * nb if rec and rec.get("fabric_read_bytes") else None),
                     "bounds": bounds(label, args.n, kernel_ms, bytes_per_launch)},
        # filled in below, placed here so that they sit in the head of the line (the driver's record keeps a bounded prefix):
        # `summary` = the one number of every secondary leg (cfg 3 / cfg 4 / the two catalog legs / the per-point callback)
        "cpu_baseline": None,
        "summary": {},
        "startup": startup,
    }
    summary = result["summary"]
    # the figure of rounds 1-2 beside it: the same batch evaluated again and again (its ~0.48 GB of lines are partly
    # still in the 256 MiB Infinity Cache when the next launch asks for them)
    try:
        rot.run_single(5)
        ms1 = rot.run_single(max(10, args.steps))
        result["roofline"]["single_batch"] = {"kernel_ms": ms1, "achieved": bytes_per_launch / (ms1 * 1e-3) / 1e9,
                                              "frac": bytes_per_launch / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "note": "one batch re-evaluated by every launch (rounds 1-2's headline)"}
    except Exception as e:           # noqa: BLE001
        result["roofline"]["single_batch"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- everything below is secondary: if it stalls, the line above is still printed ---------------------------
    done = threading.Event()
    printed = threading.Lock()

    def emit(note=None):
        with printed:
            if done.is_set():
                return
            done.set()
            if note:
                result["watchdog"] = note
            if rank == 0:
                print(json.dumps(result), flush=True)

    def watchdog():
        if not done.wait(args.extras_timeout):
            emit("secondary legs exceeded %.0f s; printed what was finished" % args.extras_timeout)
            os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()

    if not args.no_extras and not args.no_catalog:
        try:
            result["catalog"] = catalog_leg(ic, rank, world, barrier, dist if distributed else None, reduce_device,
                                            cpu_subsample=(64 if (world == 1 and not args.no_cpu_baseline) else 0))
            for key, leg in result["catalog"].items():
                if key.endswith("_stars") and isinstance(leg, dict) and "stars_per_s" in leg:
                    summary["catalog_32x250_track_%s_per_s" % key] = leg["stars_per_s"]
                    if "speedup_upper_bound" in (leg.get("projection_8gpu") or {}):
                        summary["catalog_32x250_track_projection_8gpu_speedup"] = leg["projection_8gpu"]["speedup_upper_bound"]
        except Exception as e:       # noqa: BLE001 - the extra leg must not take the benchmark line down
            result["catalog"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # The same catalog path ON THE REFERENCE'S OWN WORKLOAD: `starfit` builds get_ichrone(models, bands) = MIST_Isochrone,
        # parameters (eep, age, feh, distance, AV) (isochrones/starfit.py:86, isochrone.py:62-75), and fits with fit_mcmc's
        # defaults nwalkers=300, nburn=200, niter=100 (starmodel.py:889-893): 9 x 10^4 lnpost evaluations per star, 11 x the
        # leg above; 30 000 stored samples per parameter (summaries: k_chain_quantiles_big).  Rank 0 builds the isochrone-grid
        # tables, one broadcast ships them.
        try:
            t0 = time.perf_counter()
            ic_iso = ia.synthetic_isochrone(bands=("G", "BP", "RP")) if rank == 0 else None
            build_s = time.perf_counter() - t0
            tb = {}
            if distributed:
                ic_iso = ia.broadcast_interpolator(ic_iso, src=0, timings=tb)
            ref = catalog_leg(ic_iso, rank, world, barrier, dist if distributed else None, reduce_device, sizes=(10_000,),
                              nwalkers=300, nburn=200, niter=100, timed_passes=3,
                              cpu_subsample=(12 if (world == 1 and not args.no_cpu_baseline) else 0))
            ref["table_build_rank0_s"] = build_s
            ref.update({"broadcast_" + k: v for k, v in tb.items()})
            ref["workload"] = ("the reference's batch_starfit workload: MIST_Isochrone parametrisation, fit_mcmc defaults "
                               "300 walkers x (200 burn-in + 100 kept) iterations, 10^4 stars")
            leg = ref.get("10000_stars") or {}
            if "stars_per_s" in leg:
                summary["catalog_reference_shape_stars_per_s"] = leg["stars_per_s"]        # 300 x (200 + 100), isochrones, 10^4 stars
                summary["catalog_reference_shape_wall_s"] = leg["wall_s"]
                if "speedup_upper_bound" in (leg.get("projection_8gpu") or {}):
                    summary["catalog_reference_shape_projection_8gpu_speedup"] = leg["projection_8gpu"]["speedup_upper_bound"]
                    summary["catalog_reference_shape_1250_stars_fit_s"] = leg["projection_8gpu"]["fit_s_of_one_rank"]
                base_c = (leg.get("cpu_baseline") or {}).get("stars_per_s")
                if base_c:
                    summary["catalog_reference_shape_cpu_one_thread_stars_per_s"] = base_c
            if isinstance(result.get("catalog"), dict):
                result["catalog"]["reference_shape"] = ref
            else:
                result["catalog_reference_shape"] = ref
            ic_iso.release()
            del ic_iso
        except Exception as e:       # noqa: BLE001
            (result["catalog"] if isinstance(result.get("catalog"), dict) else result)["reference_shape_error"] = "%s: %s" % (type(e).__name__, e)
    if world == 1 and not args.no_extras:
        ms = C.c_double()
        # secondary sample distributions, same kernel, rotating in the same way (reported, never `value`)
        extras = {}
        for wl in ("prior", "prior_valid", "posterior"):
            if wl == args.workload:
                continue
            r2 = Rotation(handle, [make_samples(np.random.default_rng(999 + b), args.n, wl) for b in range(nb)], stream)
            r2.run(5)
            k2 = r2.run(max(10, args.steps // 4))
            extras[wl] = {"kernel_ms": k2, "evals_per_s": args.n / (k2 * 1e-3),
                          "finite_fraction": float(torch.isfinite(r2.outs[0]).double().mean()),
                          "roofline": bounds("cfg2/" + wl, args.n, k2, BYTES_PER_EVAL_SINGLE_1BAND * args.n)}
            del r2
        result["other_workloads"] = extras
        # BASELINE configs[2]: binary (two-component flux sum), 6 bands + parallax, isochrone parametrisation, full-size
        # tables - kernel k_lnpost_fast<1, 2, 6, ...>, 2 360 algorithmic B/eval
        try:
            import bench_configs
            ic3, mod3 = bench_configs.cfg3_model()
            h3 = mod3.handle(local_rank)
            cfg3 = {"bytes_per_eval": 2360, "kernel": "k_lnpost_fast<ISO, 2 stars, 6 bands, packed>",
                    "note": "2 360 B is SURVEY 8d's per-evaluation figure; half of the BC bytes are served by L2 / Infinity "
                            "Cache (the six-band BC pack is touched only where stars exist), so algorithmic_GBs is not an "
                            "HBM rate - the memory-side bound is the counter figure in roofline.hbm"}
            first_batch = None
            for wl in ("prior", "prior_valid", "posterior"):
                hosts = [bench_configs.cfg3_samples(args.n, wl, seed=3 + 17 * b) for b in range(nb)]
                r3 = Rotation(h3, hosts, stream)
                if first_batch is None:
                    mod3.lnpost(hosts[0][:4096])
                r3.run(5)
                k3 = r3.run(max(10, args.steps // 4))
                cfg3[wl] = {"kernel_ms": k3, "evals_per_s": args.n / (k3 * 1e-3),
                            "finite_fraction": float(torch.isfinite(r3.outs[0]).double().mean()),
                            "roofline": bounds("cfg3/" + wl, args.n, k3, 2360.0 * args.n)}
                if wl == "prior_valid":
                    first_batch = (hosts[0], r3.outs[0].cpu().numpy())
                del r3, hosts
            if not args.no_cpu_baseline and first_batch is not None:
                base3, ref3 = cpu_baseline(ic3, mod3, first_batch[0], wall_budget_s=3.0, full_passes=2, scalar_calls=5_000)
                got3 = first_batch[1]
                fin = np.isfinite(ref3)
                base3["parity_max_rel_err"] = float(np.max(np.abs(got3[fin] - ref3[fin]) / np.maximum(1.0, np.abs(ref3[fin])))) if fin.any() else 0.0
                base3["parity_pattern_ok"] = bool(np.array_equal(np.isnan(got3), np.isnan(ref3)) and
                                                  np.array_equal(np.isneginf(got3), np.isneginf(ref3)))
                base3["reference_published_us_per_call"] = 719.0
                cfg3["cpu_baseline"] = base3
                cfg3["speedup_vs_cpu"] = cfg3["prior_valid"]["evals_per_s"] / base3["value"]
                cfg3["speedup_vs_cpu_all_cores"] = cfg3["prior_valid"]["evals_per_s"] / base3["value_best_pass"]
                cfg3["speedup_vs_cpu_scalar_calls"] = cfg3["prior_valid"]["evals_per_s"] / base3["modes"]["B1_scalar_call"]["evals_per_s"]
            result["cfg3_binary_6_bands"] = cfg3
            summary["cfg3_binary_6_bands_prior_valid_evals_per_s"] = cfg3["prior_valid"]["evals_per_s"]
            summary["cfg3_kernel_ms"] = cfg3["prior_valid"]["kernel_ms"]
            del mod3, ic3
        except Exception as e:       # noqa: BLE001 - a secondary leg must not take the benchmark line down
            result["cfg3_binary_6_bands"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # BASELINE configs[3]: ensemble MCMC, 256 walkers x 5000 steps on the cfg-2 star with the device-resident sampler
        # (proposal + fused lnpost + accept in one persistent kernel; chain stored), and the same fit on the host
        try:
            from isochrones_amd.sampler import FusedEnsembleSampler
            truth = np.array([1.0, 355.0, 0.0, 100.0, 0.1])
            p0 = truth + np.array([0.01, 2.0, 0.02, 1.0, 0.02]) * np.random.default_rng(1).standard_normal((256, 5))
            p0[:, 4] = np.abs(p0[:, 4])
            fs = FusedEnsembleSampler(mod, 256, seed=2)
            fs.run_mcmc(p0, 50, store=False)
            fs.close()
            walls = []
            for _ in range(3):
                fs = FusedEnsembleSampler(mod, 256, seed=2)           # a fresh sampler: its step counter starts at 0
                torch.cuda.synchronize()
                t_s = time.perf_counter()
                fs.run_mcmc(p0, 5000, store=True)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t_s)
                if len(walls) < 3:
                    fs.close()
            c4 = {"gpu_wall_s": min(walls), "us_per_step": min(walls) / 5000 * 1e6,
                  "lnpost_evals": 256 * 5000, "acceptance": float(fs.acceptance_fraction.mean()),
                  "finite_chain": bool(torch.isfinite(fs._lnprob).all()),
                  "reference_published_estimate_s": 69e-6 * 256 * 5000}
            if not args.no_cpu_baseline:
                gpu_chain = fs.chain_steps.cpu().numpy()                # [5000, 256, 5]
                gpu_lnp = fs._lnprob.cpu().numpy()
                cpu4, cmp4 = cpu_mcmc_baseline(ic, mod, p0, 5000, seed=2, gpu_chain=gpu_chain, gpu_lnp=gpu_lnp)
                c4["cpu_fit"] = cpu4
                c4["cpu_wall_s"] = cpu4["one_walker_per_call"]["wall_s"]
                c4["cpu_kind"] = ("port: oracle/cpu_sampler.py stretch move (same Philox stream as the device sampler) around "
                                  "the oracle's lnpost, one host thread; one_walker_per_call is how emcee drives the reference")
                c4["gpu_vs_cpu_chain"] = cmp4
                c4["speedup_vs_cpu_fit"] = c4["cpu_wall_s"] / c4["gpu_wall_s"]
            # the other 255 CUs: 64 independent 256-walker ensembles of the same star in the same launches
            try:
                E = 64
                fe = FusedEnsembleSampler(mod, 256, seed=2, n_ensembles=E)
                pe = np.broadcast_to(p0, (E, 256, 5)).copy()
                fe.run_mcmc(pe, 20, store=False)
                torch.cuda.synchronize()
                t_s = time.perf_counter()
                fe.run_mcmc(pe, 5000, store=False)
                torch.cuda.synchronize()
                c4["ensembles_64x256x5000"] = {"gpu_wall_s": time.perf_counter() - t_s, "lnpost_evals": E * 256 * 5000}
                fe.close()
            except Exception as e:       # noqa: BLE001
                c4["ensembles_64x256x5000"] = {"error": "%s: %s" % (type(e).__name__, e)}
            result["cfg4_mcmc_256x5000"] = c4
            summary["cfg4_us_per_step"] = c4["us_per_step"]
            summary["cfg4_gpu_wall_s"] = c4["gpu_wall_s"]
            if "cpu_wall_s" in c4:
                summary["cfg4_cpu_wall_s_one_walker_per_call"] = c4["cpu_wall_s"]
            fs.close()
        except Exception as e:       # noqa: BLE001
            result["cfg4_mcmc_256x5000"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # end to end through the host-array API (numpy in, numpy out: H2D of 40 B + D2H of 8 B per sample
        # around the same kernel) - reported for the record, never `value`
        try:
            mod.lnpost(batches_host[0])                  # first call of a size allocates the pinned / device staging
            t_h = time.perf_counter()
            for _ in range(5):
                mod.lnpost(batches_host[0])
            dt_h = (time.perf_counter() - t_h) / 5
            result["host_array_path"] = {"ms": dt_h * 1e3, "evals_per_s": args.n / dt_h,
                                         "note": "mod.lnpost(numpy [N,5]) -> numpy [N], PCIe transfers included"}
        except Exception as e:       # noqa: BLE001
            result["host_array_path"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                base, ref = cpu_baseline(ic, mod, batches_host[0])
                rot.run(nb)                              # every batch's output is current
                got = rot.outs[0][: ref.size].cpu().numpy()
                fin = np.isfinite(ref)
                ok = (np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isneginf(got), np.isneginf(ref)))
                rel = float(np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin])))) if fin.any() else 0.0
                base["parity_max_rel_err"] = rel
                base["parity_pattern_ok"] = bool(ok)
                base["finite_fraction"] = float(fin.mean())
                result["cpu_baseline"] = base
                result["speedup_vs_cpu"] = value / base["value"]
                result["speedup_vs_cpu_all_cores"] = value / base["value_best_pass"]
                summary["speedup_vs_cpu_quota_threads"] = value / base["value"]
                summary["speedup_vs_cpu_best_pass_all_threads"] = value / base["value_best_pass"]
            except Exception as e:   # noqa: BLE001
                result["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        else:
            result["cpu_baseline"] = None
    emit()
    if distributed:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:            # noqa: BLE001
            pass


if __name__ == "__main__":
    main()
