#!/usr/bin/env python
"""Benchmark of the isochrones hot path on MI355X.

Metric (BASELINE.json): lnpost evaluations / second on a 10^6-sample batch over the MIST-shaped
grid.  A "step" is one pass of the fused lnpost kernel over one batch of synthetic samples that
are already resident in HBM.  Workload = BASELINE configs[1]: one Sun-like star
(Teff/logg/feh + V magnitude), evolution-track parametrisation (mass, eep, feh, distance, AV),
full-size synthetic MIST track table [15,196,1710,18] + BC table [70,26,18,13,1].  Default sample
distribution: uniform over the populated part of the table ("prior_valid": uncorrelated gathers
over the whole 1.9 GB packed table, ~98 % of the samples evaluate the complete path); the other two
distributions are timed as well and reported under `other_workloads`.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 1000000] [--workload prior|posterior]

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank evaluates its own
star's batch — independent posteriors, no data-path collective (weak scaling); the only
collectives are the timing barrier and the max-over-ranks reduction.

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` and `cpu_baseline`.
After the timed region (never part of `value`) the same ranks run BASELINE configs[4], the catalog path: a fixed
synthetic catalog split over the ranks with batch_starfit's rule and fitted by the device-resident sampler, reported
as `catalog` (stars/s at this N; strong scaling).  `--no-catalog` / `--no-extras` skip it.
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
    return rec if isinstance(rec, dict) and rec.get("n") == n else None


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
               source="static: profiles/pmc_traffic.json <- profiles/r02/pmc_summary.json (rocprofv3 --pmc passes, "
                      "FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; SQ_ACTIVE_INST_VALU x4 / 1024 SIMDs); kernel time: this run")
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


def cpu_baseline(ic, mod, pars_host, wall_budget_s=4.0):
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
    b2_full = one_thread(n, 3)
    dt1 = time.perf_counter() - t1
    # B1, scalar-call mode: one sample per call through the oracle's C ABI from a Python loop - how the reference
    # is actually driven by its samplers (published there: 69 us per call single star, numba); 10^4 calls
    one = [np.ascontiguousarray(soa[:, k:k + 1]) for k in range(min(n, 10_000))]
    for k in range(200):
        oic.lnpost(desc, one[k % len(one)], nthreads=1, parts=False)
    tc = time.perf_counter()
    for col in one:
        oic.lnpost(desc, col, nthreads=1, parts=False)
    scalar_us = (time.perf_counter() - tc) / len(one) * 1e6
    p1, n1 = 3, n
    return dict(value=n / min(times), unit="evals/s", cores=cores, kind="port", scalar_call_us=scalar_us,
                sample="best of %d passes over the same %d-sample batch (%.1f s wall), C restatement of the reference "
                       "(oracle/iso_oracle.c), OpenMP static over %d threads; container CPU quota: %s; 1-thread "
                       "figure: %d passes over the first %d samples (%.1f s)"
                       % (passes, n, dt, cores, ("%.1f CPUs" % quota) if quota else "none", p1, n1, dt1),
                value_median_pass=n / float(np.median(times)), cpu_quota_cores=quota,
                value_1thread=b2_full,
                modes={"B1_scalar_call": {"us_per_call": scalar_us, "calls": len(one), "evals_per_s": 1e6 / scalar_us},
                       "B2_one_thread_1e4": {"evals_per_s": b2_small}, "B2_one_thread_full_batch": {"evals_per_s": b2_full},
                       "B3_all_cores_full_batch": {"evals_per_s": n / min(times), "median_pass": n / float(np.median(times)),
                                                   "threads": cores}}), out


def catalog_leg(rank, world, barrier, dist, reduce_device, sizes=(10_000, 400_000), nwalkers=32, nburn=150, niter=100):
    """Strong scaling of the catalog path: the whole catalog has a fixed size, rank r fits the stars
    scripts/batch_starfit would give task r (NR % P), and the wall-clock is the slowest rank's.
    Every rank reaches every barrier / reduction whatever happens to its own work (a local failure is recorded
    and reported, it never leaves the other ranks waiting)."""
    import torch
    import isochrones_amd as ia
    from isochrones_amd.catalog import fit_stars_gpu, shard_indices
    bands = ["G", "BP", "RP"]
    err, ic = None, None
    try:
        ic = ia.synthetic_track(bands=bands)
        warm, _ = ia.synthetic_catalog(ic, 64, bands=bands, seed=1, mag_unc=0.01)
        fit_stars_gpu(warm, ic, np.arange(64), nwalkers=nwalkers, nburn=5, niter=5)      # framework-kernel warm-up
    except Exception as e:           # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    out = {"rule": "star i -> rank (i + 1) % P, no collective in the fit", "walkers": nwalkers, "steps": nburn + niter,
           "bands": bands}
    for n_stars in sizes:
        cat, idx, ok = None, np.empty(0, dtype=int), 0.0
        try:
            if err is None:
                cat, _truth = ia.synthetic_catalog(ic, n_stars, bands=bands, seed=7, mag_unc=0.01)
                idx = shard_indices(n_stars, rank, world)
        except Exception as e:       # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        # one untimed pass first, as the W warm-up steps of the metric: the first fit of a size pays for the allocator's
        # first 26 GB of chain storage (hipMalloc + page tables: 0.5 s -> 1.0-1.4 s for the 4 x 10^5-star shard)
        first = None
        for timed in (False, True):
            barrier()
            t0 = time.perf_counter()
            try:
                if err is None:
                    rows = fit_stars_gpu(cat, ic, idx, nwalkers=nwalkers, nburn=nburn, niter=niter, seed=11 + rank)
                    ok = float(np.mean(rows[:, -1] == 1)) if len(rows) else 1.0
            except Exception as e:       # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, e)
            barrier()
            wall = time.perf_counter() - t0
            if not timed:
                first = wall
        stats = torch.tensor([wall, -ok, 1.0 if err is not None else 0.0], dtype=torch.float64, device=reduce_device)
        share = torch.zeros(world, dtype=torch.float64, device=reduce_device)
        share[rank] = float(len(idx))
        if dist is not None:
            dist.all_reduce(stats, op=dist.ReduceOp.MAX)
            dist.all_reduce(share, op=dist.ReduceOp.SUM)
        wall, ok_min, failed = float(stats[0]), -float(stats[1]), bool(stats[2] > 0)
        if failed:
            out["%d_stars" % n_stars] = {"error": err or "a rank other than 0 failed"}
        else:
            out["%d_stars" % n_stars] = {"wall_s": wall, "stars_per_s": n_stars / wall, "first_call_wall_s": first,
                                         "stars_per_rank": int(len(idx)),
                                         "stars_per_rank_all": [int(x) for x in share.tolist()],
                                         "lnpost_evals": int(n_stars) * nwalkers * (nburn + niter), "ok_fraction_min": ok_min}
        cat = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--workload", default="prior_valid", choices=["prior", "prior_valid", "posterior"],
                    help="prior_valid (default): uniform over the populated part of the table, ~98 %% of the samples "
                         "take the full path (every evaluation moves its 560 algorithmic bytes); prior: uniform over "
                         "the table's bounding box (SURVEY 8d (i): ~34 %% fall on NaN padding / are cut by the prior); "
                         "posterior: MCMC-like Gaussian ball (cache resident)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-catalog", action="store_true", help="skip the catalog-sharding leg (BASELINE configs[4])")
    ap.add_argument("--path", default=None, choices=["auto", "compact", "generic"],
                    help="kernel/table-layout selection (default: library default = auto)")
    args = ap.parse_args()

    if args.path:
        os.environ["ISOCHRONES_AMD_PATH"] = args.path
    # the host driver only supports dmabuf IPC: RCCL between the ranks of one node needs this (already exported on the
    # GPU boxes; set here as well so that a bare launch works)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import ctypes as C
    from isochrones_amd import _cabi, device as dev

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # test hooks (single-GPU boxes): ISO_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # ISO_BENCH_BACKEND=gloo replaces RCCL for the timing barrier / max-reduce
    backend = os.environ.get("ISO_BENCH_BACKEND", "nccl")
    if os.environ.get("ISO_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    ic, mod = build_model()
    # rank r evaluates star r of the catalog: same observables, its own seeded sample batch
    rng = np.random.default_rng(12345 + rank)
    pars_host = make_samples(rng, args.n, args.workload)
    pars = torch.as_tensor(np.ascontiguousarray(pars_host.T), device="cuda")     # SoA [5, n], HBM resident
    out = torch.empty(args.n, dtype=torch.float64, device="cuda")
    handle = mod.handle(local_rank)
    lib = _cabi.lib()
    stream = dev.stream_ptr(local_rank)
    ms = C.c_double()

    def run(reps):
        _cabi.check(lib.iso_time_lnpost(handle, dev.ptr(pars), 1, args.n, args.n, dev.ptr(out), reps, stream,
                                        C.byref(ms)))
        return ms.value

    if args.warmup > 0:
        run(args.warmup)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    if distributed:
        # untimed, with the warm-up steps: the first collectives of a communicator build its channels (RCCL: lazily,
        # per operation kind), which would otherwise land inside the K-step bracket's closing barrier
        warm = torch.zeros(2, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        for _ in range(2):
            barrier()
            dist.all_reduce(warm, op=dist.ReduceOp.MAX)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = run(args.steps)            # K launches bracketed by HIP events on this stream
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tmax[0]), float(tmax[1])

    total_evals = float(args.n) * args.steps * world
    value = total_evals / elapsed
    bytes_per_launch = BYTES_PER_EVAL_SINGLE_1BAND * args.n
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    rec = pmc_record("cfg2/%s" % args.workload, args.n)
    traffic = rec["fabric_bytes"] if rec else None

    result = {
        "metric": "lnpost evals/sec over MIST grid (10^6-sample batch)",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "cfg2: single Sun-like star (Teff/logg/feh + V), track parametrisation, "
                               "synthetic MIST-shaped tables [15,196,1710,18]+[70,26,18,13,1], %d-sample "
                               "lnpost batch per GPU, samples '%s', fused interp+prior+likelihood kernel"
                               % (args.n, args.workload),
                   "samples": args.workload, "batch": args.n, "parallelism": "independent stars per GPU",
                   "kernel_path": os.environ.get("ISOCHRONES_AMD_PATH", "auto")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": ("static: profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command, "
                                        "FETCH_SIZE/WRITE_SIZE with the gfx950 corrections; not measured in this run)"
                                        if traffic is not None else None),
                     "kernel_ms": kernel_ms, "bytes_per_eval": BYTES_PER_EVAL_SINGLE_1BAND,
                     "bounds": bounds("cfg2/%s" % args.workload, args.n, kernel_ms, bytes_per_launch)},
    }
    if not args.no_extras and not args.no_catalog:
        # BASELINE configs[4] on the same ranks: a synthetic catalog split over the GPUs with the reference's
        # batch_starfit rule, every rank fitting its stars with the device-resident sampler.  No data-path
        # collective; the same barrier + max-over-ranks timing as above.  Reported next to the metric, never `value`.
        try:
            result["catalog"] = catalog_leg(rank, world, barrier, dist if distributed else None,
                                            "cuda" if backend == "nccl" else "cpu")
        except Exception as e:       # noqa: BLE001 - the extra leg must not take the benchmark line down
            result["catalog"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world == 1 and not args.no_extras:
        # secondary workloads, same kernel, same launch count/3 (reported, never `value`)
        extras = {}
        for wl in ("prior", "prior_valid", "posterior"):
            if wl == args.workload:
                continue
            ph = make_samples(np.random.default_rng(999), args.n, wl)
            pt = torch.as_tensor(np.ascontiguousarray(ph.T), device="cuda")
            o2 = torch.empty(args.n, dtype=torch.float64, device="cuda")
            for reps in (5, max(10, args.steps // 4)):
                _cabi.check(lib.iso_time_lnpost(handle, dev.ptr(pt), 1, args.n, args.n, dev.ptr(o2), reps, stream,
                                                C.byref(ms)))
            extras[wl] = {"kernel_ms": ms.value, "evals_per_s": args.n / (ms.value * 1e-3),
                          "finite_fraction": float(torch.isfinite(o2).double().mean()),
                          "roofline": bounds("cfg2/" + wl, args.n, ms.value, BYTES_PER_EVAL_SINGLE_1BAND * args.n)}
            del pt, o2
        result["other_workloads"] = extras
        # BASELINE configs[2]: binary (two-component flux sum), 6 bands + parallax, isochrone parametrisation, full-size
        # tables - kernel k_lnpost_fast<1, 2, 6, ...>, 2 360 algorithmic B/eval
        try:
            import bench_configs
            ic3, mod3, sets3 = bench_configs.cfg3_model_and_samples(args.n)
            h3 = mod3.handle(local_rank)
            cfg3 = {"bytes_per_eval": 2360, "kernel": "k_lnpost_fast<ISO, 2 stars, 6 bands, packed>"}
            for wl, ph in sets3.items():
                pt = torch.as_tensor(np.ascontiguousarray(ph.T), device="cuda")
                o3 = torch.empty(args.n, dtype=torch.float64, device="cuda")
                for reps in (5, max(10, args.steps // 4)):
                    _cabi.check(lib.iso_time_lnpost(h3, dev.ptr(pt), 1, args.n, args.n, dev.ptr(o3), reps, stream, C.byref(ms)))
                cfg3[wl] = {"kernel_ms": ms.value, "evals_per_s": args.n / (ms.value * 1e-3),
                            "finite_fraction": float(torch.isfinite(o3).double().mean()),
                            "roofline": bounds("cfg3/" + wl, args.n, ms.value, 2360.0 * args.n)}
                del pt, o3
            result["cfg3_binary_6_bands"] = cfg3
            del mod3, ic3, sets3
        except Exception as e:       # noqa: BLE001 - a secondary leg must not take the benchmark line down
            result["cfg3_binary_6_bands"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # BASELINE configs[3]: ensemble MCMC, 256 walkers x 5000 steps on the cfg-2 star with the device-resident sampler
        # (proposal + fused lnpost + accept in one persistent kernel; chain stored)
        try:
            from isochrones_amd.sampler import FusedEnsembleSampler
            truth = np.array([1.0, 355.0, 0.0, 100.0, 0.1])
            p0 = truth + np.array([0.01, 2.0, 0.02, 1.0, 0.02]) * np.random.default_rng(1).standard_normal((256, 5))
            p0[:, 4] = np.abs(p0[:, 4])
            fs = FusedEnsembleSampler(mod, 256, seed=2)
            fs.run_mcmc(p0, 50, store=False)
            walls = []
            for _ in range(3):
                fs.reset()
                torch.cuda.synchronize()
                t_s = time.perf_counter()
                fs.run_mcmc(p0, 5000, store=True)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t_s)
            result["cfg4_mcmc_256x5000"] = {"gpu_wall_s": min(walls), "us_per_step": min(walls) / 5000 * 1e6,
                                            "lnpost_evals": 256 * 5000, "acceptance": float(fs.acceptance_fraction.mean()),
                                            "finite_chain": bool(torch.isfinite(fs._lnprob).all())}
            fs.close()
        except Exception as e:       # noqa: BLE001
            result["cfg4_mcmc_256x5000"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # end to end through the host-array API (numpy in, numpy out: H2D of 40 B + D2H of 8 B per sample
        # around the same kernel) - reported for the record, never `value`
        mod.lnpost(pars_host)                        # first call of a size allocates the pinned / device staging
        t_h = time.perf_counter()
        for _ in range(5):
            mod.lnpost(pars_host)
        dt_h = (time.perf_counter() - t_h) / 5
        result["host_array_path"] = {"ms": dt_h * 1e3, "evals_per_s": args.n / dt_h,
                                     "note": "mod.lnpost(numpy [N,5]) -> numpy [N], PCIe transfers included"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            base, ref = cpu_baseline(ic, mod, pars_host)
            got = out[: ref.size].cpu().numpy()
            fin = np.isfinite(ref)
            ok = (np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isneginf(got), np.isneginf(ref)))
            rel = float(np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin])))) if fin.any() else 0.0
            base["parity_max_rel_err"] = rel
            base["parity_pattern_ok"] = bool(ok)
            base["finite_fraction"] = float(fin.mean())
            result["cpu_baseline"] = base
            result["speedup_vs_cpu_all_cores"] = value / base["value"]
            if "gpu_wall_s" in result.get("cfg4_mcmc_256x5000", {}):
                # the same 1.28 x 10^6 evaluations as one-sample-per-call host calls (how emcee drives the reference)
                c4 = result["cfg4_mcmc_256x5000"]
                c4["cpu_scalar_calls_estimated_s"] = base["scalar_call_us"] * 1e-6 * c4["lnpost_evals"]
                c4["reference_published_estimate_s"] = 69e-6 * c4["lnpost_evals"]
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
