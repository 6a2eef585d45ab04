"""bench.py under torch.distributed.run, as the driver launches it for the multi-GPU scaling runs:
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
--gpus N --steps K --warmup W`.  The GPU box of the test suite has one GPU, so the two ranks share it
(ISO_BENCH_SHARE_GPU=1) and the timing collectives use gloo (ISO_BENCH_BACKEND=gloo); everything else - rank
plumbing, per-rank batches, the barrier + max-over-ranks timing, the catalog leg's sharding rule - is the code path
the 8-GPU run takes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, extra):
    env = dict(os.environ, ISO_BENCH_SHARE_GPU="1", ISO_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    # the ranks share the one GPU with this pytest process: give back what earlier tests left in torch's caching allocator
    # (tens of GB after the full-size and catalog tests - eight ranks next to it have run out of device memory before)
    try:
        import gc
        import torch
        gc.collect()
        torch.cuda.empty_cache()
    except Exception:
        pass
    first = None
    for attempt in range(2):
        # (a rendezvous that loses its port between _free_port() and the launcher's bind - another test's ranks, a socket in
        # TIME_WAIT - is tried once more on a fresh port; what the first attempt said is kept for the failure message)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
               "--gpus", str(nproc)] + extra
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        if p.returncode == 0:
            break
        first = first or p.stderr[-3000:]
    assert p.returncode == 0, (first, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]         # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_prints_one_valid_line():
    r = _launch(2, ["--steps", "5", "--warmup", "2"])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["warmup"] == 2
    assert r["scaling"] == "weak" and r["unit"] == "evals/s" and r["dtype"] == "f64" and r["vs_baseline"] is None
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["wall_ms_per_step"] >= 0.9 * r["ms_per_step"]
    # value = the evaluations of BOTH ranks over the slowest rank's device time; the wall-clock bracket beside it
    assert abs(r["value"] - 2 * r["config"]["batch"] * r["steps"] / (r["ms_per_step"] * 1e-3 * r["steps"])) < 1e-6 * r["value"]
    assert 0 < r["value_wall"] <= r["value"] * 1.02
    assert r["config"]["distinct_batches"] == 8
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] <= 1.0
    assert r["cpu_baseline"] is None                 # rank 0 at N = 1 only
    # start-up: rank 0 built the tables, rank 1 received them in one broadcast
    st = r["startup"]
    assert st["world"] == 2 and st["tables"] == "broadcast from rank 0" and st["broadcast_bytes"] > 7e8
    cat = r["catalog"]
    assert "error" not in cat and cat["world"] == 2 and cat["backend"] == "gloo"
    for key, n in (("10000_stars", 10_000), ("400000_stars", 400_000)):
        leg = cat[key]
        assert "error" not in leg, leg
        assert leg["stars_per_s"] > 0 and leg["ok_fraction"] > 0.99
        assert leg["fit_s"] > 0 and leg["gather_s"] > 0 and leg["wall_s"] >= leg["fit_s"]
        # the fit by phase (rank 0's shard): blocks, start points, sampler, summaries (their order means nothing while the ranks share a GPU)
        assert leg["build_s"] > 0 and leg["start_s"] > 0 and leg["sample_s"] > 0 and leg["summary_s"] > 0
        assert leg["build_s"] + leg["start_s"] + leg["sample_s"] + leg["summary_s"] <= leg["fit_s"] * 1.05
        # star i -> rank (i + 1) % 2: each rank owns n // 2 stars, and after the all-gather rank 0 holds every row
        assert leg["stars_per_rank_all"] == [n // 2, n // 2]
        assert leg["rows_gathered_on_rank0"] == n
        # small catalogs are timed five times and the median pass is the one reported
        assert leg["timed_passes"] == (5 if n <= 20_000 else 1) and len(leg["wall_s_all_passes"]) == leg["timed_passes"]
        assert leg["wall_s_all_passes"] == sorted(leg["wall_s_all_passes"])


def _check_reference_shape(cat, world):
    """the catalog leg on the reference's own workload (MIST_Isochrone parametrisation, fit_mcmc's defaults)"""
    ref = cat["reference_shape"]
    assert "error" not in ref and ref["parametrisation"] == ["eep", "age", "feh", "distance", "AV"]
    assert ref["walkers"] == 300 and ref["nburn"] == 200 and ref["niter"] == 100 and ref["world"] == world
    leg = ref["10000_stars"]
    assert "error" not in leg, leg
    assert leg["stars_per_s"] > 0 and leg["ok_fraction"] > 0.99 and leg["rows_gathered_on_rank0"] == 10_000
    assert leg["lnpost_evals"] == 10_000 * 300 * 300
    # (phase clocks of rank 0's shard: all there; their ORDER means nothing while the ranks share one GPU - rank 0's start-point
    # kernel has waited behind seven other ranks' samplers: start_s 46 ms next to sample_s 23 ms in one run of the suite)
    assert leg["build_s"] > 0 and leg["start_s"] > 0 and leg["sample_s"] > 0 and leg["summary_s"] > 0
    assert sum(leg["stars_per_rank_all"]) == 10_000 and max(leg["stars_per_rank_all"]) - min(leg["stars_per_rank_all"]) <= 1
    return leg


def test_bench_eight_ranks_rehearsal_on_one_gpu():
    """The shape of the driver's scaling run at N = 8, rehearsed on the one GPU of this box (eight contexts on cuda:0,
    gloo for the collectives): rendezvous, one broadcast of each table set to seven receivers, 1 250 stars per rank in both
    catalog legs, one JSON line, exit code 0.  (batch_starfit's rule: scripts/batch_starfit:60-62.)"""
    r = _launch(8, ["--steps", "5", "--warmup", "2"])
    assert r["n_gpus"] == 8 and r["steps"] == 5 and r["scaling"] == "weak" and r["value"] > 0
    assert abs(r["value"] - 8 * r["config"]["batch"] / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    st = r["startup"]
    assert st["world"] == 8 and st["tables"] == "broadcast from rank 0"
    cat = r["catalog"]
    assert "error" not in cat and cat["world"] == 8
    for key, n in (("10000_stars", 10_000), ("400000_stars", 400_000)):
        leg = cat[key]
        assert "error" not in leg, leg
        assert leg["stars_per_rank_all"] == [n // 8] * 8 and leg["rows_gathered_on_rank0"] == n and leg["ok_fraction"] > 0.99
    leg = _check_reference_shape(cat, 8)
    assert leg["stars_per_rank_all"] == [1250] * 8


def test_bench_gpus_2_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment (how the driver starts the N = 1 run):
    bench.py starts the ranks itself and still prints exactly one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ISO_BENCH_SHARE_GPU="1", ISO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-catalog"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["startup"]["world"] == 2
    assert r["startup"]["tables"] == "broadcast from rank 0"


def test_bench_single_rank_default_line_has_roofline_and_cpu_baseline():
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert "watchdog" not in r
    assert r["n_gpus"] == 1 and r["config"]["workload"].startswith("cfg2") and r["config"]["distinct_batches"] == 8
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.3 < rf["frac"] <= 1.0
    assert rf["traffic"] is None or (rf["traffic"] > 0 and rf["traffic_source"])
    assert 0.3 < rf["single_batch"]["frac"] <= 1.0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["parity_pattern_ok"]
    assert cb["parity_max_rel_err"] < 1e-9
    assert set(cb["modes"]) == {"B1_scalar_call", "B2_one_thread_1e4", "B2_one_thread_full_batch", "B3_all_cores_full_batch",
                                "B3_quota_threads_full_batch"}
    # `value` is the reproducible figure - threads = min(cores, cgroup quota), median pass; beside it the best pass of all threads
    assert cb["value"] == cb["value_quota"] and cb["cores"] == cb["quota_threads"] >= 1
    assert cb["value_best_pass"] > 0 and cb["value"] <= cb["value_best_pass"] * 1.5
    # the one number of every secondary leg, in the head of the line
    sm = r["summary"]
    assert list(r).index("summary") < list(r).index("startup") and list(r).index("cpu_baseline") < list(r).index("startup")
    for key in ("cfg3_binary_6_bands_prior_valid_evals_per_s", "cfg4_us_per_step", "catalog_reference_shape_stars_per_s",
                "catalog_32x250_track_10000_stars_per_s", "speedup_vs_cpu_quota_threads", "catalog_reference_shape_projection_8gpu_speedup"):
        assert sm[key] > 0, key
    # every secondary line carries what bounds it (memory side or VALU issue), as a fraction that cannot exceed 1
    c3 = r["cfg3_binary_6_bands"]
    assert "error" not in c3
    legs = list(r["other_workloads"].values()) + [c3[w] for w in ("prior", "prior_valid", "posterior")]
    for leg in legs + [{"roofline": rf["bounds"]}]:
        b = leg["roofline"]
        if b["bound"] is None:                       # no counter record for this workload (yet): nothing is claimed
            continue
        assert b["bound"] in ("hbm", "valu") and 0.3 < b["frac"] <= 1.0, b
        assert b["source"].startswith("static") and 0 < b["hbm"]["frac"] <= 1.0 and 0 < b["valu"]["frac"] <= 1.0
    # cfg 3: a measured CPU baseline (B1 / B2 / B3) next to the kernel
    b3 = c3["cpu_baseline"]
    assert b3["kind"] == "port" and b3["parity_pattern_ok"] and b3["parity_max_rel_err"] < 1e-9
    assert c3["speedup_vs_cpu_all_cores"] > 1 and c3["speedup_vs_cpu_scalar_calls"] > 50
    # cfg 4: the fit on the GPU and the SAME fit run on the host (measured, not estimated), chains compared
    c4 = r["cfg4_mcmc_256x5000"]
    assert "error" not in c4 and c4["finite_chain"] and 0.1 < c4["acceptance"] < 0.8 and c4["gpu_wall_s"] < 2.0
    assert c4["cpu_wall_s"] > c4["gpu_wall_s"] and c4["cpu_fit"]["one_walker_per_call"]["lnpost_evals"] == 256 * 5000
    assert c4["gpu_vs_cpu_chain"]["steps_identical_to_1e-9"] >= 100
    assert abs(c4["cpu_fit"]["one_walker_per_call"]["acceptance"] - c4["acceptance"]) < 0.02
    # cfg 5: fit_catalog on this rank + a CPU subsample
    leg = r["catalog"]["10000_stars"]
    assert leg["ok_fraction"] > 0.99 and leg["rows_gathered_on_rank0"] == 10_000
    assert leg["cpu_baseline"]["fitted"] >= 60 and leg["cpu_baseline"]["stars_per_s"] > 0
    assert leg["gpu_over_cpu_one_thread"] if "gpu_over_cpu_one_thread" in leg else leg["cpu_baseline"]["gpu_over_cpu_one_thread"] > 10
    # ... and on the reference's own workload (isochrone parametrisation, 300 walkers x (200 + 100) iterations)
    ref = _check_reference_shape(r["catalog"], 1)
    assert ref["cpu_baseline"]["fitted"] >= 10 and ref["cpu_baseline"]["gpu_over_cpu_one_thread"] > 10
