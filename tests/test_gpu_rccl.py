"""The RCCL route of the catalog path on the one GPU of the test box: a world_size-1 `nccl` process group (backend
"nccl" is RCCL on ROCm) through which `broadcast_interpolator` and `fit_catalog` run their collectives - the broadcast
of the tables, the all-gather of the result rows, the error-flag reduction.  The sharding logic itself is covered with
world_size-2 gloo groups in tests/test_sampler_catalog_cpu.py; what this adds is that librccl is loaded and executes
this build's calls (scripts/batch_starfit:60-62 is the reference's sharding rule; SURVEY 8e names the two collectives)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
import isochrones_amd as ia
from isochrones_amd.catalog import fit_stars_gpu

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
fehs = np.array([-1.0, -0.5, 0.0, 0.5]); masses = np.array([0.7, 0.9, 1.0, 1.1, 1.3, 2.0]); eeps = np.arange(250.0, 460.0)
ic0 = ia.synthetic_track(bands=("G", "BP", "RP"), fehs=fehs, masses=masses, eeps=eeps,
                         limits=dict(mass=(0.7, 2.0), feh=(-1.0, 0.5), age=(5, 10.13)), eep_bounds=(250, 459))
# 1. the tables travel through RCCL broadcasts (device buffers) and the interpolator is rebuilt from what arrived
tm = {}
ic = ia.broadcast_interpolator(ic0, src=0, rebuild_on_src=True, timings=tm)
assert ic is not ic0
out["broadcast"] = tm
# ... and STAY on the device: the tensors that arrived are what the library's tables are made from (one device-to-device copy,
# iso_table_create_from_device); no host copy exists until somebody asks for `.grid`
assert tm["tables_stay_on_device"] is True
for a in (ic.model_grid.interp, ic.bc_grid.interp):
    assert a._device_grid is not None and a._device_grid.is_cuda and a._grid is None
assert ic.bands == ic0.bands and tuple(ic.eep_bounds) == tuple(ic0.eep_bounds) and ic.kind == ic0.kind
pts = np.random.default_rng(1).uniform([0.75, 260.0, -0.9, 50.0, 0.0], [1.9, 450.0, 0.4, 500.0, 0.5], (2000, 5))
got = ic.interp_mag([pts[:, j] for j in range(5)], ["G", "BP", "RP"])
ref = ic0.interp_mag([pts[:, j] for j in range(5)], ["G", "BP", "RP"])
assert all(np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True) for x, y in zip(got, ref))     # same tables, bit for bit
assert ic.model_grid.interp._grid is None                                # (still no download)
for a, b in ((ic.model_grid.interp, ic0.model_grid.interp), (ic.bc_grid.interp, ic0.bc_grid.interp)):
    assert np.array_equal(a.grid, b.grid, equal_nan=True) and list(a.columns) == list(b.columns)     # the lazy host copy
    assert all(np.array_equal(x, y) for x, y in zip(a.index_columns, b.index_columns))
# the host round trip of rounds 1-5 on request: the same interpolator
os.environ["ISOCHRONES_AMD_BROADCAST"] = "host"
tmh = {}
ich = ia.broadcast_interpolator(ic0, src=0, rebuild_on_src=True, timings=tmh)
os.environ.pop("ISOCHRONES_AMD_BROADCAST")
assert tmh["tables_stay_on_device"] is False and ich.model_grid.interp._device_grid is None
out["broadcast_host_round_trip"] = tmh
# 2. fit_catalog: the shard (all stars at world 1) is fitted on the device, the rows go through all_gather_into_tensor
rng = np.random.default_rng(3)
cat, truth = ia.synthetic_catalog(ic, 96, bands=["G", "BP", "RP"], seed=5, mag_unc=0.01)
res = ia.fit_catalog(cat, ic, strict=True, nwalkers=32, nburn=60, niter=40, seed=9)
tmg = res.attrs["timings"]
assert tmg["backend"] == "nccl" and tmg["world"] == 1 and tmg["stars_of_this_rank"] == 96
direct = fit_stars_gpu(cat, ic, np.arange(96), nwalkers=32, nburn=60, niter=40, seed=9)
assert np.array_equal(res.values, direct, equal_nan=True)          # gathered rows = the rows of the fit, bit for bit
out["ok_fraction"] = float((res["ok"] == 1).mean())
out["timings"] = {k: v for k, v in tmg.items() if isinstance(v, (int, float, str))}
# 3. a failing shard still reaches the collectives and reports through them (strict=False isolation at world 1)
import warnings
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    bad = ia.fit_catalog(cat, ic, strict=False, fit_fn=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("boom")))
assert bad["ok"].eq(0).all() and "boom" in bad.attrs["shard_errors"][0]
# 4. the resident waves of the per-point calls next to RCCL collectives in the same process: a scalar accessor and a model's
# lnpost(p) start their waves (kernels/k_service.h, iso_fast_mailbox.hip), collectives run while they are resident, the
# calls give the same numbers afterwards
import time
mod1 = cat.model(0, ic)
p1 = [1.0, 355.0, 0.0, 300.0, 0.1]
v0, l0 = ic.interp_value(p1[:3], ["Teff", "logg"]), mod1.lnpost(p1)
worst = 0.0
for _ in range(20):
    assert np.array_equal(ic.interp_value(p1[:3], ["Teff", "logg"]), v0) and mod1.lnpost(p1) == l0
    t0 = time.perf_counter()
    x = torch.ones(1 << 16, device="cuda")
    dist.all_reduce(x)
    dist.barrier()
    torch.cuda.synchronize()
    worst = max(worst, time.perf_counter() - t0)
    assert float(x[0]) == 1.0
out["collective_next_to_resident_waves_worst_s"] = worst
assert worst < 0.25
# which librccl the process has mapped
maps = open("/proc/self/maps").read()
out["rccl_mapped"] = sorted({ln.split()[-1] for ln in maps.splitlines() if "rccl" in ln.lower()})
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world1_nccl_group_runs_broadcast_and_fit_catalog(tmp_path):
    script = tmp_path / "rccl_world1.py"
    script.write_text(SCRIPT % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["ok_fraction"] > 0.9
    assert out["rccl_mapped"], "no librccl in the process map: the nccl backend did not load RCCL"
    assert out["broadcast"]["broadcast_bytes"] > 0 and out["broadcast"]["broadcast_s"] > 0


# ---------------------------------------------------------------------------------------------------------------
# The RECEIVING side of the device route with more than one rank.  RCCL does not let two ranks share a GPU, and the test
# box has one; gloo broadcasts CUDA tensors (staged through the host), so with ISOCHRONES_AMD_BROADCAST=device two ranks
# on cuda:0 run exactly what a rank of the 8-GPU job runs after its `dist.broadcast` returned: the tensor that arrived is
# handed to the library as it is (DFInterpolator.from_device -> iso_table_create_from_device), the host copy behind
# `.grid` is made only on request, and the interpolator built from it evaluates the same bits as the sender's.
# ---------------------------------------------------------------------------------------------------------------
SCRIPT_WORLD2 = r'''
import hashlib, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
import isochrones_amd as ia
ic = ia.synthetic_track(bands=("G", "BP", "RP")) if rank == 0 else None
tb = {}
ic = ia.broadcast_interpolator(ic, src=0, timings=tb)
m = ic.model_grid.interp
out = {"rank": rank, "stay": tb["tables_stay_on_device"], "bytes": tb["broadcast_bytes"],
       "device_grid": getattr(m, "_device_grid", None) is not None, "host_copy_before": getattr(m, "_grid", None) is not None}
mod = ia.SingleStarModel(ic, Teff=(5770, 100), logg=(4.4, 0.1), feh=(0.0, 0.15), G=(10.0, 0.05), parallax=(10.0, 0.1))
rng = np.random.default_rng(5)
x = np.column_stack([rng.uniform(0.7, 2.0, 20000), rng.uniform(250, 500, 20000), rng.uniform(-1, 0.4, 20000),
                     rng.uniform(50, 150, 20000), rng.uniform(0, 1, 20000)])
post = mod.lnpost(x)
out["finite"] = int(np.isfinite(post).sum())
out["lnpost_digest"] = hashlib.sha256(np.ascontiguousarray(post).tobytes()).hexdigest()
mags = ic.interp_mag([1.0, 355.0, 0.0, 100.0, 0.1], ["G", "BP", "RP"])
out["mag_digest"] = hashlib.sha256(np.ascontiguousarray(np.concatenate([np.ravel(v) for v in mags])).tobytes()).hexdigest()
# a catalog shard fitted on what arrived
cat, _ = ia.synthetic_catalog(ic, 64, bands=["G", "BP", "RP"], seed=3, mag_unc=0.01)
res = ia.fit_catalog(cat, ic, nwalkers=32, nburn=40, niter=20, seed=2)
out["rows"] = len(res) if res is not None else None
out["ok"] = float(np.mean(res["ok"])) if res is not None else None
out["host_copy_after_fit"] = getattr(m, "_grid", None) is not None
out["grid_digest"] = hashlib.sha256(np.ascontiguousarray(m.grid).tobytes()).hexdigest()       # (asks for the host copy)
gathered = [None, None]
dist.all_gather_object(gathered, out)
if rank == 0:
    print("RESULT " + json.dumps(gathered))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_receive_the_tables_on_the_device(tmp_path):
    script = tmp_path / "bcast_world2.py"
    script.write_text(SCRIPT_WORLD2 % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", ISOCHRONES_AMD_BROADCAST="device")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), str(script)], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    r0, r1 = sorted(json.loads(line[len("RESULT "):]), key=lambda d: d["rank"])
    assert r0["stay"] and r1["stay"] and r1["bytes"] > 7e8
    assert r1["device_grid"] and not r1["host_copy_before"]          # what arrived was adopted where it was
    assert not r1["host_copy_after_fit"]                              # evaluating and fitting never asked for a host copy
    assert r0["finite"] > 1000
    for key in ("finite", "lnpost_digest", "mag_digest", "grid_digest"):
        assert r0[key] == r1[key], key                                # the receiver evaluates the sender's bits
    assert r0["rows"] == 64 and r0["ok"] > 0.9
