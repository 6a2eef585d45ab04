#!/usr/bin/env python
"""Where a half-step of the any-model persistent sampler goes (fast/sampler_any.h around fast/tree_eval.h): shader-clock
stamps of lane 0 of workgroup 0 at the phase boundaries, from an instrumentation build
    python tools/build_variant.py phase --only iso_fast_stretch_tree -DISO_PHASE_CLOCK
    ISOCHRONES_AMD_LIB=$PWD/variants/libs/libiso_hip_phase.so python tools/phase_clock_any.py [walkers]
Model: the resolved binary of docs/multiple.ipynb on the full-size isochrone tables (bench_configs.tree_model_and_samples)."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import bench_configs  # noqa
from isochrones_amd import _cabi  # noqa
from isochrones_amd.sampler import FusedEnsembleSampler  # noqa

NAMES = ["1 move + Philox", "2 params + model brackets (leaf 0)", "3 model gather (leaf 0)", "4 BC brackets (leaf 0)",
         "5 BC gather (leaf 0)", "6 fluxes of leaf 0 + all of the other leaves", "7 priors", "8 likelihood",
         "9 accept + stores", "10 barrier"]
W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = C.CDLL(_cabi.library_path())
mod, _ = bench_configs.tree_model_and_samples(16)
rng = np.random.default_rng(3)
c = np.array([300.0, 280.0, 9.6, 0.0, 400.0, 0.1]); w = np.array([3.0, 3.0, 0.02, 0.02, 5.0, 0.01])
got = []
while len(got) < W:
    x = c + w * rng.standard_normal((4 * W, 6)); x[:, 5] = np.abs(x[:, 5]); x[:, :2] = -np.sort(-x[:, :2], axis=1)
    got.extend(x[np.isfinite(mod.lnpost(x))])
p0 = np.array(got[:W])
fs = FusedEnsembleSampler(mod, W, seed=11)
rows = []
for steps in (200, 201, 333, 334):
    fs.reset(); fs.run_mcmc(p0, steps, store=False); torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)()
    assert lib.iso_debug_phase_stamps_tree(st) == 0
    rows.append(np.diff(np.array(st[:11], dtype=np.int64)))
med = np.median(np.array(rows), axis=0).astype(int)
t0 = __import__("time").perf_counter(); fs.reset(); fs.run_mcmc(p0, 2000, store=False); torch.cuda.synchronize()
us = (__import__("time").perf_counter() - t0) / 2000 * 1e6
print(json.dumps({"model": "tree resolved binary", "walkers": W, "us_per_step": us, "half_step_ticks": int(med.sum()),
                  "phase_ticks": dict(zip(NAMES, med.tolist()))}), flush=True)
