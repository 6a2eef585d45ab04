#!/usr/bin/env python
"""Interleaved A/B of the headline kernel's time: launches rotating over 8 distinct batches vs one batch repeated
(what the 256 MiB Infinity Cache can and cannot hide), alternating the two forms so that clock / thermal drift hits both."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                             # noqa: E402
from isochrones_amd import device as dev                 # noqa: E402

ic, mod = bench.build_model()
hosts = [bench.make_samples(np.random.default_rng(12345 + b), 1_000_000, "prior_valid") for b in range(8)]
mod.lnpost(hosts[0][:4096])
rot = bench.Rotation(mod.handle(0), hosts, dev.stream_ptr(0))
rot.run(40)
rows = []
for rnd in range(8):
    rows.append({"rotating_ms": rot.run(200), "single_ms": rot.run_single(200, b=rnd % 8)})
r = np.array([x["rotating_ms"] for x in rows]); s = np.array([x["single_ms"] for x in rows])
print(json.dumps({"rounds": rows, "rotating_ms_median": float(np.median(r)), "single_ms_median": float(np.median(s)),
                  "rotating_over_single": float(np.median(r) / np.median(s)),
                  "frac_rotating": 560e6 / (float(np.median(r)) * 1e-3) / 8e12, "frac_single": 560e6 / (float(np.median(s)) * 1e-3) / 8e12}))
