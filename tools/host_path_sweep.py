"""A/B of iso_lnpost_host's large-batch pipeline: chunk size sweep, min / median of 30 calls each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ic, mod = bench.build_model()
big = bench.make_samples(np.random.default_rng(1), 1_000_000, "prior_valid")
mod.lnpost(big)
for chunk in (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20):
    os.environ["ISO_PIPE_CHUNK"] = str(chunk)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); r = mod.lnpost(big); ts.append(time.perf_counter() - t0)
    print("chunk %8d rows: min %.3f ms  median %.3f ms" % (chunk, min(ts) * 1e3, np.median(ts) * 1e3), flush=True)
out = np.empty(1_000_000)
import ctypes as C
from isochrones_amd import _cabi, device as dev
h = mod.handle(0)
for chunk in (1 << 17, 1 << 18, 1 << 19):
    os.environ["ISO_PIPE_CHUNK"] = str(chunk)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); _cabi.lib().iso_lnpost_host(h, big.ctypes.data, 1_000_000, out.ctypes.data, None, None); ts.append(time.perf_counter() - t0)
    print("C call, reused output, chunk %8d: min %.3f ms  median %.3f ms" % (chunk, min(ts) * 1e3, np.median(ts) * 1e3), flush=True)
