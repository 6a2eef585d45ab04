#!/usr/bin/env python
"""Time iso_chain_quantiles on a parameter-major chain of S stars x 32 walkers x 100 steps x 5 parameters
(what a catalog fit summarises): default dispatch vs the generic wave kernel (ISOCHRONES_AMD_QUANTILES=wave)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from isochrones_amd import _cabi, device as dev
    lib, ctx = _cabi.lib(), dev.context(0)
    out = {}
    q = np.array([0.5, 0.16, 0.84])
    for S, W, T, D in ((100_000, 32, 100, 5), (10_000, 32, 100, 5), (100_000, 16, 100, 5), (50_000, 32, 200, 5), (100_000, 32, 101, 5)):
        rows = S * W
        chain = torch.empty(T, D, rows, dtype=torch.float64, device="cuda")
        chain.normal_()
        chain += torch.arange(D, device="cuda", dtype=torch.float64)[None, :, None]
        if "--correlated" in sys.argv:          # a real chain: a walker's value repeats whenever its move was rejected (~70 %)
            for t in range(1, T):
                keep = torch.rand(D, rows, device="cuda") < 0.7
                chain[t] = torch.where(keep, chain[t - 1], chain[t])
        res = torch.empty(S, D, 3, dtype=torch.float64, device="cuda")
        rec = {}
        for mode in ("auto", "wave"):
            if mode == "auto":
                os.environ.pop("ISOCHRONES_AMD_QUANTILES", None)
            else:
                os.environ["ISOCHRONES_AMD_QUANTILES"] = mode
            def run():
                _cabi.check(lib.iso_chain_quantiles_layout(ctx, dev.ptr(chain), _cabi.CHAIN_PARAM_MAJOR, T, S, W, D,
                                                           q.ctypes.data_as(C.POINTER(C.c_double)), 3, dev.ptr(res), None))
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            rec[mode] = {"ms": ms, "ns_per_pair": ms * 1e6 / (S * D), "chain_TBs": chain.numel() * 8 / (ms * 1e-3) / 1e12,
                         "digest": float(res.sum())}
        os.environ.pop("ISOCHRONES_AMD_QUANTILES", None)
        rec["same_result"] = rec["auto"]["digest"] == rec["wave"]["digest"]
        out["%dx%dx%dx%d" % (S, W, T, D)] = rec
        del chain, res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
