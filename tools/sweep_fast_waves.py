#!/usr/bin/env python
"""Time the fused batch kernel of every (stars, bands) shape of a multiple system on the full-size isochrone tables, for
the library selected by ISOCHRONES_AMD_LIB (variants built with -DISO_FAST_WAVES_MULTI=2|3|4, tools/build_variant.py
--only iso_fast_iso2,iso_fast_iso3): one JSON line per shape with the kernel time on a 10^6-row batch of samples spread
over the populated table (memory-bound) and of a posterior-like ball (cache-resident, VALU-bound).
    python tools/sweep_fast_waves.py [--n 1000000] [--shapes 2:6,3:9,...]"""
import argparse, ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--shapes", default=",".join("%d:%d" % (ns, nb) for ns in (2, 3) for nb in range(1, 13)))
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    import torch
    import isochrones_amd as ia
    import bench_configs as bc
    from isochrones_amd import _cabi, device as dev
    lib = _cabi.lib()
    tag = os.path.basename(os.environ.get("ISOCHRONES_AMD_LIB", "default"))
    n = args.n
    base = {wl: bc.cfg3_samples(n, wl, seed=5) for wl in ("prior_valid", "posterior")}
    for shape in args.shapes.split(","):
        ns, nb = (int(v) for v in shape.split(":"))
        bands = ia.grids.KNOWN_BANDS[:nb]
        ic = ia.synthetic_isochrone(bands=bands)
        obs = {b: (10.0 + 0.1 * j, 0.02) for j, b in enumerate(bands)}
        mod = ia.BasicStarModel(ic, N=ns, parallax=(2.0, 0.05), **obs)
        rec = {"lib": tag, "stars": ns, "bands": nb, "path": mod.kernel_path()}
        for wl, two in base.items():
            if ns == 2:
                pars = two
            else:      # a third, fainter component below the second
                e2 = two[:, 1] - np.abs(np.random.default_rng(7).normal(20.0, 10.0, n))
                pars = np.column_stack([two[:, 0], two[:, 1], np.maximum(e2, 1.0), two[:, 2:]])
            pt = torch.as_tensor(np.ascontiguousarray(pars.T), device="cuda")
            out = torch.empty(n, dtype=torch.float64, device="cuda")
            h = mod.handle(0)
            ms = C.c_double()
            for r in (5, args.reps):
                _cabi.check(lib.iso_time_lnpost(h, dev.ptr(pt), 1, n, n, dev.ptr(out), r, dev.stream_ptr(), C.byref(ms)))
            rec[wl + "_us"] = ms.value * 1e3
            rec[wl + "_finite"] = float(torch.isfinite(out).double().mean())
        print(json.dumps(rec), flush=True)
        del mod
        ic.release()


if __name__ == "__main__":
    main()
