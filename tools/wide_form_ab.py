"""A/B of the wide-pack interp_value kernel's forms on the GPU box: groups of 64 samples per wave (ISOCHRONES_AMD_WIDE_GROUPS)
and the four-pass one-column instantiation (ISOCHRONES_AMD_WIDE_NARROW), interleaved in one process, 10^6 samples on the
MIST-shaped track table; values compared bit for bit with the first form.  One JSON line per (columns, form)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(n=1_000_000, reps=50, rounds=5):
    import torch
    import isochrones_amd as ia
    ic = ia.synthetic_track()
    rng = np.random.default_rng(5)
    lo = np.array([0.1, 1.0, -4.0]); hi = np.array([10.0, 1710.0, 0.5])
    pars = torch.as_tensor(np.ascontiguousarray(rng.uniform(lo, hi, size=(n, 3)).T), device="cuda")
    mi = ic.model_grid.interp
    ci = mi.column_index
    col_sets = {"1_col": np.array([ci["radius"]]), "2_cols": np.array([ci["Teff"], ci["logg"]]),
                "3_cols": np.array([ci["Teff"], ci["logg"], ci["age"]]), "18_cols": np.arange(18)}
    forms = [("g1_u8", {"ISOCHRONES_AMD_WIDE_GROUPS": "1", "ISOCHRONES_AMD_WIDE_NARROW": "0"}),
             ("g1_u4", {"ISOCHRONES_AMD_WIDE_GROUPS": "1", "ISOCHRONES_AMD_WIDE_NARROW": "1"}),
             ("g2_u4", {"ISOCHRONES_AMD_WIDE_GROUPS": "2", "ISOCHRONES_AMD_WIDE_NARROW": "1"}),
             ("g4_u4", {"ISOCHRONES_AMD_WIDE_GROUPS": "4", "ISOCHRONES_AMD_WIDE_NARROW": "1"}),
             ("g8_u4", {"ISOCHRONES_AMD_WIDE_GROUPS": "8", "ISOCHRONES_AMD_WIDE_NARROW": "1"}),
             ("g4_u8", {"ISOCHRONES_AMD_WIDE_GROUPS": "4", "ISOCHRONES_AMD_WIDE_NARROW": "0"}),
             ("default", {})]
    for n_here in (n, 100_000):
        x = [pars[2][:n_here].contiguous(), pars[0][:n_here].contiguous(), pars[1][:n_here].contiguous()]
        for label, cols in col_sets.items():
            times = {f: [] for f, _ in forms}
            ref = None
            same = {}
            for r in range(rounds):
                for f, env in forms:
                    for k in ("ISOCHRONES_AMD_WIDE_GROUPS", "ISOCHRONES_AMD_WIDE_NARROW"):
                        os.environ.pop(k, None)
                    os.environ.update(env)
                    out = mi.interp_device(x, cols); torch.cuda.synchronize()
                    if r == 0:
                        if ref is None:
                            ref = out.clone()
                        same[f] = bool(torch.equal(torch.nan_to_num(out, nan=-1.5e300), torch.nan_to_num(ref, nan=-1.5e300)))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        mi.interp_device(x, cols)
                    e1.record(); torch.cuda.synchronize()
                    times[f].append(e0.elapsed_time(e1) / reps * 1e3)
            for f, _ in forms:
                print(json.dumps({"n": n_here, "columns": label, "form": f, "us_median": float(np.median(times[f])),
                                  "us_min": float(np.min(times[f])), "bit_identical_to_g1_u8": same[f]}), flush=True)


if __name__ == "__main__":
    main()
