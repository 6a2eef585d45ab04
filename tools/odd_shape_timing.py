#!/usr/bin/env python
"""Timing of the table shapes that used to fall to the generic kernel, fused vs generic (ISOCHRONES_AMD_PATH), on the
full-size tables: the cfg-2 star on a track table whose EEP axis has lost nodes (not uniform any more), and the same
star observed in 13 / 24 / 32 bands.  10^6 prior_valid samples, rotating over 8 batches.  One JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    import isochrones_amd as ia
    from isochrones_amd import device as dev
    n, nb_rot = 1_000_000, 8
    stream = dev.stream_ptr(0)
    batches = [bench.make_samples(np.random.default_rng(12345 + b), n, "prior_valid") for b in range(nb_rot)]
    out = {}

    def timed(mod):
        r = bench.Rotation(mod.handle(0), batches, stream)
        r.run(10)
        return min(r.run(60) for _ in range(3))

    rng = np.random.default_rng(3)
    eeps = np.arange(1.0, 1711.0)
    keep = np.ones(eeps.size, bool)
    keep[rng.choice(np.arange(1, eeps.size - 1), 40, replace=False)] = False
    cases = {"uniform_1band": (None, ("V",)), "thinned_1band": (eeps[keep], ("V",))}
    wide = tuple(list(ia.grids.KNOWN_BANDS) + ["X%02d" % j for j in range(12)])
    for k in (12, 13, 24, 32):
        cases["uniform_%dbands" % k] = (None, wide[:k])
    for name, (axis, bands) in cases.items():
        for path in ("auto", "generic"):
            os.environ["ISOCHRONES_AMD_PATH"] = path
            ic = ia.synthetic_track(bands=bands, eeps=axis) if axis is not None else ia.synthetic_track(bands=bands)
            obs = dict(Teff=(5770, 100), logg=(4.5, 0.1), feh=(0.0, 0.15))
            for j, b in enumerate(bands):
                obs[b] = (10.0 + 0.05 * j, 0.05)
            mod = ia.SingleStarModel(ic, **obs)
            mod.lnpost(batches[0][:4096])
            out.setdefault(name, {})[path] = {"ms": timed(mod), "kernel": mod.kernel_path()}
            del mod, ic
        out[name]["generic_over_fused"] = out[name]["generic"]["ms"] / out[name]["auto"]["ms"]
    os.environ.pop("ISOCHRONES_AMD_PATH", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
