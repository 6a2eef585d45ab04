#!/usr/bin/env python
"""tools/phase_clock.py for any model shape: shader-clock stamps at the phase boundaries of the last evaluation of a short
single-model run (lane 0 of the workgroup), from an instrumentation build (python tools/build_variant.py phase -DISO_PHASE_CLOCK).
    ISOCHRONES_AMD_LIB=$PWD/isochrones_amd/csrc/libiso_hip_phase.so SHAPES=iso:1:3,iso:2:2 python tools/phase_clock_shape.py [walkers]
With several stars the per-star stamps (2, 3, 5, 6) are those of the LAST star: phase "3" is then the last star's model gather,
"2" everything from the proposal to it (the earlier stars' brackets and gathers included), and so on."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import isochrones_amd as ia  # noqa
from isochrones_amd import _cabi  # noqa
from isochrones_amd.sampler import FusedEnsembleSampler  # noqa
from isochrones_amd.catalog import CatalogPosterior, initial_positions  # noqa

NAMES = ["0 move entered", "1 proposal formed", "2 model brackets (+ earlier stars)", "3 model gather (last star)", "4 priors",
         "5 BC brackets (+ earlier stars' BC and fluxes)", "6 BC gather (last star)", "7 fluxes + likelihood", "8 accept + stores", "9 barrier"]
W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = C.CDLL(_cabi.library_path())
for t in os.environ.get("SHAPES", "iso:1:3,iso:2:2").split(","):
    kind, ns, nb = t.split(":"); ns, nb = int(ns), int(nb)
    bands = list(ia.grids.KNOWN_BANDS[:nb])
    ic = ia.synthetic_track(bands=bands) if kind == "track" else ia.synthetic_isochrone(bands=bands)
    cat, _ = ia.synthetic_catalog(ic, 1, bands=bands, seed=5, mag_unc=0.02, with_parallax=True)
    mod = cat.model(0, ic, N=ns)
    post = CatalogPosterior.from_catalog(cat, ic, N=ns)
    best, lnp, failed = initial_positions(post, W, rng_seed=3, oversample=8, max_tries=4)
    p0 = best[0].cpu().numpy(); post.close()
    fs = FusedEnsembleSampler(mod, W, seed=11)
    rows = []
    for steps in (200, 201, 333):
        fs.reset(); fs.run_mcmc(p0, steps, store=False); torch.cuda.synchronize()
        st = (C.c_ulonglong * 16)()
        assert getattr(lib, 'iso_debug_phase_stamps' if kind == 'track' else 'iso_debug_phase_stamps_iso%d' % ns)(st) == 0
        tt = np.array(st[:10], dtype=np.int64)
        rows.append(np.diff(tt))
    med = np.median(np.array(rows), axis=0).astype(int)
    print(json.dumps({"shape": t, "walkers": W, "star_lanes": os.environ.get("ISOCHRONES_AMD_STAR_LANES", "default"),
                      "half_step_ticks": int(med.sum()), "phase_ticks": dict(zip(NAMES[1:], med.tolist()))}), flush=True)
    fs.close(); ic.release()
