#!/usr/bin/env python
"""Where a lone workgroup's half-step goes (cfg 4: one star, 256 walkers): shader-clock stamps at the phase boundaries
of the last evaluation of a short run, from an instrumentation build of the library.

    python tools/build_variant.py phase -DISO_PHASE_CLOCK
    ISOCHRONES_AMD_LIB=$PWD/isochrones_amd/csrc/libiso_hip_phase.so python tools/phase_clock.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["0 move entered", "1 proposal formed (Philox, partner, y)", "2 model brackets", "3 model gather (coop_star)",
         "4 priors", "5 BC brackets", "6 BC gather (coop_bc)", "7 likelihood", "8 accept + stores", "9 barrier"]


def main():
    import torch
    import bench
    from isochrones_amd import _cabi
    from isochrones_amd.sampler import FusedEnsembleSampler
    ic, mod = bench.build_model()
    truth = np.array([1.0, 355.0, 0.0, 100.0, 0.1])
    p0 = truth + np.array([0.01, 2.0, 0.02, 1.0, 0.02]) * np.random.default_rng(1).standard_normal((256, 5))
    p0[:, 4] = np.abs(p0[:, 4])
    fs = FusedEnsembleSampler(mod, 256, seed=2)
    lib = C.CDLL(_cabi.library_path())
    out = []
    for steps in (200, 201, 333):
        fs.reset()
        fs.run_mcmc(p0, steps, store=False)
        torch.cuda.synchronize()
        st = (C.c_ulonglong * 16)()
        rc = lib.iso_debug_phase_stamps(st)
        assert rc == 0, rc
        t = np.array(st[:10], dtype=np.int64)
        inner = np.array(st[10:15], dtype=np.int64)
        out.append({"steps": steps, "coop_star_from_phase_start(request published, loads issued + weights, first piece in, last piece in, responses written)": (inner - t[2]).tolist(), "ticks_from_entry": (t - t[0]).tolist(), "phase_ticks": dict(zip(NAMES[1:], np.diff(t).tolist()))})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
