#!/usr/bin/env python
"""With a library built -DISO_DEBUG_PROPOSALS (the persistent sampler stores every PROPOSAL and the lnpost it computed for
it, accepted or not): random models of a pinned shape, each proposal's lnpost against the CPU oracle and against the
batch kernel.  Prints which walkers / lanes are wrong and by how much.
    ISOCHRONES_AMD_LIB=variants/libs/libiso_hip_r987_dbg.so SOAK_KIND=iso SOAK_NSTARS=3 SOAK_NB=9 python tools/probe_proposals.py [seconds] [seed]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import isochrones_amd as ia  # noqa
from isochrones_amd._cabi import IsoError
from isochrones_amd.sampler import FusedEnsembleSampler
from tests import _fixtures as fx
from tests.soak import soak, soak_sampler


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    os.environ["ISOCHRONES_AMD_SAMPLER"] = os.environ.get("PROBE_MODE", "auto")
    t0 = time.time()
    runs = bad_runs = 0
    shown = 0
    codes = {}
    while time.time() - t0 < budget:
        cfg, ic, mod, axes, lo, hi = soak.build(rng)
        W = int(os.environ.get("SOAK_W", rng.choice([4, 8, 16, 30, 64, 100, 256])))
        a = float(rng.choice([1.3, 2.0, 3.0]))
        T = int(rng.integers(3, 12))
        ball = bool(rng.random() < 0.6)
        sseed = int(rng.integers(0, 2 ** 40))
        try:
            fs = FusedEnsembleSampler(mod, W, a=a, seed=sseed)
        except IsoError:
            ic.release(); continue
        p0 = soak_sampler.start_points(rng, mod, lo, hi, W, ball)
        if p0 is None:
            ic.release(); continue
        oic = fx.make_oracle_ic(ic)
        desc = mod.model_desc()
        lnp0 = oic.lnpost(desc, np.ascontiguousarray(p0.T), nthreads=16, parts=False)
        fs.run_mcmc(p0, T, lnprob0=lnp0, store=True)
        y = fs.chain_steps.cpu().numpy()                  # [T, W, D]: the proposals
        got = fs._lnprob.cpu().numpy()                    # [T, W]: lnpost the sampler kernel computed for them
        flat = y.reshape(-1, y.shape[-1])
        want = oic.lnpost(desc, np.ascontiguousarray(flat.T), nthreads=16, parts=False).reshape(got.shape)
        batch = mod.lnpost(torch.as_tensor(flat, device="cuda")).cpu().numpy().reshape(got.shape)
        # the sampler reports -inf where the oracle has NaN or -inf (not finite -> never accepted): compare the finite ones
        fin = np.isfinite(want)
        bad = np.zeros_like(fin)
        bad[fin] = ~np.isclose(got[fin], want[fin], rtol=1e-9, atol=1e-8)
        bad[~fin] = np.isfinite(got[~fin])
        runs += 1
        if os.environ.get("PROBE_KERNARG"):        # -DISO_DEBUG_KERNARG builds: accepted[1] carries a comparison of the re-read argument blocks
            code = int(fs.accepted[1].item())
            codes[(bool(bad.any()), code // 1000000, (code // 1000) % 1000, code % 1000)] = codes.get((bool(bad.any()), code // 1000000, (code // 1000) % 1000, code % 1000), 0) + 1
        if bad.any():
            bad_runs += 1
            if shown < int(os.environ.get("PROBE_SHOW", 12)):
                shown += 1
                t, w = np.nonzero(bad)
                h = W // 2
                print("BAD run: %d of %d proposals wrong; batch kernel wrong on %d of them; cfg %s W=%d a=%g T=%d ball=%d"
                      % (bad.sum(), bad.size, int((~np.isclose(batch[bad], want[bad], rtol=1e-9, atol=1e-8, equal_nan=True)).sum()),
                         json.dumps(cfg), W, a, T, ball), flush=True)
                print("   walkers (row in ensemble) wrong, counts per row:", dict(zip(*np.unique(w, return_counts=True))))
                print("   steps wrong:", dict(zip(*np.unique(t, return_counts=True))))
                for k in range(min(6, t.size)):
                    print("   step %d row %d (half %d, k %d): got %.17g want %.17g diff %.6g  y=%s" % (
                        t[k], w[k], w[k] // h, w[k] % h, got[t[k], w[k]], want[t[k], w[k]], got[t[k], w[k]] - want[t[k], w[k]],
                        np.array2string(y[t[k], w[k]], precision=6)), flush=True)
                # are the wrong ones different in some parameter range?
                ok_rows = fin & ~bad
                print("   finite proposals: %d, wrong: %d; fraction of non-finite oracle values in the run: %.2f" % (fin.sum(), bad.sum(), 1 - fin.mean()))
        del fs
        ic.release()
    print("probe: %d runs, %d with wrong proposal values, %.0f s" % (runs, bad_runs, time.time() - t0))
    if codes:
        print("(bad run, FastArgs dwords that differ, StretchArgs dwords that differ, first differing dword or 999): runs")
        for k in sorted(codes):
            print("  ", k, codes[k])


if __name__ == "__main__":
    main()
