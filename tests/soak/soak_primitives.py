"""Randomised soak of the path's smaller primitives against the CPU oracle:
  * get_eep / interp_eep (reference interp.py:488-558, models.py:501-542) on random ragged age tables - random axis
    lengths, track lengths from 0 to the full EEP range, queries on nodes, outside the table, NaN;
  * the unit-cube transform mnest_prior (reference starmodel.py:1637-1650) of random BasicStarModel configurations
    (tests/soak/soak.py's generator), device batch form and host scalar form;
  * the host-array entry point iso_lnpost_host at random sizes across its three regimes, against the device entry point.
Usage on the GPU box: python tests/soak/soak_primitives.py [seconds] [seed].  Exit code 1 on any mismatch."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import isochrones_amd as ia
from isochrones_amd.interp import DFInterpolator
from isochrones_amd.models import EvolutionTrackGrid, EvolutionTrackInterpolator, BolometricCorrectionGrid
from oracle import oracle as orc
from tests.soak import soak

_BC = []


def eep_case(rng):
    nf, nm, ne = int(rng.integers(2, 9)), int(rng.integers(2, 40)), int(rng.integers(8, 300))
    fehs = np.sort(rng.choice(np.arange(-40, 6) * 0.1, nf, replace=False))
    masses = np.sort(rng.choice(np.arange(1, 400) * 0.02, nm, replace=False))
    eeps = np.arange(1.0, ne + 1.0)
    grid, ax, cols = ia.grids.synthetic_track_grid(fehs, masses, eeps)
    ages = np.full((nf, nm, ne), np.nan)
    lengths = rng.integers(0, ne + 1, size=(nf, nm))
    lengths[rng.random((nf, nm)) < 0.5] = ne
    for i in range(nf):
        for j in range(nm):
            L = int(lengths[i, j])
            if L:
                steps = rng.uniform(0.0, 0.05, L)
                steps[rng.random(L) < 0.05] = 0.0                    # repeated ages (plateaus) as in real tracks
                ages[i, j, :L] = rng.uniform(5.0, 8.0) + np.cumsum(steps)
    grid[..., cols.index("age")] = ages
    if not _BC:
        bcg, bax, bands = ia.grids.synthetic_bc_grid(("G",))
        _BC.append((BolometricCorrectionGrid(DFInterpolator.from_arrays(bcg, bax, bands), bands=bands), bands))
    ic = EvolutionTrackInterpolator(EvolutionTrackGrid(DFInterpolator.from_arrays(grid, ax, cols)), _BC[0][0], bands=_BC[0][1])
    n = 20_000
    m = rng.uniform(masses[0] - 0.1, masses[-1] + 0.1, n)
    f = rng.uniform(fehs[0] - 0.1, fehs[-1] + 0.1, n)
    a = rng.uniform(4.5, 12.0, n)
    k = n // 10
    m[:k] = rng.choice(masses, k); f[k:2 * k] = rng.choice(fehs, k)
    fin = ages[np.isfinite(ages)]
    if fin.size:
        a[2 * k:3 * k] = rng.choice(fin, k)                          # exact table ages
    for col in (m, f, a):
        col[rng.integers(0, n, 5)] = np.nan
        col[rng.integers(0, n, 2)] = np.inf
    got = ic.get_eep(m, a, f)                  # (builds the ragged age arrays the oracle is handed below)
    want = orc.interp_eep(a, f, m, fehs, masses, ic._age_grid, ic._array_lengths)
    got_dev = ic.get_eep(torch.as_tensor(m, device="cuda"), torch.as_tensor(a, device="cuda"),
                         torch.as_tensor(f, device="cuda")).cpu().numpy()
    cfg = dict(what="interp_eep", nf=nf, nm=nm, ne=ne)
    bad = (soak.same(got, want, "get_eep host arrays", cfg) >= 0) + (soak.same(got_dev, want, "get_eep device", cfg) >= 0)
    # ONE point at a time through the context's resident service wave (kernels/k_service.h): the batch's numbers bit for bit -
    # get_eep, interp_value (two column lists, alternating: the wave restages the axes when the target changes) and interp_mag
    idx = rng.integers(0, n, 24)
    one = np.array([ic.get_eep(float(m[i]), float(a[i]), float(f[i])) for i in idx])
    if not np.array_equal(one, got[idx], equal_nan=True):
        print("MISMATCH get_eep one point", json.dumps(cfg), flush=True)
        bad += 1
    e = rng.uniform(eeps[0] - 2, eeps[-1] + 2, idx.size)
    for cols_ in (["Teff", "logg", "age"], ["mass"], list(cols)):
        batch = np.atleast_2d(ic.interp_value([m[idx], e, f[idx]], cols_))
        pts = np.array([np.asarray(ic.interp_value([float(m[i]), float(ee), float(f[i])], cols_)) for i, ee in zip(idx, e)])
        if not np.array_equal(pts.reshape(batch.shape), batch, equal_nan=True):
            print("MISMATCH interp_value one point", cols_[:3], json.dumps(cfg), flush=True)
            bad += 1
    d_, av = rng.uniform(5.0, 3000.0, idx.size), rng.uniform(-0.1, 1.2, idx.size)
    bT, bg, bf, bm = ic.interp_mag([m[idx], e, f[idx], d_, av], ["G"])
    for k_, i in enumerate(idx):
        T1, g1, f1, m1 = ic.interp_mag([float(m[i]), float(e[k_]), float(f[i]), float(d_[k_]), float(av[k_])], ["G"])
        if not (np.array_equal([T1, g1, f1], [bT[k_], bg[k_], bf[k_]], equal_nan=True) and np.array_equal(np.asarray(m1), np.asarray(bm[k_]), equal_nan=True)):
            print("MISMATCH interp_mag one point", json.dumps(cfg), flush=True)
            bad += 1
            break
    ic.release()
    return bad, 2 * n + 6 * idx.size


def cube_case(rng):
    cfg, ic, mod, axes, lo, hi = soak.build(rng)
    n = 5000
    cube = rng.random((n, lo.size))
    cube[:4] = [[0.0] * lo.size, [1.0] * lo.size, [0.5] * lo.size, [np.nan] * lo.size]
    want = orc.unit_cube(mod.model_desc(), ic.kind, cube.copy())
    dev = torch.as_tensor(cube, device="cuda")
    mod.mnest_prior(dev)
    cfg["what"] = "unit cube"
    bad = soak.same(dev.cpu().numpy().ravel(), want.ravel(), "mnest_prior device", cfg) >= 0
    for r in range(6):
        row = list(cube[r])
        mod.mnest_prior(row, None, None)
        bad += soak.same(np.array(row), want[r], "mnest_prior scalar", cfg) >= 0
    # host-array lnpost: the three size regimes of iso_lnpost_host against the device entry point
    for size in (int(rng.integers(1, 257)), int(rng.integers(257, 32769)), int(rng.integers(32769, 300_000))):
        x = soak.samples(rng, axes, lo, hi, max(size, 64))[:size]
        want_l = mod.evaluate_device(torch.as_tensor(x, device="cuda")).cpu().numpy()
        if not np.array_equal(mod.lnpost(x), want_l, equal_nan=True):
            print("MISMATCH host-array lnpost", size, json.dumps(cfg), flush=True)
            bad += 1
    ic.release()
    return int(bad), n * lo.size


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    os.environ["ISOCHRONES_AMD_PATH"] = "auto"
    t0 = time.time()
    cases = {"interp_eep": 0, "unit cube + host arrays": 0}
    fails = vals = 0
    while time.time() - t0 < budget:
        if rng.random() < 0.5:
            b, v = eep_case(rng); cases["interp_eep"] += 1
        else:
            b, v = cube_case(rng); cases["unit cube + host arrays"] += 1
        fails += b; vals += v
    print("primitives soak: %s cases, %.3g values compared, %d mismatching checks, %.0f s" % (cases, vals, fails, time.time() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
