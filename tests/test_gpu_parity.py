"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI
by the host layer, against (a) the committed golden vectors produced by the reference itself and
(b) the CPU oracle on seeded random inputs.

Tolerance: BASELINE.json's north_star asks for <= 1e-6 relative vs the reference CPU path; the
kernels are float64 like the reference, so the tests hold them to RTOL = 1e-9 (differences come
only from FMA contraction and device-libm ulps).  NaN / -inf patterns must match exactly."""
import os
import sys

import numpy as np
import pytest

import isochrones_amd as ia
from isochrones_amd.interp import DFInterpolator
from tests import _fixtures as fx

pytestmark = pytest.mark.gpu

NORTH_STAR_RTOL = 1e-6
RTOL = 1e-9
ATOL = 1e-11


def test_library_loaded_and_version():
    from isochrones_amd import _cabi
    assert b"gfx950" in _cabi.lib().iso_version()


def test_kats_3d_2d_4d():
    k = fx.load("interp_kats")
    t3 = DFInterpolator.from_arrays(k["t3_grid"], [k["t3_axes0"], k["t3_axes1"], k["t3_axes2"]], ["val"])
    pts = k["t3_pts"]
    got = t3([pts[:, 0], pts[:, 1], pts[:, 2]], ["val"])
    fx.assert_close(got, k["t3_vals"], RTOL, what="3d")
    # reference tests/test_interp.py:31,35 — scalar call form, exact at a node
    assert t3([6.0, 50.0, 200.0], ["val"])[0] == 6.0 ** 2 * np.cos(5.0) + 200.0
    assert np.isclose(t3([3.1, 44.0, 503.0], ["val"])[0], k["t3_scalar"][1], rtol=RTOL)
    axes2 = [k["t2_axes0"], k["t2_axes1"]]
    full = DFInterpolator.from_arrays(k["t2_grid"], axes2, ["sum", "product", "power"])
    miss = DFInterpolator.from_arrays(k["t2_grid_missing"], axes2, ["sum", "product", "power"])
    assert np.allclose(full([1.4, 2.1]), k["t2_doc_cell3"], atol=1e-12)          # docs/interpolate.ipynb cell 3
    assert np.allclose(full([2.2, 4.6], ["product"]), k["t2_doc_cell5"], atol=1e-12)   # cell 5
    assert np.allclose(miss([1.3, 2.2]), k["t2_doc_cell12"], atol=1e-12)         # cell 12
    assert np.all(np.isnan(miss([2.3, 3.0])))                                    # cell 14
    q = k["t2_pts"]
    fx.assert_close(full([q[:, 0], q[:, 1]]), k["t2_vals"], RTOL, what="2d")
    fx.assert_close(miss([q[:, 0], q[:, 1]]), k["t2_vals_missing"], RTOL, what="2d missing")
    t4 = DFInterpolator.from_arrays(k["t4_grid"], [k["t4_axes%d" % i] for i in range(4)], ["a", "b", "c"])
    p = k["t4_pts"]
    fx.assert_close(t4([p[:, i] for i in range(4)], ["c", "a"]), k["t4_vals"], RTOL, atol=1e-13, what="4d")


def test_dfinterpolator_from_dataframe_ragged():
    """DataFrame constructor incl. the NaN-padding of a ragged index (interp.py:590-614)."""
    import itertools
    import pandas as pd
    x, y = np.arange(1, 4), np.arange(1, 6)
    index = pd.MultiIndex.from_product((x, y), names=["x", "y"])
    df = pd.DataFrame(index=index)
    df["sum"] = [a + b for a, b in itertools.product(x, y)]
    df["product"] = [a * b for a, b in itertools.product(x, y)]
    df["power"] = [a ** b for a, b in itertools.product(x, y)]
    k = fx.load("interp_kats")
    d = DFInterpolator(df)
    assert np.array_equal(d.grid, k["t2_grid"])
    dm = DFInterpolator(df.drop([(3, 3), (3, 4)]))
    assert np.array_equal(dm.grid, k["t2_grid_missing"], equal_nan=True)
    assert np.allclose(dm([1.3, 2.2]), [3.5, 2.86, 2.14], atol=1e-12)


@pytest.fixture(params=["auto", "compact", "generic"])
def kernel_path(request, monkeypatch):
    """lnpost kernel selection (read by libiso_hip when an interpolator / model is created):
    auto = fast kernel on corner-packed tables, compact = fast kernel on compact tables,
    generic = the generic kernel.  Every path must meet the same parity bar."""
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", request.param)
    return request.param


@pytest.mark.parametrize("case", fx.MODEL_CASES)
def test_model_case_vs_reference_golden(case, kernel_path):
    g = fx.load(case)
    meta = g["meta"]
    ic = fx.make_ic(meta)
    mod = fx.make_model(meta, ic)
    N = meta["n_stars"]
    pars = g["pars"]
    prim = [pars[:, 0]] + [pars[:, N + j] for j in range(4)]

    vals = ic.interp_value(prim, meta["interp_value_cols"])
    fx.assert_close(vals, g["interp_value"], RTOL, atol=ATOL, what="interp_value")
    T, lg, fe, mags = ic.interp_mag(prim, meta["bands"])
    ok = g["mag_defined"]
    fx.assert_close(T[ok], g["Teff"][ok], RTOL, what="Teff")
    fx.assert_close(lg[ok], g["logg"][ok], RTOL, what="logg")
    fx.assert_close(fe[ok], g["feh"][ok], RTOL, atol=ATOL, what="feh")
    fx.assert_close(mags[ok], g["mags"][ok], RTOL, atol=ATOL, what="mags")

    fx.assert_close(mod.lnprior(pars), g["lnprior"], RTOL, atol=ATOL, what="lnprior")
    fx.assert_close(mod.lnpost(pars), g["lnpost"], RTOL, atol=ATOL, what="lnpost")
    d = ~g["lnlike_undefined"]
    fx.assert_close(mod.lnlike(pars)[d], g["lnlike"][d], RTOL, atol=ATOL, what="lnlike")

    # scalar call form returns python floats, as the reference does for a sampler callback
    for i in (0, len(pars) // 2, len(pars) - 30):
        v = mod.lnpost(pars[i])
        assert isinstance(v, float)
        fx.assert_close([v], [g["lnpost"][i]], RTOL, atol=ATOL, what="scalar lnpost")

    # device-resident forms: row-major [N, n_par] and SoA [n_par, N]
    import torch
    dev_pars = torch.as_tensor(pars, device="cuda")
    a = mod.lnpost(dev_pars)
    b = mod.lnpost(dev_pars.T.contiguous(), soa=True)
    assert a.is_cuda and torch.equal(torch.nan_to_num(a, nan=1.5), torch.nan_to_num(b, nan=1.5))
    fx.assert_close(a.cpu().numpy(), g["lnpost"], RTOL, atol=ATOL, what="lnpost (device)")

    # mnest_prior: host scalar form and device batch form
    row = g["cube_in"][0].copy()
    mod.mnest_prior(row, None, None)
    fx.assert_close(row, g["cube_out"][0], 1e-15, what="mnest_prior scalar")
    cube = torch.as_tensor(g["cube_in"].copy(), device="cuda")
    mod.mnest_prior(cube)
    fx.assert_close(cube.cpu().numpy(), g["cube_out"], 1e-15, what="mnest_prior device")


def _random_model(kind, n_stars, bands, rng):
    obs_bands = tuple(bands)
    bands = obs_bands or ("G",)        # the BC table needs >= 1 column even if no band is observed
    if kind == "track":
        fehs = np.array([-2.0, -1.0, -0.5, -0.25, 0.0, 0.25, 0.5])
        masses = ia.grids.mist_masses()[20:150:3]
        eeps = np.arange(200.0, 900.0)
        ic = ia.synthetic_track(bands=bands, fehs=fehs, masses=masses, eeps=eeps, eep_bounds=(200, 899),
                                limits=dict(mass=(masses[0], masses[-1]), feh=(-2.0, 0.5), age=(5, 10.13)))
        lo = np.array([masses[0], 200, -2.0, 5.0, 0.0])
        hi = np.array([masses[-1], 899, 0.5, 2000.0, 1.0])
    else:
        ages = ia.grids.mist_log_ages()[40::3]
        fehs = np.array([-2.0, -1.0, -0.5, -0.25, 0.0, 0.25, 0.5])
        eeps = np.arange(150.0, 900.0)
        ic = ia.synthetic_isochrone(bands=bands, ages=ages, fehs=fehs, eeps=eeps, eep_bounds=(150, 899),
                                    limits=dict(age=(ages[0], ages[-1]), feh=(-2.0, 0.5)))
        lo = np.array([150.0] * n_stars + [ages[0], -2.0, 5.0, 0.0])
        hi = np.array([899.0] * n_stars + [ages[-1], 0.5, 2000.0, 1.0])
    obs = dict(Teff=(5770, 100), logg=(4.4, 0.1), feh=(0.0, 0.15), parallax=(2.0, 0.05))
    for j, b in enumerate(obs_bands):
        obs[b] = (10.0 + 0.3 * j, 0.02)
    mod = ia.BasicStarModel(ic, N=n_stars, **obs)
    return ic, mod, lo, hi


@pytest.mark.parametrize("kind,n_stars,nb", [("track", 1, 1), ("track", 1, 3), ("iso", 1, 1), ("iso", 2, 6),
                                             ("iso", 3, 2), ("iso", 2, 11), ("iso", 1, 0)])
def test_random_batch_vs_oracle(kind, n_stars, nb, kernel_path):
    """Mid-size tables, 2e5 seeded samples, every kernel specialisation (nb 0..8 compile-time,
    >8 runtime loop) against the CPU oracle."""
    rng = np.random.default_rng(1000 + 10 * n_stars + nb)
    bands = ia.grids.DEFAULT_BANDS[:nb]
    ic, mod, lo, hi = _random_model(kind, n_stars, bands, rng)
    n = 200_000
    span = hi - lo
    pars = rng.uniform(lo - 0.02 * span, hi + 0.02 * span, size=(n, lo.size))
    if n_stars > 1:
        pars[: n // 2, :n_stars] = -np.sort(-pars[: n // 2, :n_stars], axis=1)
    oic = fx.make_oracle_ic(ic)
    w_post, w_prior, w_like = oic.lnpost(mod.model_desc(), pars.T.copy(), nthreads=8)
    assert np.isfinite(w_post).sum() > n // 50
    fx.assert_close(mod.lnpost(pars), w_post, RTOL, atol=ATOL, what="lnpost")
    fx.assert_close(mod.lnprior(pars), w_prior, RTOL, atol=ATOL, what="lnprior")
    fx.assert_close(mod.lnlike(pars), w_like, RTOL, atol=ATOL, what="lnlike")
    if nb:
        prim = np.column_stack([pars[:, 0]] + [pars[:, n_stars + j] for j in range(4)]).T.copy()
        prim[3] = np.abs(prim[3]) + 1.0
        wT, wg, wf, wm = oic.interp_mag(prim, [ic.bc_grid.interp.column_index[b] for b in bands], nthreads=8)
        T, g_, f, m = ic.interp_mag(list(prim), list(bands))
        fx.assert_close(T, wT, RTOL, what="Teff")
        fx.assert_close(m, wm, RTOL, atol=ATOL, what="mags")


@pytest.mark.parametrize("k", [1, 2, 3, 7, 18, 40])
def test_interp_wide_pack_matches_column_parallel_kernel(k, monkeypatch):
    """3-D DFInterpolator batches >= 32768 rows build the [cell][column][corner] pack and use the
    quad-per-(sample, column) kernel; it must agree with the column-parallel kernel (which the oracle
    tests pin) for any column subset, incl. NaN cells, NaN / out-of-range / exact-node / upper-edge inputs."""
    import torch
    rng = np.random.default_rng(100 + k)
    ax = [np.sort(rng.uniform(-2, 2, 11)), np.array([0.1, 0.2, 0.5, 0.9, 1.0, 1.5, 4.0]), np.arange(1.0, 201.0)]
    ncol = 40
    grid = rng.standard_normal((11, 7, 200, ncol))
    grid[3:5, 2:4, 150:, :] = np.nan                       # ragged tail
    grid[7, 5, 20, 11] = np.nan                            # a single missing value in one column
    t = DFInterpolator.from_arrays(grid, ax, ["c%d" % j for j in range(ncol)])
    n = 70_001                                             # last wave partially filled
    x = [rng.uniform(a[0] - 0.02 * (a[-1] - a[0]), a[-1] + 0.02 * (a[-1] - a[0]), n) for a in ax]
    for d in range(3):
        x[d][d * 100:d * 100 + 50] = rng.choice(ax[d], 50)            # exact nodes (incl. first / last)
        x[d][1000 + d] = np.nan
    x[0][2000:2010], x[1][2000:2010], x[2][2000:2010] = ax[0][-1], ax[1][-1], ax[2][-1]   # upper corner
    cols = list(rng.choice(ncol, size=k, replace=False))
    xt = [torch.as_tensor(v, device="cuda") for v in x]
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", "generic")
    want = t.interp_device(xt, np.array(cols)).cpu().numpy()
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", "auto")
    got = t.interp_device(xt, np.array(cols)).cpu().numpy()
    assert got.shape == (n, k) and np.isfinite(want).mean() > 0.5
    fx.assert_close(got, want, 1e-12, atol=1e-12, what="wide pack k=%d" % k)
    small = t.interp_device([v[:1500].contiguous() for v in xt], np.array(cols)).cpu().numpy()
    assert np.array_equal(np.nan_to_num(small, nan=7.0), np.nan_to_num(got[:1500], nan=7.0))


@pytest.mark.parametrize("k", [1, 2, 3, 5])
@pytest.mark.parametrize("groups", [2, 3, 4, 7])
def test_interp_wide_pack_groups_per_wave(k, groups, monkeypatch):
    """Large 1-3 column batches let one wave serve several groups of 64 samples (next group's coordinates prefetched; one column
    through the four-pass instantiation).  Forced here on a batch whose size leaves the last wave with fewer groups than the others
    and its last group partially filled: bit for bit the one-group form, which the test above pins; bad samples at group borders."""
    import torch
    rng = np.random.default_rng(300 + 10 * k + groups)
    ax = [np.sort(rng.uniform(-2, 2, 9)), np.array([0.1, 0.2, 0.5, 0.9, 1.0, 1.5, 4.0]), np.arange(1.0, 151.0)]
    ncol = 12
    grid = rng.standard_normal((9, 7, 150, ncol))
    grid[2:4, 2:4, 100:, :] = np.nan
    t = DFInterpolator.from_arrays(grid, ax, ["c%d" % j for j in range(ncol)])
    n = 64 * groups * 4 * 37 + 64 * (groups - 1) + 17
    n = max(n, 40_000 + 17)
    x = [rng.uniform(a[0] - 0.02 * (a[-1] - a[0]), a[-1] + 0.02 * (a[-1] - a[0]), n) for a in ax]
    for i in (63, 64, 65, 64 * groups - 1, 64 * groups, n - 18, n - 1):
        x[i % 3][i] = np.nan
    cols = np.array(list(rng.choice(ncol, size=k, replace=False)))
    xt = [torch.as_tensor(v, device="cuda") for v in x]
    monkeypatch.setenv("ISOCHRONES_AMD_WIDE_GROUPS", "1")
    monkeypatch.setenv("ISOCHRONES_AMD_WIDE_NARROW", "0")
    want = t.interp_device(xt, cols).cpu().numpy()
    assert np.isfinite(want).mean() > 0.5
    for narrow in ("0", "1"):
        monkeypatch.setenv("ISOCHRONES_AMD_WIDE_GROUPS", str(groups))
        monkeypatch.setenv("ISOCHRONES_AMD_WIDE_NARROW", narrow)
        got = t.interp_device(xt, cols).cpu().numpy()
        assert np.array_equal(np.nan_to_num(got, nan=7.0), np.nan_to_num(want, nan=7.0)), (k, groups, narrow)
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", "generic")
    ref = t.interp_device(xt, cols).cpu().numpy()
    fx.assert_close(want, ref, 1e-12, atol=1e-12, what="wide pack, one group, k=%d" % k)


def test_interp_mag_packed_path_matches_generic_kernel(monkeypatch):
    """iso_interp_mag switches to the corner-packed tables for large batches (a pack per band list, built on
    first use, at most 6 kept): both kernels against each other on the same samples incl. NaN / out-of-range
    / exact-node inputs, optional outputs, small batches after a pack exists, and pack eviction."""
    import torch
    rng = np.random.default_rng(77)
    bands = ia.grids.DEFAULT_BANDS[:8]
    ic, mod, lo, hi = _random_model("track", 1, bands, rng)
    n = 150_000
    span = hi - lo
    pars = rng.uniform(lo - 0.02 * span, hi + 0.02 * span, size=(n, 5))
    pars[:50, 0] = ic.model_grid.masses[rng.integers(0, len(ic.model_grid.masses), 50)]      # exact nodes
    pars[50:60, rng.integers(0, 5, 10)] = np.nan
    pars[60:70, 3] = [0.0, -1.0, np.inf, 1e300, 1e-300, 10.0, 1.0, 5.0, 2.0, 3.0]
    pars[70:80, 4] = [0.0, 1.0, -0.0, 1.0000001, np.inf, -np.inf, 0.5, 0.25, 0.75, 0.1]
    pt = torch.as_tensor(np.ascontiguousarray(pars.T), device="cuda")
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", "auto")
    ic.interp_mag_device(pt[:, :4].contiguous(), list(bands[:1]))      # interpolator handle built with packed tables
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", "generic")
    ref = {}
    for nb in (1, 2, 5, 8):
        ref[nb] = [t.clone() for t in ic.interp_mag_device(pt, list(bands[:nb]))]
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", "auto")
    for nb in (1, 2, 5, 8):
        got = ic.interp_mag_device(pt, list(bands[:nb]))
        for a, b, what in zip(got, ref[nb], ("Teff", "logg", "feh", "mags")):
            fx.assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-12, atol=1e-12, what="%s nb=%d" % (what, nb))
        small = ic.interp_mag_device(pt[:, :2000].contiguous(), list(bands[:nb]))     # pack exists -> packed kernel
        assert torch.equal(torch.nan_to_num(small[3], nan=7.0), torch.nan_to_num(got[3][:2000], nan=7.0))
    # more band lists than pack slots: earlier packs are dropped and rebuilt transparently
    for k in range(8):
        sel = [bands[k], bands[(k + 3) % 8]]
        got = ic.interp_mag_device(pt, sel)
        monkeypatch.setenv("ISOCHRONES_AMD_PATH", "generic")
        want = ic.interp_mag_device(pt, sel)
        monkeypatch.setenv("ISOCHRONES_AMD_PATH", "auto")
        fx.assert_close(got[3].cpu().numpy(), want[3].cpu().numpy(), 1e-12, atol=1e-12, what="mags %s" % sel)
    got = ic.interp_mag_device(pt, list(bands[:1]))
    fx.assert_close(got[3].cpu().numpy(), ref[1][3].cpu().numpy(), 1e-12, atol=1e-12, what="mags after eviction")


@pytest.mark.parametrize("kind,n_stars,nb,dnu", [("track", 1, 2, True), ("iso", 1, 1, False), ("iso", 2, 5, True)])
def test_asteroseismic_terms_fast_kernel_vs_oracle(kind, n_stars, nb, dnu, kernel_path):
    """nu_max / delta_nu terms (starmodel.py:1603-1612; delta_nu uses its VALUE as the uncertainty) on every
    kernel path — on `auto` through the ASTERO instantiation of the fused kernel with its own
    corner-packed (nu_max, delta_nu) table — against the oracle, incl. the parts and the scalar form."""
    rng = np.random.default_rng(4000 + n_stars + nb)
    ic, mod0, lo, hi = _random_model(kind, n_stars, ia.grids.DEFAULT_BANDS[:nb], rng)
    kw = dict(mod0.kwargs)
    kw["nu_max"] = (2000.0, 150.0)
    if dnu:
        kw["delta_nu"] = (100.0, 3.0)
    mod = ia.BasicStarModel(ic, N=n_stars, **kw)
    d = mod.model_desc()
    assert d.has_numax == 1 and d.has_dnu == int(dnu)
    n = 120_000
    span = hi - lo
    pars = rng.uniform(lo - 0.02 * span, hi + 0.02 * span, size=(n, lo.size))
    pars[:40, 0] = np.nan
    if n_stars > 1:
        pars[: n // 2, :n_stars] = -np.sort(-pars[: n // 2, :n_stars], axis=1)
    oic = fx.make_oracle_ic(ic)
    w_post, w_prior, w_like = oic.lnpost(d, pars.T.copy(), nthreads=8)
    assert np.isfinite(w_post).sum() > n // 50
    fx.assert_close(mod.lnpost(pars), w_post, RTOL, atol=ATOL, what="lnpost")
    fx.assert_close(mod.lnlike(pars), w_like, RTOL, atol=ATOL, what="lnlike")
    fx.assert_close(mod.lnprior(pars), w_prior, RTOL, atol=ATOL, what="lnprior")
    k = int(np.flatnonzero(np.isfinite(w_post))[0])
    assert np.isclose(mod.lnpost(pars[k]), w_post[k], rtol=RTOL)
    # the terms really are in: the same model without them differs
    assert not np.allclose(mod0.lnpost(pars[k:k + 1]), w_post[k:k + 1])
    # sampling an asteroseismic model works (framework-op sampler around the same kernel)
    if kind == "track" and kernel_path == "auto":
        good = pars[np.isfinite(w_post)][:64]
        s = mod.fit_mcmc(nwalkers=64, nburn=5, niter=5, p0=None if len(good) < 64 else good[0], seed=1)
        assert s.chain.shape[1] == 5


def test_size_independent_properties_large_batch():
    """10^6-sample batch: permutation equivariance, batch-split invariance, and exactness of
    interpolation on a table whose columns are affine in the coordinates."""
    import torch
    rng = np.random.default_rng(7)
    ic, mod, lo, hi = _random_model("track", 1, ("G",), rng)
    n = 1_000_000
    pars = torch.as_tensor(rng.uniform(lo, hi, size=(n, 5)), device="cuda")
    full = mod.lnpost(pars)
    perm = torch.randperm(n, device="cuda")
    assert torch.equal(torch.nan_to_num(mod.lnpost(pars[perm]), nan=7.0), torch.nan_to_num(full[perm], nan=7.0))
    parts = torch.cat([mod.lnpost(pars[:333_333]), mod.lnpost(pars[333_333:])])
    assert torch.equal(torch.nan_to_num(parts, nan=7.0), torch.nan_to_num(full, nan=7.0))
    # affine table: multilinear interpolation must reproduce it to rounding
    ax = [np.linspace(0, 1, 9), np.sort(rng.uniform(0, 5, 40)), np.arange(1.0, 301.0)]
    A, B, Cc = np.meshgrid(*ax, indexing="ij")
    grid = np.stack([2 * A - 3 * B + 0.5 * Cc + 1, A + B + Cc], axis=-1)
    t = DFInterpolator.from_arrays(grid, ax, ["u", "v"])
    x = [torch.as_tensor(rng.uniform(a[0], a[-1], n), device="cuda") for a in ax]
    out = t(x, ["u", "v"])
    want_u = 2 * x[0] - 3 * x[1] + 0.5 * x[2] + 1
    assert torch.allclose(out[:, 0], want_u, rtol=0, atol=1e-10)
    assert torch.allclose(out[:, 1], x[0] + x[1] + x[2], rtol=0, atol=1e-10)


def test_errors_are_loud():
    from isochrones_amd import _cabi
    k = fx.load("interp_kats")
    t3 = DFInterpolator.from_arrays(k["t3_grid"], [k["t3_axes0"], k["t3_axes1"], k["t3_axes2"]], ["val"])
    with pytest.raises(KeyError):
        t3([1.0, 1.0, 1.0], ["nope"])
    with pytest.raises(ValueError):
        DFInterpolator.from_arrays(k["t3_grid"], [k["t3_axes0"][::-1], k["t3_axes1"], k["t3_axes2"]], ["val"])
    ic = ia.synthetic_track(bands=("V",), fehs=[-1, 0], masses=[0.8, 1.0, 1.2], eeps=np.arange(300., 340.))
    with pytest.raises(ValueError):
        ia.BinaryStarModel(ic, V=(10, 0.1))      # multiples need the isochrone parametrisation
    rc = _cabi.lib().iso_lnpost(None, None, 1, 1, 0, None, None, None, None)
    assert rc == -1 and b"NULL" in _cabi.lib().iso_last_error()
    # shape / dtype misuse is refused on the host before any kernel could read out of bounds
    import torch
    mod = ia.SingleStarModel(ic, V=(10, 0.1), Teff=(5700, 100))
    for bad in (np.zeros((10, 4)), np.zeros((10, 6)), [1.0, 320.0, 0.0, 100.0], np.zeros((2, 3, 5))):
        with pytest.raises(ValueError):
            mod.lnpost(bad)
    assert mod.lnpost(np.zeros((0, 5))).shape == (0,)
    with pytest.raises(ValueError):
        ic.interp_mag([1.0, 320.0, 0.0, 100.0], ["V"])                       # four parameters
    with pytest.raises(ValueError):
        ic.interp_mag_device(torch.zeros(4, 8, device="cuda", dtype=torch.float64), ["V"])
    with pytest.raises(ValueError):
        ic.interp_mag_device(torch.zeros(5, 8, device="cuda", dtype=torch.float32), ["V"])
    with pytest.raises(ValueError):
        ic.interp_mag([1.0, 320.0, 0.0, 100.0, 0.1], ["Z"])                  # unknown band
    with pytest.raises(ValueError):
        t3.interp_device([torch.zeros(4, device="cuda", dtype=torch.float64)] * 2, np.array([0]))
    with pytest.raises(ValueError):
        t3.interp_device([torch.zeros(4, device="cuda", dtype=torch.float64), torch.zeros(3, device="cuda", dtype=torch.float64),
                          torch.zeros(4, device="cuda", dtype=torch.float64)], np.array([0]))


def test_get_eep_and_generate():
    """'next' row f2: get_eep (interp_eeps) vs the reference golden and the oracle; generate()."""
    import torch
    from oracle import oracle as orc
    g = fx.load("interp_eep")
    fehs, masses = g["fehs"], g["masses"]
    n_eep = g["ages"].shape[1]
    # rebuild a track interpolator whose `age` column is the fixture's ragged array
    grid, ax, cols = ia.grids.synthetic_track_grid(fehs, masses, np.arange(1.0, n_eep + 1.0))
    grid[..., cols.index("age")] = g["ages"].reshape(fehs.size, masses.size, n_eep)
    grid[np.isnan(grid[..., cols.index("age")])] = np.nan
    from isochrones_amd.models import EvolutionTrackGrid, EvolutionTrackInterpolator, BolometricCorrectionGrid
    bcg, bax, bands = ia.grids.synthetic_bc_grid(("G",))
    ic = EvolutionTrackInterpolator(EvolutionTrackGrid(DFInterpolator.from_arrays(grid, ax, cols)),
                                    BolometricCorrectionGrid(DFInterpolator.from_arrays(bcg, bax, bands), bands=bands),
                                    bands=bands)
    got = ic.get_eep(g["mass"], g["age"], g["feh"])
    fx.assert_close(got, g["eep"], 1e-12, what="get_eep vs reference")
    dev_out = ic.get_eep(torch.as_tensor(g["mass"], device="cuda"), torch.as_tensor(g["age"], device="cuda"),
                         torch.as_tensor(g["feh"], device="cuda"))
    fx.assert_close(dev_out.cpu().numpy(), g["eep"], 1e-12, what="get_eep device")
    k = int(np.flatnonzero(np.isfinite(g["eep"]))[0])
    v = ic.get_eep(float(g["mass"][k]), float(g["age"][k]), float(g["feh"][k]))
    assert isinstance(v, float) and np.isclose(v, g["eep"][k], rtol=1e-12)
    # a larger random batch against the oracle
    rng = np.random.default_rng(5)
    n = 100_000
    m, a, f = rng.uniform(0.45, 7.3, n), rng.uniform(4.8, 11.0, n), rng.uniform(-1.1, 0.55, n)
    want = orc.interp_eep(a, f, m, fehs, masses, ic._age_grid, ic._array_lengths)
    fx.assert_close(ic.get_eep(m, a, f), want, 1e-12, what="get_eep vs oracle")
    # generate(): columns consistent with interp_value / interp_mag at the interpolated EEP
    ok = np.flatnonzero(np.isfinite(want))[:50]
    df = ic.generate(m[ok], a[ok], f[ok], distance=100.0, AV=0.1)
    assert {"Teff", "logg", "age", "G_mag", "distance", "requested_age"} <= set(df.columns) and len(df) == 50
    T = ic.interp_value([m[ok], want[ok], f[ok]], ["Teff"])[:, 0]
    fx.assert_close(df["Teff"].values, T, 1e-12, what="generate Teff")
    with pytest.raises(ValueError):           # an isochrone grid asks its companion track grid (reference: .track)
        ia.synthetic_isochrone(bands=("G",), ages=[9.0, 9.5], fehs=[-0.5, 0.0], eeps=np.arange(1., 9.)).get_eep(1.0, 9.2, 0.0)
    # accurate=True: Nelder-Mead on (age - age(eep))^2 from the fast estimate (reference models.py:544-578)
    for j in ok[:3]:
        e = ic.get_eep(float(m[j]), float(a[j]), float(f[j]), accurate=True)
        assert abs(ic.interp_value([float(m[j]), e, float(f[j])], ["age"])[0] - a[j]) < 0.02
        assert 1.0 <= e <= ic.max_eep(float(m[j]), float(f[j]))
    assert np.isclose(ic.mass_age_resid(want[ok[0]], m[ok[0]], a[ok[0]], f[ok[0]]),
                      (a[ok[0]] - ic.interp_value([m[ok[0]], want[ok[0]], f[ok[0]]], ["age"])[0]) ** 2)
    assert ic.masses is ic.model_grid.masses
    with pytest.raises(AttributeError):
        ic.ages


def test_get_eep_exact_age_hits_and_repeated_ages():
    """Reference fixture with runs of equal ages inside tracks and 3600 queries that hit table ages exactly: which of
    the equal elements the reference's bisection returns (interp.py:26-29) decides the EEP, so the kernel walks the
    same probe sequence.  Host-array, device-tensor and scalar forms."""
    import torch
    from isochrones_amd.models import EvolutionTrackGrid, EvolutionTrackInterpolator, BolometricCorrectionGrid
    g = fx.load("interp_eep_plateaus")
    fehs, masses = g["fehs"], g["masses"]
    n_eep = g["ages"].shape[1]
    grid, ax, cols = ia.grids.synthetic_track_grid(fehs, masses, np.arange(1.0, n_eep + 1.0))
    grid[..., cols.index("age")] = g["ages"].reshape(fehs.size, masses.size, n_eep)
    bcg, bax, bands = ia.grids.synthetic_bc_grid(("G",))
    ic = EvolutionTrackInterpolator(EvolutionTrackGrid(DFInterpolator.from_arrays(grid, ax, cols)),
                                    BolometricCorrectionGrid(DFInterpolator.from_arrays(bcg, bax, bands), bands=bands),
                                    bands=bands)
    want = g["eep"]
    fx.assert_close(ic.get_eep(g["mass"], g["age"], g["feh"]), want, 1e-13, what="get_eep, repeated ages")
    dev_out = ic.get_eep(*(torch.as_tensor(g[k], device="cuda") for k in ("mass", "age", "feh")))
    fx.assert_close(dev_out.cpu().numpy(), want, 1e-13, what="get_eep device, repeated ages")
    for k in range(3000, 3040):
        v = ic.get_eep(float(g["mass"][k]), float(g["age"][k]), float(g["feh"][k]))
        assert (np.isnan(v) and np.isnan(want[k])) or np.isclose(v, want[k], rtol=1e-13)
    # the fixture does exercise the difference: "first index with age >= x" is not the reference's answer everywhere
    from oracle import oracle as orc
    assert np.array_equal(orc.interp_eep(g["age"], g["feh"], g["mass"], fehs, masses, g["ages"], g["lengths"]), want,
                          equal_nan=True)


def test_ingest_derived_columns():
    """'next' row f1: dt_deep / dm_deep as np.gradient over the populated points of each track."""
    from isochrones_amd import ingest
    g, ax, cols = ia.grids.synthetic_track_grid([-0.5, 0.0], [0.5, 1.0, 2.0], np.arange(1.0, 500.0))
    keep = [c for c in cols if c != "dt_deep"]
    dfi = DFInterpolator.from_arrays(g[..., [cols.index(c) for c in keep]], ax, keep)
    ingest.add_dt_deep(dfi)
    assert dfi.columns[-1] == "dt_deep"
    d = dfi.grid[..., -1]
    for i in range(2):
        for j in range(3):
            sa = dfi.grid[i, j, :, dfi.column_index["star_age"]]
            okk = ~np.isnan(sa)
            want = np.gradient(np.log10(sa[okk]), ax[2][okk])
            assert np.allclose(d[i, j, okk], want, rtol=1e-13) and np.isnan(d[i, j, ~okk]).all()
    analytic = g[..., cols.index("dt_deep")]
    inner = ~np.isnan(analytic)
    assert np.nanmax(np.abs(d[inner] - analytic[inner])) < 1e-3
    # the table with the appended column serves queries after the (lazy) re-upload
    v = dfi([0.0, 1.0, 250.0], ["dt_deep"])
    assert np.isclose(v[0], d[1, 1, 249], rtol=1e-12)


def test_batch_size_edges():
    """Empty, single-sample, ragged (non multiple of the 256-thread workgroup) and very large
    batches through every entry point."""
    import torch
    rng = np.random.default_rng(21)
    ic, mod, lo, hi = _random_model("track", 1, ("G", "BP"), rng)
    oic = fx.make_oracle_ic(ic)
    desc = mod.model_desc()
    for n in (0, 1, 63, 255, 256, 257, 1000):
        pars = rng.uniform(lo, hi, size=(n, 5))
        got = mod.lnpost(pars)
        assert got.shape == (n,)
        if n:
            fx.assert_close(got, oic.lnpost(desc, pars.T.copy(), parts=False), RTOL, atol=ATOL, what="n=%d" % n)
            fx.assert_close(mod.lnlike(pars), oic.lnpost(desc, pars.T.copy())[2], RTOL, atol=ATOL, what="lnlike n=%d" % n)
        v = ic.interp_value([pars[:, 0], pars[:, 1], pars[:, 2]], ["Teff", "Mbol"])
        assert v.shape == (n, 2)
        T, g, f, m = ic.interp_mag([pars[:, j] for j in range(5)], ["G", "BP"])
        if n == 1:      # five length-1 arrays squeeze to one parameter vector: scalar form, as the reference
            assert np.ndim(T) == 0 and m.shape == (2,)
        else:
            assert T.shape == (n,) and m.shape == (n, 2)
    # 10^7 samples: output fully written, checksum-of-chunks equals chunked evaluation
    n = 10_000_000
    big = torch.rand(n, 5, dtype=torch.float64, device="cuda") * torch.as_tensor(hi - lo, device="cuda") + torch.as_tensor(lo, device="cuda")
    out = mod.lnpost(big)
    assert out.shape == (n,) and not bool(torch.isnan(out[torch.isfinite(out)]).any())
    k = 3_333_333
    parts = torch.cat([mod.lnpost(big[:k]), mod.lnpost(big[k:2 * k]), mod.lnpost(big[2 * k:])])
    assert torch.equal(torch.nan_to_num(parts, nan=3.0, neginf=-1e300), torch.nan_to_num(out, nan=3.0, neginf=-1e300))
    fin = torch.isfinite(out)
    assert 0.05 < float(fin.double().mean()) < 1.0
    sub = big[::1000].cpu().numpy()
    fx.assert_close(out[::1000].cpu().numpy(), oic.lnpost(desc, sub.T.copy(), parts=False), RTOL, atol=ATOL, what="1e7 subsample")


def test_handles_survive_any_destroy_order():
    import gc
    rng = np.random.default_rng(2)
    for order in range(3):
        ic, mod, lo, hi = _random_model("iso", 2, ("G",), rng)
        p = rng.uniform(lo, hi, size=(100, lo.size))
        mod.lnpost(p)
        if order == 0:
            del ic; gc.collect(); del mod
        elif order == 1:
            del mod; gc.collect(); del ic
        else:
            ic.model_grid.interp.release(); ic.release(); mod._dirty()
            mod.lnpost(p)            # everything is rebuilt lazily
        gc.collect()


def test_exact_upper_edge_is_defined(kernel_path):
    """The reference reads past the table for a query exactly on an axis' last node (undefined);
    this build defines it as the value at that node (i = n-2, t = 1) on every path."""
    rng = np.random.default_rng(3)
    ic, mod, lo, hi = _random_model("track", 1, ("G",), rng)
    f, m, e = ic.model_grid.interp.index_columns
    oic = fx.make_oracle_ic(ic)
    pars = np.array([[m[-1], 450.0, 0.0, 300.0, 0.2], [1.0, e[-1], 0.0, 300.0, 0.2], [1.0, 450.0, f[-1], 300.0, 0.2],
                     [m[-1], e[-1], f[-1], 300.0, 0.2], [m[0], e[0], f[0], 300.0, 0.2]])
    want = oic.lnpost(mod.model_desc(), pars.T.copy())
    got = mod.lnpost(pars)
    fx.assert_close(got, want[0], 1e-9, atol=1e-9, what="upper edge lnpost")
    v = ic.interp_value([pars[:, 0], pars[:, 1], pars[:, 2]], ["Teff", "Mbol"])
    wv = oic.model.interp([pars[:, 2], pars[:, 0], pars[:, 1]], [ic.model_grid.interp.column_index[c] for c in ("Teff", "Mbol")])
    fx.assert_close(v, wv, 1e-12, what="upper edge interp_value")
    node = ic.model_grid.interp.grid[-1, 20, 100, ic.model_grid.interp.column_index["Teff"]]
    assert np.isclose(ic.interp_value([m[20], e[100], f[-1]], ["Teff"])[0], node, rtol=1e-13)


def test_streams_and_graph_capture():
    """Calls are asynchronous on the caller's (torch current) stream and can be captured into a
    HIP graph; results on a side stream / from a replay equal the default-stream results."""
    import torch
    rng = np.random.default_rng(4)
    ic, mod, lo, hi = _random_model("track", 1, ("G",), rng)
    x = torch.as_tensor(rng.uniform(lo, hi, size=(4096, 5)), device="cuda")
    ref = mod.lnpost(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y = x * 1.0                       # produced on the side stream, consumed by the kernel on it
        out = mod.lnpost(y)
    side.synchronize()
    clean = lambda t: torch.nan_to_num(t, nan=5.0, neginf=-1e300)
    assert torch.equal(clean(out), clean(ref))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        mod.lnpost(x)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        captured = mod.lnpost(x)
    x.copy_(torch.as_tensor(rng.uniform(lo, hi, size=(4096, 5)), device="cuda"))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(clean(captured), clean(mod.lnpost(x)))


def test_isochrone_and_generate_binary():
    ages = ia.grids.mist_log_ages()[70::4]
    ic = ia.synthetic_isochrone(bands=("G", "K"), ages=ages, fehs=[-0.5, 0.0, 0.5], eeps=np.arange(200.0, 400.0),
                                eep_bounds=(200, 399), limits=dict(age=(ages[0], ages[-1]), feh=(-0.5, 0.5), eep=(200, 400)))
    iso = ic.isochrone(float(ages[2]), feh=0.1, distance=100.0, AV=0.1)
    assert len(iso) > 100 and {"Teff", "mass", "G_mag", "K_mag"} <= set(iso.columns)
    assert np.all(np.diff(iso["eep"].values) > 0)
    trk = ia.synthetic_track(bands=("G", "K"), fehs=[-0.5, 0.0, 0.5], masses=[0.7, 0.9, 1.0, 1.1, 1.3], eeps=np.arange(1.0, 500.0))
    df = trk.generate_binary([1.0, 1.05], [0.8, 0.9], 8.0, 0.0, distance=50.0)
    tot = -2.5 * np.log10(10 ** (-0.4 * df["G_mag_0"]) + 10 ** (-0.4 * df["G_mag_1"]))
    assert np.allclose(df["G_mag"], tot) and np.all(df["G_mag"] < df["G_mag_0"])


@pytest.mark.parametrize("kind,n_stars", [("track", 1), ("iso", 2), ("iso", 3)])
def test_special_value_fuzz_vs_oracle(kind, n_stars, kernel_path):
    """Every parameter slot takes NaN, +-inf, 0, -0, huge, tiny, negative and exact-node values (in
    random combinations); the kernels must reproduce the oracle's value / NaN / -inf for each."""
    rng = np.random.default_rng(77 + n_stars)
    ic, mod, lo, hi = _random_model(kind, n_stars, ("G", "BP"), rng)
    n, D = 40_000, lo.size
    pars = rng.uniform(lo, hi, size=(n, D))
    if n_stars > 1:
        pars[:, :n_stars] = -np.sort(-pars[:, :n_stars], axis=1)
    specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e300, -1e300, 5e-324, 1e-300, -1.0, 1.0])
    hit = rng.random((n, D)) < 0.08
    pars[hit] = rng.choice(specials, size=int(hit.sum()))
    ax = ic.model_grid.interp.index_columns
    node_rows = rng.choice(n, 2000, replace=False)               # exact table nodes
    if kind == "track":
        pars[node_rows, 0] = rng.choice(ax[1], 2000)
        pars[node_rows, 1] = rng.choice(ax[2], 2000)
        pars[node_rows, 2] = rng.choice(ax[0], 2000)
    else:
        pars[node_rows, 0] = rng.choice(ax[2], 2000)
        pars[node_rows, n_stars] = rng.choice(ax[0], 2000)
        pars[node_rows, n_stars + 1] = rng.choice(ax[1], 2000)
    oic = fx.make_oracle_ic(ic)
    with np.errstate(all="ignore"):
        want = oic.lnpost(mod.model_desc(), pars.T.copy(), nthreads=8)
    fx.assert_close(mod.lnpost(pars), want[0], RTOL, atol=ATOL, what="lnpost")
    fx.assert_close(mod.lnprior(pars), want[1], RTOL, atol=ATOL, what="lnprior")
    d_ok = pars[:, n_stars + 2] > 0          # lnlike with distance <= 0 is undefined in the reference
    fx.assert_close(mod.lnlike(pars)[d_ok], want[2][d_ok], RTOL, atol=ATOL, what="lnlike")


def test_randomised_soak_short(monkeypatch):
    """tests/soak/soak.py for a few seconds: random model configurations (bands, stars, observables, prior
    families, bounds) x special-value-laden samples on all three kernel paths against the oracle
    (the long runs are recorded in profiles/r01/soak.txt)."""
    import runpy
    monkeypatch.setattr(sys, "argv", ["soak.py", "6", "11"])
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", "auto")          # the soak switches kernel paths through this variable:
                                                               # monkeypatch puts it back for the tests that follow
    with pytest.raises(SystemExit) as e:
        runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "soak", "soak.py"),
                       run_name="__main__")
    assert e.value.code == 0


def test_docs_grid_interpolator_notebook_flow():
    """docs/grid_interpolator.ipynb, call for call, on the MIST-shaped synthetic tables (values differ from
    the notebook's, the calls, shapes and cross-consistency do not)."""
    import pandas as pd
    from isochrones_amd.mist import MIST_EvolutionTrack, MIST_Isochrone
    mist = MIST_Isochrone()
    T0, g0, f0, m0 = mist.initialize()                          # tables on the device + one evaluation
    assert np.isfinite(T0) and m0.shape == (len(mist.bands),) and mist.name == "IsochroneInterpolator"
    pars = [353, 9.78, -1.24]                                    # eep, log(age), feh
    v = mist.interp_value(pars, ["mass", "radius", "Teff"])
    assert v.shape == (3,) and np.all(np.isfinite(v))
    T, g, f, mags = mist.interp_mag(pars + [200, 0.11], ["K", "BP", "RP"])
    assert np.isfinite([T, g, f]).all() and mags.shape == (3,)
    mist_track = MIST_EvolutionTrack()
    tp = [float(v[0]), 353, -1.24]                               # mass, eep, feh [matching above]
    w = mist_track.interp_value(tp, ["mass", "radius", "Teff", "age"])
    assert w.shape == (4,) and np.isfinite(w).all()
    assert np.isclose(mist_track.mass(*tp), w[0]) and isinstance(float(mist_track.mass(*tp)), float)
    iso = mist.isochrone(9.53, 0.1)
    assert isinstance(iso, pd.DataFrame) and len(iso) > 100 and {"Teff", "logg", "K_mag", "eep"} <= set(iso.columns)
    assert np.allclose(iso["age"], 9.53) and not iso.isna().any().any()
    df = mist_track([0.8, 0.9, 1.0], 350, 0.0, distance=100, AV=0.1)
    assert len(df) == 3 and "G_mag" in df.columns and "distance" not in df.columns      # as the reference
    far = mist_track([0.8, 0.9, 1.0], 350, 0.0, distance=1000, AV=0.1)
    assert np.allclose(far["G_mag"] - df["G_mag"], 5.0)
    gen = mist_track.generate([0.81, 0.91, 1.01], 9.51, 0.01)
    assert len(gen) == 3 and np.allclose(gen["requested_age"], 9.51)
    e = mist_track.get_eep(1.01, 9.51, 0.01)
    ea = mist_track.get_eep(1.01, 9.51, 0.01, accurate=True)
    assert isinstance(e, float) and isinstance(ea, float) and abs(e - ea) < 3
    ages = [mist_track.interp_value([1.01, x, 0.01], ["age"])[0] for x in (e, ea)]
    assert abs(ages[1] - 9.51) <= abs(ages[0] - 9.51) + 1e-9 and abs(ages[1] - 9.51) < 1e-3
    df0 = mist_track.generate([0.81, 0.91, 1.01], 9.51, 0.01, accurate=True)
    rel = ((gen - df0) / df0).mean()
    assert np.all(np.abs(rel[["Teff", "radius", "mass"]]) < 0.01)
    d = mist_track.generate(1.0, 9.6, 0.0, return_dict=True, all_As=True, AV=0.3, bands=["G", "K"])
    assert isinstance(d, dict) and d["A_G"][0] > 0 and d["A_K"][0] > 0        # toy BC tables: no band ordering
    big = mist_track.generate(np.ones(10000) * 1.01, np.ones(10000) * 9.82, np.ones(10000) * 0.02)
    assert len(big) == 10000


def test_broken_prior_with_custom_parameters_on_the_device():
    """priors.BrokenPrior([LogNormal, PowerLaw], [bp]) (reference priors.py:143-232) with non-Chabrier numbers as
    the mass prior of a track model: the kernel's lnprior moves by exactly log(custom(m)) - log(default(m)), and
    the oracle agrees with the kernel on the whole batch."""
    from isochrones_amd import priors as P
    g = fx.load("track_single_spec_phot")
    mod = fx.make_model(g["meta"])
    ic = mod.ic
    pars = g["pars"]
    base = mod.lnprior(pars)
    custom = P.BrokenPrior([P.LogNormalPrior(np.log(0.2), 0.6), P.PowerLawPrior(-2.0, (0.8, 50.0))], [0.8], bounds=(0.05, 50.0))
    default = mod._priors["mass"]
    mod.set_prior(mass=custom)
    got = mod.lnprior(pars)
    fin = np.isfinite(base) & np.isfinite(got)
    assert fin.sum() > 50
    want = np.array([custom.lnpdf(m) - default.lnpdf(m) for m in pars[fin, 0]])
    assert np.allclose(got[fin] - base[fin], want, rtol=1e-10, atol=1e-10)
    oic = fx.make_oracle_ic(ic)
    ref = oic.lnpost(mod.model_desc(), np.ascontiguousarray(pars.T))
    fx.assert_close(mod.lnpost(pars), ref[0], RTOL, atol=ATOL, what="lnpost with a custom broken prior")
    with pytest.raises(NotImplementedError):
        P.BrokenPrior([P.FlatPrior((0, 1)), P.PowerLawPrior(-2.0, (1.0, 50.0))], [1.0])


def test_lnpost_on_mass_age_feh_distance_av_samples():
    """BASELINE north star, literally: samples given as (mass, age, feh, distance, AV) -> EEP by get_eep on the device
    -> fused lnpost on the device, against the same two steps on the CPU oracle (interp_eep + lnpost), <= 1e-9
    (north star: 1e-6), identical NaN / -inf patterns; all of it on device tensors without a host round trip."""
    import torch
    from oracle import oracle as orc
    ic = ia.synthetic_track(bands=("V",), fehs=np.array([-1.0, -0.5, -0.25, 0.0, 0.25, 0.5]),
                            masses=ia.grids.mist_masses()[30:120:2], eeps=np.arange(150.0, 900.0), eep_bounds=(150, 899))
    mod = ia.SingleStarModel(ic, Teff=(5770, 100), logg=(4.5, 0.1), feh=(0.0, 0.15), V=(10.0, 0.05), parallax=(10.0, 0.2))
    rng = np.random.default_rng(99)
    n = 200_000
    lim = ic.model_grid.get_limits
    mass = rng.uniform(lim("mass")[0] - 0.02, min(lim("mass")[1], 2.5), n)
    age = rng.uniform(8.0, 10.2, n)
    feh = rng.uniform(-1.05, 0.55, n)
    dist, AV = rng.uniform(50.0, 200.0, n), rng.uniform(-0.01, 0.8, n)
    t = lambda x: torch.as_tensor(x, device="cuda")
    eep_dev = ic.get_eep(t(mass), t(age), t(feh))                                        # CUDA tensor
    pars_dev = torch.stack([t(mass), eep_dev, t(feh), t(dist), t(AV)], dim=1).contiguous()
    got = mod.lnpost(pars_dev).cpu().numpy()
    ic._eep_handle(0)                                                                   # makes the ragged age arrays available
    fehs, masses, _ = ic.model_grid.interp.index_columns
    # the reference's interp_eeps returns 1 + the fractional row index (its EEP axis starts at 1); this table's starts at 150
    eep_cpu = orc.interp_eep(age, feh, mass, fehs, masses, ic._age_grid, ic._array_lengths) + (150.0 - 1.0)
    fx.assert_close(eep_dev.cpu().numpy(), eep_cpu, 1e-12, what="EEP of the (mass, age, feh) samples")
    want = fx.make_oracle_ic(ic).lnpost(mod.model_desc(), np.ascontiguousarray(np.stack([mass, eep_cpu, feh, dist, AV])))[0]
    fx.assert_close(got, want, RTOL, atol=ATOL, what="lnpost of (mass, age, feh, distance, AV) samples")
    assert np.isfinite(want).sum() > 20_000 and np.isneginf(want).sum() > 1000


def test_gpu_box_runs_the_library_built_from_these_sources():
    """The prebuilt libiso_hip.so that travelled to the GPU box carries the digest of exactly the sources next to
    it (isochrones_amd/csrc/build.py: source_digest) and is byte for byte the file that build wrote (sha256 in the
    stamp), and it is the file this process has mapped."""
    from isochrones_amd import _cabi
    from isochrones_amd.csrc import build as hip_build
    assert hip_build.built_digest() == hip_build.source_digest()
    # ... and the binary is the one that build linked: its sha256 is stored next to the source digest
    assert hip_build.built_library_sha256() == hip_build.file_sha256(_cabi.library_path())
    _cabi.lib()
    assert os.path.realpath(_cabi.library_path()) in {os.path.realpath(l.split()[-1]) for l in open("/proc/self/maps")
                                                      if "libiso_hip.so" in l}


@pytest.mark.parametrize("n", [1, 2, 100, 256, 257, 8192, 8193, 32768, 32769, 131072 + 5, 1_000_003])
def test_host_array_entry_point_every_size_regime(n, monkeypatch):
    """iso_lnpost_host: the flag-completed single-workgroup call (<= 256 rows), the staged loop (<= 32768 rows) and the
    chunked upload / download pipeline with its helper thread (beyond) all return what the device-pointer entry point
    returns for the same rows, bit for bit - lnpost, and lnprior / lnlike on request."""
    import torch
    import bench
    ic = _ic_for_host_test()
    mod = ia.SingleStarModel(ic, Teff=(5770, 100), logg=(4.5, 0.1), feh=(0.0, 0.15), V=(10.0, 0.05), parallax=(10.0, 0.1))
    rng = np.random.default_rng(n)
    lo = np.array([0.65, 295.0, -1.1, 50.0, 0.0])
    hi = np.array([2.1, 425.0, 0.6, 150.0, 1.0])
    pars = rng.uniform(lo, hi, size=(n, 5))
    if n >= 100:
        pars[rng.integers(0, n, n // 50), rng.integers(0, 5, n // 50)] = np.nan
    dv = torch.as_tensor(pars, device="cuda")
    want, want_prior, want_like = (t.cpu().numpy() for t in mod.evaluate_device(dv, parts=True))
    same = lambda a, b: np.array_equal(a, b, equal_nan=True)
    got = mod.lnpost(pars)
    assert got.shape == (n,) and same(got, want) and np.isfinite(want).sum() > 0
    assert same(mod.lnprior(pars), want_prior) and same(mod.lnlike(pars), want_like)
    assert same(mod.lnpost(pars), want)                                       # again: staging buffers are reused
    if n <= 256:
        monkeypatch.setenv("ISOCHRONES_AMD_HOST_SYNC", "1")                   # the stream-synchronise completion
        assert same(mod.lnpost(pars), want)
    if n > 1:
        assert mod.lnpost(list(pars[1])) == want[1] or (np.isnan(want[1]) and np.isnan(mod.lnpost(list(pars[1]))))


def test_host_array_entry_point_from_two_threads_at_once():
    """ctypes releases the GIL for the call, so two Python threads can be inside iso_lnpost_host of ONE model at the same
    time (its staging areas belong to the model): every call must still return its own rows' values, in all three size
    regimes."""
    import threading
    import torch
    ic = _ic_for_host_test()
    mod = ia.SingleStarModel(ic, Teff=(5770, 100), logg=(4.5, 0.1), feh=(0.0, 0.15), V=(10.0, 0.05), parallax=(10.0, 0.1))
    lo = np.array([0.65, 295.0, -1.1, 50.0, 0.0])
    hi = np.array([2.1, 425.0, 0.6, 150.0, 1.0])
    sizes = [1, 64, 300, 9000, 40_000]
    jobs = []
    for t in range(2):
        rng = np.random.default_rng(100 + t)
        batches = [rng.uniform(lo, hi, size=(n, 5)) for n in sizes]
        want = [mod.evaluate_device(torch.as_tensor(b, device="cuda")).cpu().numpy() for b in batches]
        jobs.append((batches, want))
    bad = []

    def work(t):
        batches, want = jobs[t]
        for rep in range(40):
            for b, w in zip(batches, want):
                if not np.array_equal(mod.lnpost(b), w, equal_nan=True):
                    bad.append((t, rep, len(b)))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not bad, bad[:5]


_HOST_IC = []


def _ic_for_host_test():
    if not _HOST_IC:
        fehs = np.array([-1.0, -0.5, 0.0, 0.5])
        masses = np.array([0.7, 0.9, 1.0, 1.1, 1.3, 2.0])
        _HOST_IC.append(ia.synthetic_track(bands=("V", "J", "K"), fehs=fehs, masses=masses, eeps=np.arange(300.0, 420.0),
                                           limits=dict(mass=(0.7, 2.0), feh=(-1.0, 0.5), age=(5, 10.13)), eep_bounds=(300, 419)))
    return _HOST_IC[0]


def test_tables_built_by_the_ingest_route_drive_the_kernels():
    """Row f1 end to end: raw MIST-like frames -> ingest (bit-identical to the reference's grid classes, CPU test
    test_ingest_golden.py) -> device interpolator -> interp_value / interp_mag / lnpost, against the oracle on the
    dense tables the *reference* built from the same frames (tests/golden/ingest.npz: track_grid, bc_grid)."""
    import pandas as pd
    from isochrones_amd import ingest
    from isochrones_amd.models import BolometricCorrectionGrid, EvolutionTrackGrid, EvolutionTrackInterpolator
    from oracle import oracle as orc
    g = fx.load("ingest")
    raw = pd.DataFrame(g["track_raw"], columns=[str(c) for c in g["track_raw_columns"]])
    model = ingest.model_table_from_raw(raw, tracks=True)
    bands = [str(b) for b in g["bc_bands"]]
    frames = [(g["bc_index"], g["bc_%s_values" % p], [str(c) for c in g["bc_%s_columns" % p]]) for p in ("UBVRIplus", "WISE")]
    bc = ingest.bc_table_from_frames(frames, bands)
    ax = model.index_columns
    ic = EvolutionTrackInterpolator(EvolutionTrackGrid(model, limits=dict(mass=(ax[1][0], ax[1][-1]), feh=(ax[0][0], ax[0][-1]),
                                                                         age=(5, 10.13))),
                                    BolometricCorrectionGrid(bc, bands=bands), bands=bands, eep_bounds=(1, ax[2][-1]))
    # the oracle sees the reference's own arrays (column order of the reference's frames)
    ref_cols = [str(c) for c in g["track_columns"]]
    ref_bands = [str(c) for c in g["bc_columns"]]
    ci = {c: i for i, c in enumerate(ref_cols)}
    oic = orc.OracleIC(0, orc.OracleTable(g["track_grid"], [g["track_axes%d" % k] for k in range(3)]),
                       orc.OracleTable(g["bc_grid"], [g["bc_axes%d" % k] for k in range(4)]),
                       [ci["Teff"], ci["logg"], ci["feh"], ci["Mbol"]], [ci["age"], ci["dt_deep"]], [ci["nu_max"], ci["delta_nu"]])
    rng = np.random.default_rng(5)
    n = 20_000
    pars = np.column_stack([rng.uniform(ax[1][0] - 0.05, ax[1][-1] + 0.05, n), rng.uniform(0.5, ax[2][-1] + 1, n),
                            rng.uniform(ax[0][0] - 0.05, ax[0][-1] + 0.05, n), rng.uniform(5, 500, n), rng.uniform(-0.02, 1.02, n)])
    want_T, want_g, want_f, want_m = oic.interp_mag(pars.T.copy(), [ref_bands.index(b) for b in ("J", "G", "W1")], nthreads=8)
    T, gg, f, m = ic.interp_mag([pars[:, j] for j in range(5)], ["J", "G", "W1"])
    fx.assert_close(T, want_T, 1e-12, what="Teff")
    fx.assert_close(m, want_m, 1e-11, atol=1e-12, what="mags")
    assert np.isfinite(want_m).all(axis=1).sum() > n // 10 and np.isnan(want_T).sum() > n // 50        # ragged: both occur
    mod = ia.SingleStarModel(ic, Teff=(5700, 100), J=(9.0, 0.03), W1=(8.5, 0.05), parallax=(8.0, 0.2))
    desc = mod.model_desc()
    # the descriptor's band columns index this build's BC table; the oracle's table is in the reference's column order
    for k in range(desc.n_bands):
        desc.bc_cols[k] = ref_bands.index(bc.columns[desc.bc_cols[k]])
    want = oic.lnpost(desc, pars.T.copy(), nthreads=8)
    fx.assert_close(mod.lnpost(pars), want[0], RTOL, atol=ATOL, what="lnpost")
    fx.assert_close(mod.lnprior(pars), want[1], RTOL, atol=ATOL, what="lnprior")
    assert np.isfinite(want[0]).sum() > 200
