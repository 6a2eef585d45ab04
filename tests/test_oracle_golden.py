"""CPU: pin the C oracle (oracle/iso_oracle.c) against the golden vectors produced by the
reference itself, and against the reference's own data-free tests / recorded answers."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import _fixtures as fx

RTOL = 1e-12   # same float64 arithmetic in the same order; libm-vs-numpy ulp differences only


def test_kat_3d_reference_test_interp():
    """isochrones/tests/test_interp.py:11-46: exact at a node, matches the trilinear answer."""
    k = fx.load("interp_kats")
    t = orc.OracleTable(k["t3_grid"], [k["t3_axes0"], k["t3_axes1"], k["t3_axes2"]])
    pts = k["t3_pts"]
    got = t.interp([pts[:, 0], pts[:, 1], pts[:, 2]], [0])
    fx.assert_close(got, k["t3_vals"], RTOL, what="3d")
    func = lambda x, y, z: x ** 2 * np.cos(y / 10) + z
    assert got[0, 0] == func(6.0, 50.0, 200.0)            # exact node value (test_interp.py:31)
    from scipy.interpolate import RegularGridInterpolator
    rgi = RegularGridInterpolator([k["t3_axes0"], k["t3_axes1"], k["t3_axes2"]], k["t3_grid"][..., 0])
    assert np.allclose(got[:, 0], rgi(pts), atol=1e-11)   # test_interp.py:35,44-46


def test_kat_2d_docs_notebook():
    """docs/interpolate.ipynb cells 3, 5, 12, 14 (recorded outputs)."""
    k = fx.load("interp_kats")
    axes = [k["t2_axes0"], k["t2_axes1"]]
    full = orc.OracleTable(k["t2_grid"], axes)
    miss = orc.OracleTable(k["t2_grid_missing"], axes)
    assert np.allclose(full.interp([[1.4], [2.1]], [0, 1, 2])[0], k["t2_doc_cell3"], atol=1e-12)
    assert np.allclose(full.interp([[2.2], [4.6]], [1])[0], k["t2_doc_cell5"], atol=1e-12)
    assert np.allclose(miss.interp([[1.3], [2.2]], [0, 1, 2])[0], k["t2_doc_cell12"], atol=1e-12)
    assert np.all(np.isnan(miss.interp([[2.3], [3.0]], [0, 1, 2])[0]))
    q = k["t2_pts"]
    fx.assert_close(full.interp([q[:, 0], q[:, 1]], [0, 1, 2]), k["t2_vals"], RTOL, what="2d")
    fx.assert_close(miss.interp([q[:, 0], q[:, 1]], [0, 1, 2]), k["t2_vals_missing"], RTOL, what="2d missing")


def test_kat_4d_random_table():
    k = fx.load("interp_kats")
    t = orc.OracleTable(k["t4_grid"], [k["t4_axes%d" % i] for i in range(4)])
    p = k["t4_pts"]
    got = t.interp([p[:, i] for i in range(4)], [2, 0])
    fx.assert_close(got, k["t4_vals"], RTOL, atol=1e-14, what="4d")
    assert np.isnan(k["t4_vals"]).any() and np.isfinite(k["t4_vals"]).any()


@pytest.mark.parametrize("case", fx.MODEL_CASES)
def test_model_case(case):
    g = fx.load(case)
    meta = g["meta"]
    ic = fx.make_ic(meta)
    mod = fx.make_model(meta, ic)
    oic = fx.make_oracle_ic(ic)
    N = meta["n_stars"]
    pars = g["pars"]

    # interp_value / interp_mag of the primary
    prim = np.column_stack([pars[:, 0]] + [pars[:, N + j] for j in range(4)]).T.copy()
    ci = ic.model_grid.interp.column_index
    order = ic.param_index_order
    xs = [prim[order[0]], prim[order[1]], prim[order[2]]]
    vals = oic.model.interp(xs, [ci[c] for c in meta["interp_value_cols"]])
    fx.assert_close(vals, g["interp_value"], RTOL, what="interp_value")
    T, lg, fe, mags = oic.interp_mag(prim, [ic.bc_grid.interp.column_index[b] for b in meta["bands"]])
    ok = g["mag_defined"]
    fx.assert_close(T[ok], g["Teff"][ok], RTOL, what="Teff")
    fx.assert_close(lg[ok], g["logg"][ok], RTOL, what="logg")
    fx.assert_close(fe[ok], g["feh"][ok], RTOL, atol=1e-15, what="feh")
    fx.assert_close(mags[ok], g["mags"][ok], RTOL, what="mags")

    # posterior
    desc = mod.model_desc()
    post, prior, like = oic.lnpost(desc, pars.T.copy())
    fx.assert_close(prior, g["lnprior"], 1e-11, atol=1e-12, what="lnprior")
    fx.assert_close(post, g["lnpost"], 1e-11, atol=1e-12, what="lnpost")
    d = ~g["lnlike_undefined"]
    fx.assert_close(like[d], g["lnlike"][d], 1e-11, atol=1e-12, what="lnlike")
    post_only = oic.lnpost(desc, pars.T.copy(), parts=False)
    fx.assert_close(post_only, g["lnpost"], 1e-11, atol=1e-12, what="lnpost (no parts)")

    # mnest_prior
    cube = orc.unit_cube(desc, ic.kind, g["cube_in"].copy())
    fx.assert_close(cube, g["cube_out"], 1e-15, what="unit cube")


def test_threads_agree():
    g = fx.load("iso_binary_phot6")
    ic = fx.make_ic(g["meta"])
    mod = fx.make_model(g["meta"], ic)
    oic = fx.make_oracle_ic(ic)
    a = oic.lnpost(mod.model_desc(), g["pars"].T.copy(), nthreads=1, parts=False)
    b = oic.lnpost(mod.model_desc(), g["pars"].T.copy(), nthreads=max(2, orc.max_threads()), parts=False)
    assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("fixture", ["interp_eep", "interp_eep_plateaus"])
def test_interp_eep_vs_reference(fixture):
    """'next' row f2: the oracle's interp_eep against the reference's interp_eeps.  The second fixture has runs of
    repeated ages inside tracks and 3600 queries that hit table ages exactly: the reference's searchsorted returns
    the equal element its bisection lands on (interp.py:26-29), and the index decides the EEP."""
    g = fx.load(fixture)
    got = orc.interp_eep(g["age"], g["feh"], g["mass"], g["fehs"], g["masses"], g["ages"], g["lengths"])
    fx.assert_close(got, g["eep"], 1e-13, what=fixture)
    assert np.isnan(g["eep"]).any() and np.isfinite(g["eep"]).sum() > 1000


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_model_case_on_fresh_draws_against_the_reference_itself(seed, tmp_path, monkeypatch):
    """Container-only (needs /root/reference): instead of the committed vectors, the reference's StarModel classes are run
    here on freshly drawn cases - parametrisation, multiplicity, observables and their values, bounds keywords, prior
    families and sample points all random - and the oracle is put through the same checks as test_model_case."""
    import os
    if not os.path.isdir("/root/reference/isochrones"):
        pytest.skip("the reference tree is only present in the build container")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(os.path.join(root, "oracle"))
    import make_golden as mg
    fx.tables()                                            # the committed small tables = mg.small_*(): cache them first
    monkeypatch.setattr(mg, "OUT", str(tmp_path))
    monkeypatch.setattr(fx, "GOLDEN", str(tmp_path))
    rng = np.random.default_rng(seed)
    trk, iso, bc = mg.small_track(), mg.small_iso(), mg.small_bc()
    for k in range(3):
        kind = str(rng.choice(["track", "iso"]))
        n_stars = 1 if kind == "track" else int(rng.choice([1, 2, 3]))
        obs = {}
        if rng.random() < 0.6: obs["Teff"] = (float(rng.uniform(4500, 6800)), 100.0)
        if rng.random() < 0.5: obs["logg"] = (float(rng.uniform(3.8, 4.7)), 0.1)
        if rng.random() < 0.5: obs["feh"] = (float(rng.uniform(-0.4, 0.3)), 0.15)
        if rng.random() < 0.6: obs["parallax"] = (float(rng.choice([2.5, 10.0])), float(rng.choice([0.05, 0.5])))
        if rng.random() < 0.3:
            obs["nu_max"] = (float(rng.uniform(800, 3200)), 60.0)
            if rng.random() < 0.6: obs["delta_nu"] = (float(rng.uniform(50, 150)), 2.0)
        for b in rng.choice(list(mg.BANDS), int(rng.integers(0, len(mg.BANDS) + 1)), replace=False):
            obs[str(b)] = (float(rng.uniform(8.5, 11.0)), float(rng.choice([0.002, 0.02, 0.1])))
        if not obs:
            obs["Teff"] = (5700.0, 100.0)
        monkeypatch.setitem(mg.OBS, "fresh", obs)
        kw = {}
        if rng.random() < 0.4: kw["maxAV"] = float(rng.uniform(0.3, 1.2))
        if rng.random() < 0.4: kw["max_distance"] = float(rng.uniform(300, 3000))
        if rng.random() < 0.3: kw["halo_fraction"] = float(rng.uniform(0.0, 0.2))
        pri = {}
        if rng.random() < 0.3: pri["mass"] = [("LogNormal", 0.0, 0.4), ("PowerLaw", -2.35, 0.1, 10.0)][int(rng.integers(0, 2))]
        if rng.random() < 0.3: pri["age"] = [("Gaussian", 9.6, 0.3, 8.0, 10.1), ("Flat", 8.5, 10.1), ("FlatLog", 8.0, 10.0)][int(rng.integers(0, 3))]
        if rng.random() < 0.3: pri["feh"] = [("Flat", -1.5, 0.4), ("Gaussian", -0.2, 0.3, -1.0, 0.5)][int(rng.integers(0, 2))]
        if rng.random() < 0.3: pri["distance"] = [("Gaussian", 150.0, 60.0, 1.0, 600.0), ("LogNormal", float(np.log(300.0)), 0.5), ("PowerLaw", 2.0, 0.0, 500.0)][int(rng.integers(0, 3))]
        if rng.random() < 0.3: pri["AV"] = [("PowerLaw", 0.5, 0.0, 1.0), ("Gaussian", 0.2, 0.1, 0.0, 1.0)][int(rng.integers(0, 2))]
        eep_orig = None
        if rng.random() < 0.25:
            eep_orig = ("Gaussian", 9.6, 0.3, 8.0, 10.1) if kind == "track" else ("LogNormal", float(np.log(0.9)), 0.5)
        name = "fresh_%d_%d" % (seed, k)
        mg.run_model_case(name, kind, n_stars, "fresh", trk if kind == "track" else iso, bc, rng, 90, 90, extra_kw=kw,
                          priors=pri or None, eep_orig_prior=eep_orig)
        test_model_case(name)
