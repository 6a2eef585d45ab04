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
