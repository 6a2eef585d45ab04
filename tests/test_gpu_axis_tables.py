"""The fused kernels enter every LDS-staged axis through a bucket table (csrc/fast/axis_lut.h) and bisect only a short
window.  The integer they arrive at must be the one the reference's searchsorted / find_indices produce
(isochrones/interp.py:10-35,116-123) on *any* strictly increasing axis, so this file feeds them axes that are unkind to
bucketing - clustered nodes, log-spaced over ten decades, nodes one ulp apart from a power of two, axes crossing zero,
axes too long for byte tables - and compares lnpost / lnprior / lnlike and interp_mag with the CPU oracle at nodes,
next to nodes and in between.  HIP through the C ABI, exact NaN / -inf patterns."""
import numpy as np
import pytest

import isochrones_amd as ia
from tests import _fixtures as fx

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-10


def _axes(case, rng):
    """(masses, fehs, bc_axes) - every one strictly increasing, none of them friendly"""
    teff, logg, feh4, av = ia.grids.bc_axes()
    if case == "clustered":
        masses = np.unique(np.concatenate([0.1 + 0.9 * rng.random(40) ** 3, 1.0 + 1e-6 * np.arange(12), np.linspace(1.2, 8.0, 20)]))
        fehs = np.array([-2.0, -1.0, -0.999, -0.5, -1e-9, 0.0, 1e-9, 0.25, 0.5])
        teff = np.unique(np.concatenate([2500.0 * 80.0 ** (np.arange(40) / 39.0), 5700.0 + np.arange(20) * 0.5]))
        av = np.array([0.0, 1e-12, 1e-6, 0.01, 0.02, 0.5, 0.50001, 1.0, 6.0])
    elif case == "powers_of_two":
        masses = np.unique(np.concatenate([2.0 ** np.arange(-3.0, 4.0), np.nextafter(2.0 ** np.arange(-3.0, 4.0), 0),
                                           np.nextafter(2.0 ** np.arange(-3.0, 3.0), 100.0), [0.3, 0.7, 1.5, 3.0, 6.0]]))
        fehs = np.array([-2.0, -1.5, -1.0, -0.5, -0.25, -0.125, 0.0, 0.125, 0.25, 0.5])
        logg = np.unique(np.concatenate([logg, [np.nextafter(0.0, 1.0), np.nextafter(4.0, 0.0), np.nextafter(4.0, 9.0)]]))
    elif case == "long":
        masses = np.sort(0.1 + 7.9 * rng.random(300))          # 300 nodes: no byte table, plain bisection
        fehs = np.linspace(-2.0, 0.5, 11)
        teff = 2500.0 * 80.0 ** (np.arange(280) / 279.0)         # 280 nodes
    else:  # "wide"
        masses = np.unique(np.concatenate([10.0 ** np.linspace(-1.0, 0.9, 60), [0.1000001, 0.10001]]))
        fehs = np.array([-2.0, -1.99999, -1.0, 0.0, 0.4999, 0.5])
        logg = np.array([-4.0, -3.9999999, 0.0, 2.0, 3.0, 4.0, 4.0000001, 4.5, 5.0, 8.5])
        av = np.array([0.0, 0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.6, 0.8, 1.0, 2.0, 4.0, 6.0, 60.0, 600.0])
    return masses, fehs, (teff, logg, feh4, av)


def _probe(axis, rng, n):
    a = np.asarray(axis, float)
    near = np.concatenate([a, np.nextafter(a, -np.inf), np.nextafter(a, np.inf)])
    x = np.concatenate([near, rng.uniform(a[0], a[-1], n)])
    return rng.permutation(x)[:n] if x.size > n else np.resize(x, n)


@pytest.mark.parametrize("case", ["clustered", "powers_of_two", "long", "wide"])
@pytest.mark.parametrize("nb", [1, 3])
def test_fused_kernels_on_axes_that_are_unkind_to_bucket_tables(case, nb):
    rng = np.random.default_rng({"clustered": 11, "powers_of_two": 22, "long": 33, "wide": 44}[case] + nb)
    masses, fehs, bc_axes = _axes(case, rng)
    eeps = np.arange(200.0, 500.0)
    bands = ia.grids.DEFAULT_BANDS[:nb]
    ic = ia.synthetic_track(bands=bands, fehs=fehs, masses=masses, eeps=eeps, bc_axes=bc_axes, eep_bounds=(eeps[0], eeps[-1]),
                            limits=dict(mass=(masses[0], masses[-1]), feh=(fehs[0], fehs[-1]), age=(5, 10.13)))
    obs = dict(Teff=(5770, 100), logg=(4.4, 0.1), feh=(0.0, 0.15), parallax=(2.0, 0.05))
    for j, b in enumerate(bands):
        obs[b] = (10.0 + 0.1 * j, 0.02)
    mod = ia.BasicStarModel(ic, **obs)
    assert mod.kernel_path() == "fused-packed"
    n = 120_000
    pars = np.column_stack([_probe(masses, rng, n), rng.uniform(eeps[0], eeps[-1], n), _probe(fehs, rng, n),
                            rng.uniform(5.0, 2000.0, n), _probe(bc_axes[3], rng, n)])
    oic = fx.make_oracle_ic(ic)
    w_post, w_prior, w_like = oic.lnpost(mod.model_desc(), pars.T.copy(), nthreads=8)
    assert np.isfinite(w_like).sum() > n // 200
    fx.assert_close(mod.lnpost(pars), w_post, RTOL, atol=ATOL, what="lnpost (%s axes)" % case)
    fx.assert_close(mod.lnprior(pars), w_prior, RTOL, atol=ATOL, what="lnprior")
    fx.assert_close(mod.lnlike(pars), w_like, RTOL, atol=ATOL, what="lnlike")
    # interp_mag brackets the same way; its (Teff, logg, [Fe/H]) land wherever the synthetic physics puts them, so the BC
    # axes are probed through AV here and through the interpolated surface values above
    prim = np.column_stack([pars[:, 0], pars[:, 1], pars[:, 2], pars[:, 3], pars[:, 4]]).T.copy()
    wT, wg, wf, wm = oic.interp_mag(prim, [ic.bc_grid.interp.column_index[b] for b in bands], nthreads=8)
    T, g_, f, m = ic.interp_mag(list(prim), list(bands))
    fx.assert_close(T, wT, RTOL, what="Teff")
    fx.assert_close(m, wm, RTOL, atol=ATOL, what="mags")
