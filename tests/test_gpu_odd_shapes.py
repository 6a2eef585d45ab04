"""Table shapes the reference accepts (it bisects every axis alike, isochrones/interp.py:10-35, and sums over any number
of bands, mags.py:35-61) on the FUSED kernels: a third model axis that is not uniform (bisected: every 8th node in
LDS, a 9-node window from the axis itself) and 13-32 bands (band tiles over the corner-packed BC cell).  HIP through the
C ABI vs the CPU oracle, exact NaN / -inf patterns, and the model must report the fused path."""
import numpy as np
import pytest

import isochrones_amd as ia
from tests import _fixtures as fx

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-10


def _thin(eeps, rng, drop):
    keep = np.ones(eeps.size, bool)
    keep[rng.choice(np.arange(1, eeps.size - 1), drop, replace=False)] = False
    return eeps[keep]


def _model(kind, n_stars, obs_bands, eeps, rng):
    fehs = np.array([-2.0, -1.0, -0.5, -0.25, 0.0, 0.25, 0.5])
    bands = obs_bands or ("G",)        # the BC table needs >= 1 column even if no band is observed
    if kind == "track":
        masses = ia.grids.mist_masses()[20:150:3]
        ic = ia.synthetic_track(bands=bands, fehs=fehs, masses=masses, eeps=eeps, eep_bounds=(eeps[0], eeps[-1]),
                                limits=dict(mass=(masses[0], masses[-1]), feh=(-2.0, 0.5), age=(5, 10.13)))
        lo = np.array([masses[0], eeps[0], -2.0, 5.0, 0.0]); hi = np.array([masses[-1], eeps[-1], 0.5, 2000.0, 1.0])
    else:
        ages = ia.grids.mist_log_ages()[40::3]
        ic = ia.synthetic_isochrone(bands=bands, ages=ages, fehs=fehs, eeps=eeps, eep_bounds=(eeps[0], eeps[-1]),
                                    limits=dict(age=(ages[0], ages[-1]), feh=(-2.0, 0.5)))
        lo = np.array([eeps[0]] * n_stars + [ages[0], -2.0, 5.0, 0.0]); hi = np.array([eeps[-1]] * n_stars + [ages[-1], 0.5, 2000.0, 1.0])
    obs = dict(Teff=(5770, 100), logg=(4.4, 0.1), feh=(0.0, 0.15), parallax=(2.0, 0.05))
    for j, b in enumerate(obs_bands):
        obs[b] = (10.0 + 0.1 * j, 0.02)
    return ic, ia.BasicStarModel(ic, N=n_stars, **obs), lo, hi


def _samples(rng, lo, hi, n, n_stars, eeps, kind):
    span = hi - lo
    pars = rng.uniform(lo - 0.02 * span, hi + 0.02 * span, size=(n, lo.size))
    e_cols = [1] if kind == "track" else list(range(n_stars))
    for c in e_cols:                                    # exact node hits, both ends of the axis, just outside
        pars[:4000, c] = rng.choice(eeps, 4000)
        pars[4000:4010, c] = eeps[-1]
        pars[4010:4020, c] = eeps[0]
        pars[4020:4030, c] = np.nextafter(eeps[-1], np.inf)
        pars[4030:4040, c] = np.nextafter(eeps[0], -np.inf)
        pars[4040:4050, c] = np.nan
    if n_stars > 1:
        pars[: n // 2, :n_stars] = -np.sort(-pars[: n // 2, :n_stars], axis=1)
    return pars


@pytest.mark.parametrize("kind,n_stars,nb,spacing", [("track", 1, 1, "thinned"), ("track", 1, 3, "quadratic"), ("iso", 1, 2, "thinned"),
                                                    ("iso", 2, 6, "thinned"), ("iso", 3, 2, "quadratic"), ("track", 1, 0, "short")])
def test_fused_kernel_bisects_a_non_uniform_third_axis(kind, n_stars, nb, spacing):
    rng = np.random.default_rng(4000 + 10 * n_stars + nb)
    base = np.arange(200.0, 900.0) if kind == "track" else np.arange(150.0, 900.0)
    if spacing == "thinned":
        eeps = _thin(base, rng, 40)                      # missing nodes here and there
    elif spacing == "quadratic":
        eeps = base[0] + (base[-1] - base[0]) * np.linspace(0, 1, 333) ** 2     # dense at the start, sparse at the end
    else:
        eeps = np.array([200.0, 201.0, 203.0, 210.0, 260.0, 300.0, 301.5, 420.0, 640.0, 641.0, 899.0])   # 11 nodes: two windows
    bands = ia.grids.DEFAULT_BANDS[:nb]
    ic, mod, lo, hi = _model(kind, n_stars, bands, eeps, rng)
    assert mod.kernel_path() == "fused-packed"
    n = 200_000
    pars = _samples(rng, lo, hi, n, n_stars, eeps, kind)
    oic = fx.make_oracle_ic(ic)
    w_post, w_prior, w_like = oic.lnpost(mod.model_desc(), pars.T.copy(), nthreads=8)
    assert np.isfinite(w_post).sum() > n // 100
    fx.assert_close(mod.lnpost(pars), w_post, RTOL, atol=ATOL, what="lnpost, non-uniform EEP axis")
    fx.assert_close(mod.lnprior(pars), w_prior, RTOL, atol=ATOL, what="lnprior")
    fx.assert_close(mod.lnlike(pars), w_like, RTOL, atol=ATOL, what="lnlike")
    if nb:                                               # the packed interp_mag kernel brackets the same way
        prim = np.column_stack([pars[:, 0]] + [pars[:, n_stars + j] for j in range(4)]).T.copy()
        prim[3] = np.abs(prim[3]) + 1.0
        wT, wg, wf, wm = oic.interp_mag(prim, [ic.bc_grid.interp.column_index[b] for b in bands], nthreads=8)
        T, g_, f, m = ic.interp_mag(list(prim), list(bands))
        fx.assert_close(T, wT, RTOL, what="Teff")
        fx.assert_close(m, wm, RTOL, atol=ATOL, what="mags")


def test_fused_sampler_and_catalog_take_a_non_uniform_third_axis():
    """The sampler kernels share the bracket code: a short fit on a thinned axis runs on the fused sampler and its stored
    lnprob equals the oracle's lnpost of the stored positions."""
    from isochrones_amd.sampler import FusedEnsembleSampler
    rng = np.random.default_rng(7)
    eeps = _thin(np.arange(200.0, 900.0), rng, 60)
    ic, mod, lo, hi = _model("track", 1, ("G", "BP", "RP"), eeps, rng)
    truth = np.array([1.0, 355.0, 0.0, 480.0, 0.1])
    p0 = truth + np.array([0.01, 2.0, 0.02, 4.0, 0.02]) * rng.standard_normal((64, 5))
    p0[:, 4] = np.abs(p0[:, 4])
    assert np.isfinite(mod.lnpost(p0)).all()
    fs = FusedEnsembleSampler(mod, 64, seed=3)
    fs.run_mcmc(p0, 60, store=True)
    chain = fs.chain_steps.cpu().numpy().reshape(-1, 5)
    want = fx.make_oracle_ic(ic).lnpost(mod.model_desc(), chain.T.copy(), parts=False)
    fx.assert_close(fs._lnprob.cpu().numpy().reshape(-1), want, RTOL, atol=ATOL, what="sampler lnprob on a thinned axis")
    assert 0.05 < float(fs.acceptance_fraction.mean()) < 0.9


@pytest.mark.parametrize("kind,n_stars,nb", [("track", 1, 13), ("track", 1, 16), ("iso", 1, 21), ("iso", 2, 13), ("iso", 2, 32),
                                             ("iso", 3, 17), ("track", 1, 32)])
def test_band_tiled_kernel_13_to_32_bands(kind, n_stars, nb):
    rng = np.random.default_rng(5000 + 10 * n_stars + nb)
    bands = tuple(list(ia.grids.KNOWN_BANDS) + ["X%02d" % j for j in range(32)])[:nb]
    eeps = np.arange(200.0, 900.0) if kind == "track" else np.arange(150.0, 900.0)
    ic, mod, lo, hi = _model(kind, n_stars, bands, eeps, rng)
    assert mod.kernel_path() == "fused-packed"
    n = 100_000
    pars = _samples(rng, lo, hi, n, n_stars, eeps, kind)
    oic = fx.make_oracle_ic(ic)
    w_post, w_prior, w_like = oic.lnpost(mod.model_desc(), pars.T.copy(), nthreads=8)
    assert np.isfinite(w_post).sum() > n // 100
    fx.assert_close(mod.lnpost(pars), w_post, RTOL, atol=ATOL, what="lnpost, %d bands" % nb)
    fx.assert_close(mod.lnprior(pars), w_prior, RTOL, atol=ATOL, what="lnprior")
    fx.assert_close(mod.lnlike(pars), w_like, RTOL, atol=ATOL, what="lnlike")
    # the scalar / small-batch host forms go through the same kernel (single workgroup + completion flag)
    k = np.flatnonzero(np.isfinite(w_post))[:3]
    for i in k:
        assert mod.lnpost(list(pars[i])) == pytest.approx(w_post[i], rel=1e-9, abs=1e-10)


def test_band_tiled_kernel_matches_generic_and_is_the_faster_one(monkeypatch):
    """Same model on the generic kernel (ISOCHRONES_AMD_PATH=generic) and on the band-tiled fused kernel: same numbers."""
    rng = np.random.default_rng(99)
    bands = tuple(list(ia.grids.KNOWN_BANDS) + ["X%02d" % j for j in range(12)])[:24]
    eeps = np.arange(200.0, 900.0)
    ic, mod, lo, hi = _model("track", 1, bands, eeps, rng)
    pars = _samples(rng, lo, hi, 60_000, 1, eeps, "track")
    fused = mod.lnpost(pars)
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", "generic")
    ic2, mod2, _, _ = _model("track", 1, bands, eeps, rng)
    assert mod2.kernel_path() == "generic"
    fx.assert_close(fused, mod2.lnpost(pars), 1e-11, atol=1e-11, what="band-tiled vs generic kernel")
