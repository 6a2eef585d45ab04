"""GPU tests of the catalog / sampler layer: batched multi-star lnpost vs the oracle, the
on-device ensemble sampler driven by the fused kernel, and the end-to-end catalog fit."""
import numpy as np
import pytest

import isochrones_amd as ia
from isochrones_amd.catalog import CatalogPosterior, fit_catalog, synthetic_catalog
from tests import _fixtures as fx

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-11


def _small_track(bands=("G", "BP", "RP")):
    fehs = np.array([-1.0, -0.5, -0.25, 0.0, 0.25, 0.5])
    masses = ia.grids.mist_masses()[25:140:2]
    eeps = np.arange(150.0, 700.0)
    return ia.synthetic_track(bands=bands, fehs=fehs, masses=masses, eeps=eeps, eep_bounds=(150, 699),
                              limits=dict(mass=(masses[0], masses[-1]), feh=(-1.0, 0.5), age=(5, 10.13)))


@pytest.mark.parametrize("kind,n_stars", [("track", 1), ("iso", 2)])
def test_catalog_lnpost_vs_oracle(kind, n_stars):
    import torch
    rng = np.random.default_rng(11)
    bands = ("G", "BP", "RP")
    if kind == "track":
        ic = _small_track(bands)
        lo = np.array([ic.model_grid.masses[0], 150, -1.0, 20.0, 0.0])
        hi = np.array([ic.model_grid.masses[-1], 699, 0.5, 1500.0, 1.0])
    else:
        ages = ia.grids.mist_log_ages()[60::2]
        ic = ia.synthetic_isochrone(bands=bands, ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                                    eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))
        lo = np.array([150.0] * n_stars + [ages[0], -1.0, 20.0, 0.0])
        hi = np.array([699.0] * n_stars + [ages[-1], 0.5, 1500.0, 1.0])
    S = 40
    models = []
    for k in range(S):
        obs = {b: (9.0 + 3 * rng.random(), 0.01 + 0.03 * rng.random()) for b in bands}
        if k % 3:
            obs["parallax"] = (1.0 + 8 * rng.random(), 0.05)
        if k % 2:
            obs["Teff"] = (4500 + 2500 * rng.random(), 80.0)
        models.append(ia.BasicStarModel(ic, N=n_stars, **obs))
    post = CatalogPosterior(ic, models)
    n = 60_000
    pars = rng.uniform(lo, hi, size=(n, lo.size))
    if n_stars > 1:
        pars[:, :n_stars] = -np.sort(-pars[:, :n_stars], axis=1)
    sid = rng.integers(0, S, n).astype(np.int32)
    got = post.lnpost(torch.as_tensor(pars, device="cuda"), torch.as_tensor(sid, device="cuda")).cpu().numpy()
    oic = fx.make_oracle_ic(ic)
    want = np.empty(n)
    for k in range(S):
        m = sid == k
        want[m] = oic.lnpost(models[k].model_desc(), pars[m].T.copy(), parts=False, nthreads=8)
    assert np.isfinite(want).sum() > n // 20
    fx.assert_close(got, want, RTOL, atol=ATOL, what="catalog lnpost")
    # the single-model entry point gives the same numbers for one star's rows
    m = sid == 3
    fx.assert_close(models[3].lnpost(pars[m]), want[m], RTOL, atol=ATOL, what="single-model lnpost")


def test_fit_mcmc_on_device():
    ic = _small_track(("G", "BP", "RP"))
    truth = np.array([1.05, 330.0, -0.1, 200.0, 0.15])
    T, g, f, mags = ic.interp_mag(truth, ["G", "BP", "RP"])
    mod = ia.SingleStarModel(ic, Teff=(T, 80), logg=(g, 0.1), feh=(f, 0.1), G=(mags[0], 0.01), BP=(mags[1], 0.01),
                             RP=(mags[2], 0.01), parallax=(1000 / truth[3], 0.05))
    assert np.isfinite(mod.lnpost(truth))
    sampler = mod.fit_mcmc(nwalkers=64, nburn=300, niter=150, p0=truth, seed=4)
    assert sampler.chain.shape == (64, 150, 5) and sampler.chain.is_cuda
    lp = sampler.flatlnprobability
    assert bool(lp.isfinite().all())
    acc = float(sampler.acceptance_fraction.mean())
    assert 0.1 < acc < 0.8
    s = mod.samples
    assert {"mass", "eep", "feh", "distance", "AV", "lnprob", "Teff", "G_mag"} <= set(s.columns)
    assert abs(s["distance"].median() - truth[3]) < 15.0
    assert abs(s["mass"].median() - truth[0]) < 0.15
    # the chain's best lnpost cannot be far below the truth's
    assert float(lp.max()) > mod.lnpost(truth) - 5.0


def test_fit_catalog_single_gpu_recovers_truth():
    ic = _small_track(("G", "BP", "RP"))
    cat, truth = synthetic_catalog(ic, 48, bands=["G", "BP", "RP"], seed=5, mag_unc=0.01)
    res = fit_catalog(cat, ic, nwalkers=32, nburn=250, niter=100, seed=9)
    assert res.shape == (48, 3 * 5 + 3) and list(res.index) == list(cat.df.index)
    ok = res["ok"].values == 1.0
    assert ok.mean() > 0.9
    assert np.isfinite(res.loc[ok, "lnpost_max"]).all()
    rel_d = np.abs(res.loc[ok, "distance_median"].values - truth.loc[ok, "distance"].values) / truth.loc[ok, "distance"].values
    assert np.median(rel_d) < 0.05                                     # parallax is 2 %
    width = (res.loc[ok, "distance_p84"] - res.loc[ok, "distance_p16"]).values
    assert np.all(width > 0)
    assert 0.05 < res.loc[ok, "acceptance"].median() < 0.9
