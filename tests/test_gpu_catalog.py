"""GPU tests of the catalog / sampler layer: batched multi-star lnpost vs the oracle, the
on-device ensemble sampler driven by the fused kernel, and the end-to-end catalog fit."""
import os

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


def test_catalog_whose_stars_have_different_priors(monkeypatch):
    """A catalog built from per-star descriptors may give every star its own priors (each star.ini of a batch_starfit
    folder can set them).  The catalog kernels read the priors the stars share from the first star's block only when the
    library has found them equal; here they are not - two stars carry their own [Fe/H], A_V and mass priors - and both
    the batch kernel and the samplers must use each star's own.  Also: the same catalog with equal priors gives the same
    numbers whether the shared block is used or not (ISOCHRONES_AMD_SHARED_PRIORS=0)."""
    import torch
    from isochrones_amd import priors as P
    from isochrones_amd.sampler import FusedEnsembleSampler
    from isochrones_amd.catalog import initial_positions
    rng = np.random.default_rng(12)
    bands = ("G", "BP", "RP")
    ic = _small_track(bands)
    lo = np.array([ic.model_grid.masses[0], 150, -1.0, 20.0, 0.0])
    hi = np.array([ic.model_grid.masses[-1], 699, 0.5, 1500.0, 1.0])
    truth = np.array([1.0, 355.0, 0.0, 300.0, 0.1])
    mags = ic.interp_mag(list(truth), list(bands))[3]

    def build(different):
        models = []
        for k in range(6):
            obs = {b: (float(mags[j]) + 0.01 * k, 0.02) for j, b in enumerate(bands)}
            obs["parallax"] = (1000.0 / 300.0, 0.05)
            m = ia.BasicStarModel(ic, **obs)
            if different and k == 2:
                m.set_prior(feh=P.FlatPrior((-0.6, 0.3)), AV=P.GaussianPrior(0.2, 0.1, bounds=(0.0, 1.0)))
            if different and k == 4:
                m.set_prior(mass=P.PowerLawPrior(-2.35, (0.2, 3.0)))
            models.append(m)
        return models

    models = build(True)
    post = CatalogPosterior(ic, models)
    n = 30_000
    pars = rng.uniform(lo, hi, size=(n, lo.size))
    pars[: n // 2] = truth + np.array([0.05, 10.0, 0.2, 30.0, 0.1]) * rng.standard_normal((n // 2, 5))
    sid = rng.integers(0, len(models), n).astype(np.int32)
    got = post.lnpost(torch.as_tensor(pars, device="cuda"), torch.as_tensor(sid, device="cuda")).cpu().numpy()
    oic = fx.make_oracle_ic(ic)
    want = np.empty(n)
    for k in range(len(models)):
        m = sid == k
        want[m] = oic.lnpost(models[k].model_desc(), pars[m].T.copy(), parts=False, nthreads=8)
    assert np.isfinite(want).sum() > n // 10
    fx.assert_close(got, want, RTOL, atol=ATOL, what="catalog lnpost, per-star priors")
    # the samplers (step-wise and persistent) on it: stored lnprob = oracle lnpost of the stored positions, star by star
    W = 32
    pos, lnp, failed = initial_positions(post, W, rng_seed=3)
    assert not bool(failed.any())
    for mode in ("stepwise", "persistent"):
        monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", mode)
        fs = FusedEnsembleSampler(post, W, seed=5)
        fs.run_mcmc(pos, 12, lnprob0=lnp, store=True)
        chain = fs.chain.cpu().numpy()                     # [S, W, steps, D]
        lp = fs.lnprobability.cpu().numpy()                # [S, W, steps]
        for k in range(len(models)):
            w = oic.lnpost(models[k].model_desc(), chain[k].reshape(-1, 5).T.copy(), parts=False, nthreads=4)
            fx.assert_close(lp[k].reshape(-1), w, RTOL, atol=ATOL, what="stored lnprob, star %d, %s" % (k, mode))
    post.close()
    # equal priors: with and without the shared block
    same = build(False)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("ISOCHRONES_AMD_SHARED_PRIORS", flag)
        p2 = CatalogPosterior(ic, same)
        outs.append(p2.lnpost(torch.as_tensor(pars, device="cuda"), torch.as_tensor(sid, device="cuda")).cpu().numpy())
        p2.close()
    assert np.array_equal(outs[0], outs[1], equal_nan=True)


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


def test_fused_sampler_consistency_and_statistics():
    """'next' row f3: the single-kernel stretch-move sampler.  (a) its stored lnprob is exactly the
    lnpost of its stored positions (same device function as the batch kernel); (b) never accepts a
    non-finite proposal; (c) samples the same posterior as the framework-op sampler."""
    import torch
    from isochrones_amd.sampler import EnsembleSampler, FusedEnsembleSampler
    ic = _small_track(("G", "BP", "RP"))
    truth = np.array([1.05, 330.0, -0.1, 200.0, 0.15])
    T, g, f, mags = ic.interp_mag(truth, ["G", "BP", "RP"])
    mod = ia.SingleStarModel(ic, Teff=(T, 80), logg=(g, 0.1), feh=(f, 0.1), G=(mags[0], 0.01), BP=(mags[1], 0.01),
                             RP=(mags[2], 0.01), parallax=(1000 / truth[3], 0.05))
    W = 128
    rng = np.random.default_rng(0)
    p0 = truth + np.array([0.01, 1.0, 0.01, 1.0, 0.01]) * rng.standard_normal((W, 5))
    p0[:, 4] = np.abs(p0[:, 4])
    fs = FusedEnsembleSampler(mod, W, seed=7)
    pos, lnp = fs.run_mcmc(p0, 400, store=False)
    fs.reset()
    pos, lnp = fs.run_mcmc(pos, 300, lnprob0=lnp)
    assert fs.chain.shape == (W, 300, 5) and fs.lnprobability.shape == (W, 300)
    # (a) exact bookkeeping
    again = mod.lnpost(pos)
    assert torch.allclose(again, lnp, rtol=1e-12, atol=1e-12)
    flat = fs.flatchain
    again = mod.lnpost(flat)
    assert torch.allclose(again, fs.flatlnprobability, rtol=1e-12, atol=1e-12)
    # (b)
    assert bool(torch.isfinite(fs.flatlnprobability).all())
    acc = float(fs.acceptance_fraction.mean())
    assert 0.15 < acc < 0.8
    # (c) against the framework-op sampler on the same posterior
    ts = EnsembleSampler(W, 5, mod.lnpost, seed=11, device=torch.device("cuda"))
    q, lq = ts.run_mcmc(p0, 400, store=False)
    ts.reset()
    ts.run_mcmc(q, 300, lnprob0=lq)
    a, b = fs.flatchain.cpu().numpy(), ts.flatchain.cpu().numpy()
    sd = b.std(axis=0)
    assert np.all(np.abs(a.mean(axis=0) - b.mean(axis=0)) < 0.35 * sd), (a.mean(axis=0), b.mean(axis=0), sd)
    assert np.all(np.abs(a.std(axis=0) / sd - 1.0) < 0.35)
    # reproducible: same seed, same start -> identical chain
    fs2 = FusedEnsembleSampler(mod, W, seed=7)
    pos2, lnp2 = fs2.run_mcmc(p0, 400, store=False)
    fs2.reset()
    pos2, _ = fs2.run_mcmc(pos2, 300, lnprob0=lnp2)
    assert torch.equal(pos2, pos)


def test_fused_catalog_sampler_matches_per_star_bookkeeping():
    import torch
    from isochrones_amd.sampler import FusedEnsembleSampler
    from isochrones_amd.catalog import initial_positions
    ic = _small_track(("G", "BP", "RP"))
    cat, truth = synthetic_catalog(ic, 12, bands=["G", "BP", "RP"], seed=3, mag_unc=0.01)
    models = list(cat.iter_models(ic))
    post = CatalogPosterior(ic, models)
    pos, lnp, failed = initial_positions(post, 16, rng_seed=1)
    assert not bool(failed.any())
    fs = FusedEnsembleSampler(post, 16, seed=5)
    pos, lnp = fs.run_mcmc(pos, 120, lnprob0=lnp, store=True)
    assert fs.chain.shape == (12, 16, 120, 5)
    for s in (0, 5, 11):                                  # every star's rows follow its own posterior
        want = models[s].lnpost(pos[s])
        assert torch.allclose(want, lnp[s], rtol=1e-12, atol=1e-12)


def _run_fused(target, p0, lnp0, W, nsteps, mode, monkeypatch, seed=9):
    from isochrones_amd.sampler import FusedEnsembleSampler
    monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", mode)
    fs = FusedEnsembleSampler(target, W, seed=seed)
    pos, lnp = fs.run_mcmc(p0, nsteps, lnprob0=lnp0, store=True)
    pos2, lnp2 = fs.run_mcmc(pos, 7, lnprob0=lnp, store=True)      # a second call continues the RNG stream
    return pos2.clone(), lnp2.clone(), fs.chain.clone(), fs.lnprobability.clone(), fs.accepted.clone()


@pytest.mark.parametrize("W", [16, 100, 128, 258, 300, 384, 600])
def test_persistent_sampler_kernel_bit_identical_to_stepwise(W, monkeypatch):
    """The one-launch persistent kernel (workgroup per ensemble, positions in LDS) and the
    launch-per-half-step kernel make the same moves with the same Philox numbers: chains,
    lnprob, final state and acceptance counters must agree bit for bit (W=600 exercises the
    multi-chunk half, W=16 the mostly idle workgroup; 258 / 300 / 384 the three-wave workgroups the
    register-capped form is launched with for 129 ... 192 moves per half-step)."""
    import torch
    from isochrones_amd.catalog import initial_positions
    ic = _small_track(("G", "BP", "RP"))
    cat, truth = synthetic_catalog(ic, 5, bands=["G", "BP", "RP"], seed=4, mag_unc=0.01)
    models = list(cat.iter_models(ic))
    post = CatalogPosterior(ic, models)
    pos, lnp, failed = initial_positions(post, W, rng_seed=2)
    assert not bool(failed.any())
    a = _run_fused(post, pos, lnp, W, 40, "stepwise", monkeypatch)
    for mode in ("persistent", "persistent-dense"):        # uncapped and register-capped instantiations
        b = _run_fused(post, pos, lnp, W, 40, mode, monkeypatch)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert int(a[4].sum()) > 0
    # single-model form
    a = _run_fused(models[2], pos[2], lnp[2], W, 25, "stepwise", monkeypatch)
    b = _run_fused(models[2], pos[2], lnp[2], W, 25, "persistent", monkeypatch)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("N,nb,W", [(2, 3, 300), (1, 8, 300), (3, 2, 260), (1, 3, 384)])
def test_three_wave_workgroups_of_the_register_capped_form(N, nb, W, monkeypatch):
    """The register-capped catalog kernel is launched with THREE waves for ensembles of 129 ... 192 moves per half-step (the
    reference's default 300 walkers), five workgroups per CU instead of four (iso_sampler_run, fast/sampler.h NT): binaries
    and triples, more than six bands (the forms that keep lnpost and the counters in LDS), and a full 192 moves - chains,
    lnprob, final state and acceptance counters bit for bit those of the step-wise kernel and of the four-wave launch
    (ISOCHRONES_AMD_DENSE_THREADS=256)."""
    import torch
    from isochrones_amd.catalog import initial_positions
    bands = list(ia.grids.KNOWN_BANDS[:nb])
    ic = ia.synthetic_isochrone(bands=bands)
    cat, truth = synthetic_catalog(ic, 3, bands=bands, seed=6, mag_unc=0.02, with_parallax=True)
    post = CatalogPosterior.from_catalog(cat, ic, N=N)
    pos, lnp, failed = initial_positions(post, W, rng_seed=4, oversample=8, max_tries=4)
    assert not bool(failed.any())
    a = _run_fused(post, pos, lnp, W, 10, "stepwise", monkeypatch)
    b = _run_fused(post, pos, lnp, W, 10, "persistent-dense", monkeypatch)
    monkeypatch.setenv("ISOCHRONES_AMD_DENSE_THREADS", "256")
    c = _run_fused(post, pos, lnp, W, 10, "persistent-dense", monkeypatch)
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert int(a[4].sum()) > 0
    post.close()


@pytest.mark.parametrize("W,nb,n_ens", [(16, 1, 1), (32, 3, 1), (100, 6, 1), (128, 9, 1), (128, 12, 1), (256, 9, 1), (32, 3, 2), (32, 3, 4), (300, 3, 1)])
def test_single_triple_with_one_star_per_row_makes_the_plain_kernels_chain(W, nb, n_ens, monkeypatch):
    """k_stretch_triple (a single triple with the default priors: lanes l, l + 16, l + 32 of a wave share a move, one star
    each, 64 moves per chunk) against the plain persistent kernel (ISOCHRONES_AMD_STAR_LANES=0) and the step-wise kernel:
    chains, lnprob, final state and acceptance counters bit for bit; the library names the kernel it took - the row form
    up to 64 moves per half-step and workgroup, the plain one beyond.  The 256-walker, 9-band case is the shape whose LDS
    (67 KB laid out for two ensembles) sent it to one launch per half-step until round 6."""
    import torch
    from isochrones_amd import _cabi
    from isochrones_amd.catalog import initial_positions
    bands = list(ia.grids.KNOWN_BANDS[:nb])
    ic = ia.synthetic_isochrone(bands=bands)
    cat, truth = synthetic_catalog(ic, 1, bands=bands, seed=8, mag_unc=0.02, with_parallax=True)
    post = CatalogPosterior.from_catalog(cat, ic, N=3)
    pos, lnp, failed = initial_positions(post, W, rng_seed=5, oversample=8, max_tries=4)
    assert not bool(failed.any())
    post.close()
    mod = cat.model(0, ic, N=3)
    p0, l0 = pos[0], lnp[0]
    if n_ens > 1:
        p0, l0 = p0.unsqueeze(0).repeat(n_ens, 1, 1), l0.unsqueeze(0).repeat(n_ens, 1)

    def run(mode):
        from isochrones_amd.sampler import FusedEnsembleSampler
        monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", mode)
        kw = dict(n_ensembles=n_ens) if n_ens > 1 else {}
        fs = FusedEnsembleSampler(mod, W, seed=9, **kw)
        a, b = fs.run_mcmc(p0, 20, lnprob0=l0, store=True)
        a2, b2 = fs.run_mcmc(a, 5, lnprob0=b, store=True)
        return a2.clone(), b2.clone(), fs.chain.clone(), fs.lnprobability.clone(), fs.accepted.clone()

    monkeypatch.setenv("ISOCHRONES_AMD_STAR_LANES", "0")
    step = run("stepwise")
    plain = run("persistent")
    monkeypatch.delenv("ISOCHRONES_AMD_STAR_LANES")
    _cabi.trace_kernels(True)
    try:
        rows = run("persistent")
        names = _cabi.traced_kernels()
        plan = _cabi.last_sampler_plan()
    finally:
        _cabi.trace_kernels(False)
    assert plan["persistent"] == 1, plan                   # (also the 256-walker, 9-band ensemble: 67 KB of LDS, asked for)
    moves_per_workgroup = min(n_ens, plan["group"]) * W // 2
    assert any(n.startswith("k_stretch_triple<%d>" % nb) for n in names) == (moves_per_workgroup <= 64), (names, plan)
    for x, y, z in zip(step, plain, rows):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert int(rows[4].sum()) > 0
    ic.release()


@pytest.mark.parametrize("W,nb,n_ens", [(16, 1, 1), (32, 3, 1), (100, 6, 1), (128, 11, 1), (200, 2, 1), (256, 4, 1), (32, 3, 4), (300, 3, 1)])
def test_single_binary_with_one_star_per_lane_makes_the_plain_kernels_chain(W, nb, n_ens, monkeypatch):
    """k_stretch_pair (a single binary, lanes l and l + 32 of a wave share a move, one star each) against the plain
    persistent kernel (ISOCHRONES_AMD_STAR_LANES=0) and the step-wise kernel: chains, lnprob, final state and acceptance
    counters bit for bit; the library names the kernel it took (up to 16 moves per wave at any band count, up to 32 at
    <= 4 bands, otherwise - and beyond half a workgroup of moves - the plain kernel)."""
    import torch
    from isochrones_amd import _cabi
    from isochrones_amd.catalog import initial_positions
    bands = list(ia.grids.KNOWN_BANDS[:nb])
    ic = ia.synthetic_isochrone(bands=bands)
    cat, truth = synthetic_catalog(ic, 1, bands=bands, seed=8, mag_unc=0.02, with_parallax=True)
    post = CatalogPosterior.from_catalog(cat, ic, N=2)
    pos, lnp, failed = initial_positions(post, W, rng_seed=5, oversample=8, max_tries=4)
    assert not bool(failed.any())
    post.close()
    mod = cat.model(0, ic, N=2)
    p0, l0 = pos[0], lnp[0]
    if n_ens > 1:
        p0, l0 = p0.unsqueeze(0).repeat(n_ens, 1, 1), l0.unsqueeze(0).repeat(n_ens, 1)

    def run(mode):
        from isochrones_amd.sampler import FusedEnsembleSampler
        monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", mode)
        kw = dict(n_ensembles=n_ens) if n_ens > 1 else {}
        fs = FusedEnsembleSampler(mod, W, seed=9, **kw)
        a, b = fs.run_mcmc(p0, 30, lnprob0=l0, store=True)
        a2, b2 = fs.run_mcmc(a, 7, lnprob0=b, store=True)
        return a2.clone(), b2.clone(), fs.chain.clone(), fs.lnprobability.clone(), fs.accepted.clone()

    monkeypatch.setenv("ISOCHRONES_AMD_STAR_LANES", "0")
    step = run("stepwise")
    plain = run("persistent")
    monkeypatch.delenv("ISOCHRONES_AMD_STAR_LANES")
    _cabi.trace_kernels(True)
    try:
        pair = run("persistent")
        names = _cabi.traced_kernels()
    finally:
        _cabi.trace_kernels(False)
    moves = n_ens * W // 2
    expect_pair = moves <= 128 and (moves <= 64 or nb <= 4)
    assert any(n.startswith("k_stretch_pair<%d>" % nb) for n in names) == expect_pair, names
    for x, y, z in zip(step, plain, pair):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert int(pair[4].sum()) > 0
    ic.release()


@pytest.mark.parametrize("group", ["1", "2", "4", "8"])
def test_persistent_chain_does_not_depend_on_the_ensembles_per_workgroup(group, monkeypatch):
    """A small catalog is spread over more CUs by giving a workgroup fewer ensembles (iso_sampler_run picks the number;
    ISOCHRONES_AMD_PERSIST_GROUP pins it).  The moves are keyed by (step, half, row), not by the lane that makes them:
    chains, lnprob, final state and acceptance counters are bit for bit those of the step-wise kernel."""
    import torch
    from isochrones_amd.catalog import initial_positions
    ic = _small_track(("G", "BP", "RP"))
    cat, truth = synthetic_catalog(ic, 70, bands=["G", "BP", "RP"], seed=14, mag_unc=0.01)
    post = CatalogPosterior.from_catalog(cat, ic)
    W = 32
    pos, lnp, failed = initial_positions(post, W, rng_seed=3)
    assert not bool(failed.any())
    a = _run_fused(post, pos, lnp, W, 30, "stepwise", monkeypatch)
    monkeypatch.setenv("ISOCHRONES_AMD_PERSIST_GROUP", group)
    for mode in ("persistent", "persistent-dense"):
        b = _run_fused(post, pos, lnp, W, 30, mode, monkeypatch)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    monkeypatch.delenv("ISOCHRONES_AMD_PERSIST_GROUP")
    b = _run_fused(post, pos, lnp, W, 30, "auto", monkeypatch)          # the library's own choice for 70 stars
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    post.close()


def test_catalogs_of_one_interpolator_share_a_band_pack_and_outlive_each_other():
    """The corner-packed BC + staged axes of a band list are built once per interpolator and shared by its catalogs
    (reference-counted: closing one catalog must not pull the pack from under another; another band list gets its own)."""
    import torch
    ic = _small_track(("G", "BP", "RP", "J"))
    cat1, _ = synthetic_catalog(ic, 40, bands=["G", "BP", "RP"], seed=1, mag_unc=0.01)
    cat2, _ = synthetic_catalog(ic, 50, bands=["G", "BP", "RP"], seed=2, mag_unc=0.01)
    cat3, _ = synthetic_catalog(ic, 30, bands=["J", "G"], seed=3, mag_unc=0.01)
    p1 = CatalogPosterior.from_catalog(cat1, ic)
    p2 = CatalogPosterior.from_catalog(cat2, ic)
    p3 = CatalogPosterior.from_catalog(cat3, ic)
    rng = np.random.default_rng(0)

    def rows(post, n):
        x = np.array([1.0, 355.0, 0.0, 300.0, 0.1]) + np.array([0.05, 10.0, 0.1, 50.0, 0.05]) * rng.standard_normal((n, 5))
        x[:, 4] = np.abs(x[:, 4])
        sid = rng.integers(0, post.n_models, n).astype(np.int32)
        return torch.as_tensor(x, device="cuda"), torch.as_tensor(sid, device="cuda")
    x2, s2 = rows(p2, 3000)
    x3, s3 = rows(p3, 3000)
    before2, before3 = p2.lnpost(x2, s2).clone(), p3.lnpost(x3, s3).clone()
    p1.close()                                       # the first holder of the (G, BP, RP) pack goes
    torch.cuda.synchronize()
    junk = torch.full((64 << 20,), float("nan"), dtype=torch.float64, device="cuda")      # reuse freed memory if any was freed
    after2, after3 = p2.lnpost(x2, s2), p3.lnpost(x3, s3)
    assert torch.equal(torch.nan_to_num(before2, nan=1.5), torch.nan_to_num(after2, nan=1.5))
    assert torch.equal(torch.nan_to_num(before3, nan=1.5), torch.nan_to_num(after3, nan=1.5))
    del junk
    # and the values are right: star by star against the oracle
    oic = fx.make_oracle_ic(ic)
    got = after2.cpu().numpy()
    xs, ss = x2.cpu().numpy(), s2.cpu().numpy()
    for k in np.unique(ss)[:10]:
        sel = np.flatnonzero(ss == k)
        want = oic.lnpost(cat2.model(int(k), ic).model_desc(), np.ascontiguousarray(xs[sel].T), nthreads=4, parts=False)
        fx.assert_close(got[sel], want, 1e-9, atol=1e-10, what="catalog sharing a band pack")
    p2.close(); p3.close()


def test_persistent_sampler_binary_model_and_bad_mode(monkeypatch):
    import torch
    from isochrones_amd.sampler import FusedEnsembleSampler
    ages = ia.grids.mist_log_ages()[60::2]
    ic = ia.synthetic_isochrone(bands=("J", "H", "K"), ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                                eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))
    truth = np.array([380.0, 330.0, 9.6, -0.1, 300.0, 0.1])
    mags = ic.interp_mag([truth[0], *truth[2:]], ["J", "H", "K"])[3]
    mod = ia.BinaryStarModel(ic, J=(mags[0] - 0.3, 0.02), H=(mags[1] - 0.3, 0.02), K=(mags[2] - 0.3, 0.02),
                             parallax=(1000 / 300.0, 0.05))
    rng = np.random.default_rng(5)
    W = 64
    p0 = truth + np.array([1.0, 1.0, 0.01, 0.01, 1.0, 0.01]) * rng.standard_normal((W, 6))
    p0[:, 5] = np.abs(p0[:, 5])
    lnp0 = mod.lnpost(torch.as_tensor(p0, device="cuda"))
    assert bool(torch.isfinite(lnp0).all())
    a = _run_fused(mod, p0, lnp0, W, 30, "stepwise", monkeypatch)
    b = _run_fused(mod, p0, lnp0, W, 30, "persistent", monkeypatch)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", "bogus")
    fs = FusedEnsembleSampler(mod, W, seed=1)
    with pytest.raises(ia.IsoError):
        fs.run_mcmc(p0, 2, lnprob0=lnp0)


def test_fit_multinest_native_matches_mcmc_posterior(tmp_path):
    """fit_multinest (reference starmodel.py:717-802) through the batched nested sampler: the posterior
    agrees with the ensemble sampler's, the evidence is reproducible within its error, the equal-weight
    file is written, and the evidence equals a brute-force Monte-Carlo integral of exp(lnpost) over the box."""
    import torch
    ic = _small_track(("G", "BP", "RP"))
    truth = np.array([1.05, 330.0, -0.1, 200.0, 0.15])
    T, g, f, mags = ic.interp_mag(truth, ["G", "BP", "RP"])
    mod = ia.SingleStarModel(ic, Teff=(T, 80), logg=(g, 0.1), feh=(f, 0.1), G=(mags[0], 0.02), BP=(mags[1], 0.02),
                             RP=(mags[2], 0.02), parallax=(1000 / truth[3], 0.1), max_distance=1000)
    base = str(tmp_path / "chains" / "single-")
    res = mod.fit_multinest(n_live_points=600, basename=base, seed=1)
    logz, err = mod.evidence
    assert np.isfinite(logz) and 0.02 < err < 0.5 and res.ncall < 5e7
    post = np.loadtxt(base + "post_equal_weights.dat")
    assert post.shape[1] == 6 and post.shape[0] > 200
    s_nest = mod.samples
    assert {"mass", "eep", "feh", "distance", "AV", "lnprob", "Teff", "G_mag"} <= set(s_nest.columns)
    res2 = mod.fit_multinest(n_live_points=600, seed=2, batched=False)          # classic loop: same integral
    assert abs(res2.logz - logz) < 4 * np.hypot(err, res2.logz_err) + 0.05
    # brute force: Z = mean over the prior box of exp(lnpost)
    names = mod.param_names
    lo = np.array([mod.bounds(nm)[0] for nm in names]); hi = np.array([mod.bounds(nm)[1] for nm in names])
    c, w = s_nest[list(names)].mean().values, 6 * s_nest[list(names)].std().values
    blo, bhi = np.maximum(lo, c - w), np.minimum(hi, c + w)          # importance box around the posterior
    rng = np.random.default_rng(3)
    x = torch.as_tensor(rng.uniform(blo, bhi, size=(4_000_000, 5)), device="cuda")
    lp = mod.lnpost(x)
    lp = torch.where(torch.isfinite(lp), lp, torch.full_like(lp, -float("inf")))
    brute = float(torch.logsumexp(lp, 0)) - np.log(x.shape[0]) + np.sum(np.log((bhi - blo) / (hi - lo)))
    assert abs(brute - logz) < 4 * err + 0.1, (brute, logz, err)
    # posterior vs the ensemble sampler
    mod.fit_mcmc(nwalkers=200, nburn=400, niter=200, seed=5)
    s_mc = mod.samples
    for nm in ("mass", "feh", "distance"):
        sd = s_mc[nm].std()
        assert abs(s_nest[nm].mean() - s_mc[nm].mean()) < 0.3 * sd, nm
        assert abs(s_nest[nm].std() / sd - 1) < 0.3, nm


def test_reference_test_fits_flow(tmp_path):
    """reference tests/test_fits.py:27-31,79-100 with its own (tiny) settings: StarModel(ic, **props) ->
    fit_mcmc(nburn=20, niter=20, ninitial=20) -> samples; fit_multinest(n_live_points=5, max_iter=50, basename=...)."""
    ic = ia.get_ichrone("mist", bands=["J", "K"])
    props = dict(Teff=(5800, 100), logg=(4.5, 0.1), J=(3.58, 0.05), K=(3.22, 0.05))
    mod = ia.StarModel(ic, **props)
    mod.fit_mcmc(nburn=20, niter=20, ninitial=20, seed=3)
    s = mod.samples
    mod.fit_mcmc(nburn=20, niter=20, ninitial=20, initial_burn=True, seed=3)      # the reference's re-initialisation
    s2 = mod.samples
    assert len(s2) == len(s) and np.isfinite(s2["lnprob"]).all() and s2["lnprob"].max() >= s["lnprob"].max() - 5
    assert len(s) == 300 * 20 and np.isfinite(s["lnprob"]).all()
    assert {"eep", "age", "feh", "distance", "AV", "Teff", "logg", "J_mag", "K_mag"} <= set(s.columns)
    base = str(tmp_path / "chains" / "123456-")
    res = mod.fit_multinest(n_live_points=5, max_iter=50, basename=base, verbose=False, seed=4)
    assert res.niter <= 50 and np.isfinite(res.logz)
    assert len(mod.samples) >= 1 and np.isfinite(mod.samples["lnprob"]).all()
    assert len(np.atleast_2d(np.loadtxt(base + "post_equal_weights.dat"))) >= 1


def test_chain_quantiles_kernel_matches_numpy():
    """iso_chain_quantiles (LDS bitonic sort per (ensemble, parameter)) vs numpy.percentile on the stored
    chains of a catalog sampler and of a single model, incl. a non-power-of-two sample count."""
    import torch
    from isochrones_amd.sampler import FusedEnsembleSampler
    from isochrones_amd.catalog import initial_positions
    ic = _small_track(("G", "BP", "RP"))
    cat, truth = synthetic_catalog(ic, 7, bands=["G", "BP", "RP"], seed=5, mag_unc=0.01)
    models = list(cat.iter_models(ic))
    post = CatalogPosterior(ic, models)
    for W, nsteps in ((16, 37), (32, 100)):
        pos, lnp, failed = initial_positions(post, W, rng_seed=1)
        fs = FusedEnsembleSampler(post, W, seed=3)
        fs.run_mcmc(pos, nsteps, lnprob0=lnp, store=True)
        got = fs.quantiles((0.5, 0.16, 0.84, 0.0, 1.0)).cpu().numpy()
        flat = fs.flatchain.cpu().numpy()                               # [S, W*nsteps, D]
        want = np.percentile(flat, [50, 16, 84, 0, 100], axis=1)        # [5, S, D]
        assert got.shape == (7, 5, 5)
        assert np.allclose(got, np.moveaxis(want, 0, 2), rtol=1e-14, atol=0)
    fs1 = FusedEnsembleSampler(models[1], 32, seed=4)
    fs1.run_mcmc(pos[1], 50, lnprob0=lnp[1], store=True)
    got = fs1.quantiles((0.5,)).cpu().numpy()
    assert np.allclose(got[:, 0], np.percentile(fs1.flatchain.cpu().numpy(), 50, axis=0), rtol=1e-14, atol=0)


def test_fit_catalog_two_ranks_on_the_gpu(tmp_path):
    """SURVEY 8e end to end with real device fits: two ranks (gloo rendezvous, both on this box's GPU - RCCL needs
    one GPU per rank) - rank 0 builds the tables, broadcast_interpolator ships them, each rank fits the shard
    batch_starfit's rule gives it, all_gather returns the same table everywhere, truth is recovered."""
    import subprocess, sys, socket
    import pandas as pd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ISO_WORLD_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tools", "catalog_world.py"), str(tmp_path), "48"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r0, r1 = pd.read_pickle(tmp_path / "res0.pkl"), pd.read_pickle(tmp_path / "res1.pkl")
    truth = pd.read_pickle(tmp_path / "truth.pkl")
    assert r0.equals(r1) and len(r0) == 48
    assert (r0["ok"] == 1).mean() > 0.9
    rel = np.abs(r0["distance_median"].values - truth["distance"].values) / truth["distance"].values
    assert np.nanmedian(rel) < 0.1


# ---- "next" row f4: generic StarModel over an ObservationTree ---------------------------------
from tests.test_tree_cpu import INI_CASES, INI_DIR, TREE_CASES, make_tree_model  # noqa: E402


@pytest.fixture(params=["auto", "auto-runtime-leaves", "generic"])
def tree_kernel_path(request, monkeypatch):
    """auto: k_lnpost_tree_fast (cooperative gathers on the corner-packed tables; register-resident for 1-4
    stars, or with a runtime star count); generic: k_lnpost_tree."""
    monkeypatch.setenv("ISOCHRONES_AMD_PATH", request.param.split("-")[0])
    if request.param.endswith("runtime-leaves"):
        monkeypatch.setenv("ISOCHRONES_AMD_TREE_RUNTIME_LEAVES", "1")
    return request.param


@pytest.mark.parametrize("case", TREE_CASES + INI_CASES)
def test_tree_model_vs_reference_golden(case, tree_kernel_path):
    import torch
    g = fx.load(case)
    ic, mod = make_tree_model(g["meta"])
    pars = g["pars"]
    fx.assert_close(mod.lnprior(pars), g["lnprior"], RTOL, atol=ATOL, what="lnprior")
    fx.assert_close(mod.lnlike(pars), g["lnlike"], RTOL, atol=1e-9, what="lnlike")
    fx.assert_close(mod.lnpost(pars), g["lnpost"], RTOL, atol=1e-9, what="lnpost")
    k = int(np.flatnonzero(np.isfinite(g["lnpost"]))[0])
    v = mod.lnpost(pars[k])
    assert isinstance(v, float) and np.isclose(v, g["lnpost"][k], rtol=RTOL)
    dev_out = mod.lnpost(torch.as_tensor(pars, device="cuda"))
    fx.assert_close(dev_out.cpu().numpy(), g["lnpost"], RTOL, atol=1e-9, what="lnpost device")


def test_reference_test_ini_checks(tmp_path):
    """reference tests/test_ini.py (IniCheck.check_asserts / check_p0) on the star.ini fixtures: mass-based
    parameter vectors convert to EEPs, lnlike is finite there, every emcee_p0 draw has a finite lnpost;
    plus BasicStarModel's from_ini / write_ini round trip."""
    import os
    checks = {"ini_single": [1.0, 9.4, 0.0, 300, 0.2], "ini_binary": [1.0, 0.7, 9.4, 0.0, 300, 0.2],
              "ini_binary_unassoc": [1.0, 9.4, 0.0, 300, 0.2, 0.8, 9.7, 0.1, 500, 0.3],
              "ini_triple": [1.0, 0.8, 0.7, 9.4, 0.0, 300, 0.2],
              "ini_triple_unassoc1": [1.0, 0.8, 9.4, 0.0, 300, 0.2, 1.0, 9.7, 0.0, 400, 0.5],
              "ini_triple_unassoc2": [1.0, 9.4, 0.0, 300, 0.2, 1.0, 0.8, 9.7, 0.0, 400, 0.5]}
    mist = ia.get_ichrone("mist", bands=["J", "H", "K", "G", "BP", "RP"])      # get_eep needs the companion track grid
    for case, pars in checks.items():
        meta = fx.load(case)["meta"]
        mod = ia.TreeStarModel.from_ini(mist, folder=os.path.join(INI_DIR, meta["ini"]), **meta["from_ini_kwargs"])
        ic = mist
        eep_pars = mod.convert_pars_to_eep(pars)
        assert len(eep_pars) == mod.n_params
        pd_ = mod.obs.p2pardict(eep_pars)
        pm = mod.obs.p2pardict(pars)
        for star, sp in pd_.items():                               # the EEP reproduces the mass asked for
            got = ic.interp_value([sp[0], sp[1], sp[2]], ["mass"])
            assert abs(float(np.ravel(got)[0]) - pm[star][0]) < 0.02
        assert np.isfinite(mod.lnlike(eep_pars)), case
        p0 = mod.emcee_p0(10, rng=np.random.default_rng(3))
        assert p0.shape == (10, mod.n_params) and np.all(np.isfinite(mod.lnpost(p0)))
    # BasicStarModel: flat file, sectioned unresolved file, refusal of resolved companions, round trip
    ic = fx.make_ic(dict(kind="iso", limits=fx.load("ini_flat")["meta"]["limits"],
                         eep_bounds=fx.load("ini_flat")["meta"]["eep_bounds"]))
    flat = ia.BasicStarModel.from_ini(ic, folder=os.path.join(INI_DIR, "flat"))
    assert flat.name == "flat" and set(flat.kwargs) == {"J", "H", "K", "Teff", "parallax"} and flat.N == 1
    assert flat.kwargs["Teff"] == (5800.0, 150.0)
    b2 = ia.BasicStarModel.from_ini(ic, folder=os.path.join(INI_DIR, "flat"), N=2)
    assert b2.N == 2 and len(b2.param_names) == 6
    single = ia.SingleStarModel.from_ini(ic, folder=os.path.join(INI_DIR, "single"))
    assert set(single.bands) == {"J", "H", "K", "G", "RP"} and single.ra == 281.4412
    with pytest.raises(ValueError, match="resolved companions"):
        ia.BasicStarModel.from_ini(ic, folder=os.path.join(INI_DIR, "triple"))
    path = single.write_ini(str(tmp_path))
    again = ia.SingleStarModel.from_ini(ic, folder=os.path.dirname(path))
    assert again.kwargs == single.kwargs and again.ra == single.ra and again.name == single.name
    p = fx.load("ini_single")["pars"][:64]
    assert np.array_equal(again.lnpost(p), single.lnpost(p), equal_nan=True)
    # the basic and the tree model of the same unresolved file agree once they share priors
    tree = ia.TreeStarModel.from_ini(ic, folder=os.path.join(INI_DIR, "single"))
    for k in ["mass", "feh", "age", "distance", "AV", "eep"]:
        single.set_prior(**{k: tree._priors[k]})
    a, b = tree.lnpost(p), single.lnpost(p)
    fin = np.isfinite(b)
    assert fin.sum() > 20 and np.array_equal(np.isfinite(a), fin) and np.allclose(a[fin], b[fin], rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("which", ["all", "spec", "phot"])
def test_reference_test_likelihood_compare_starmodels(which):
    """reference tests/test_likelihood.py:15-58, test for test: the generic StarModel (observation tree built
    from keywords) and BasicStarModel give the same lnlike / lnprior / lnpost for N = 1, 2, 3 once they share
    their prior objects (incl. the EEP prior), at the reference's own parameter vectors."""
    ic = ia.get_ichrone("mist", bands=["J", "K"])
    props = dict(Teff=(5800, 100), logg=(4.5, 0.1), J=(3.58, 0.05), K=(3.22, 0.05), parallax=(100, 0.1))
    if which == "spec":
        props = {k: props[k] for k in ("Teff", "logg", "parallax")}
    elif which == "phot":
        props = {k: props[k] for k in ("J", "K", "parallax")}
    for N, pars in ((1, [300, 9.8, 0.01, 100, 0.1]), (2, [300, 280, 9.8, 0.01, 100, 0.1]),
                    (3, [300, 280, 260.0, 9.8, 0.01, 100, 0.1])):
        m1 = ia.TreeStarModel(ic, N=N, **props)
        m2 = ia.BasicStarModel(ic, N=N, **props)
        for k in ["mass", "feh", "age", "distance", "AV", "eep"]:
            m2.set_prior(**{k: m1._priors[k]})
        assert np.isfinite(m2.lnpost(pars))
        assert np.isclose(m1.lnlike(pars), m2.lnlike(pars), rtol=1e-10)
        assert np.isclose(m1.lnprior(pars), m2.lnprior(pars), rtol=1e-10)
        assert np.isclose(m1.lnpost(pars), m2.lnpost(pars), rtol=1e-10)


def test_tree_model_random_batch_vs_oracle_and_basic_model(tree_kernel_path):
    """A 4-star, 2-system resolved configuration on mid-size tables against the oracle, and the
    keyword (unresolved) form against BasicStarModel evaluated by the fused kernel."""
    from oracle import oracle as orc
    from tests.test_tree_cpu import build_notebook_tree
    rng = np.random.default_rng(31)
    ages = ia.grids.mist_log_ages()[60::2]
    ic = ia.synthetic_isochrone(bands=("J", "H", "K"), ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                                eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))
    mod = ia.TreeStarModel(ic, obs=build_notebook_tree("x"), N=2, index=[0, 1], parallax=(2.0, 0.05), Teff=(5800, 100))
    mod.obs.add_limit(label="1_0", logg=(3.0, None))
    mod._dirty()
    n = 100_000
    cols = []
    for s in range(2):
        e = -np.sort(-rng.uniform(150, 699, size=(n, 2)), axis=1)
        cols += [e, rng.uniform(ages[0], ages[-1], (n, 1)), rng.uniform(-1, 0.5, (n, 1)), rng.uniform(100, 900, (n, 1)),
                 rng.uniform(0, 1, (n, 1))]
    pars = np.hstack(cols)
    want = orc.tree_lnpost(fx.make_oracle_ic(ic), mod.tree_desc(), pars.T.copy(), nthreads=8)
    assert np.isfinite(want[0]).sum() > n // 20
    fx.assert_close(mod.lnpost(pars), want[0], RTOL, atol=1e-9, what="tree lnpost")
    fx.assert_close(mod.lnprior(pars), want[1], RTOL, atol=ATOL, what="tree lnprior")
    fx.assert_close(mod.lnlike(pars), want[2], RTOL, atol=1e-9, what="tree lnlike")
    # keyword form == BasicStarModel (reference tests/test_likelihood.py)
    tree = ia.StarModel(ic, obs=None, J=(13.3, 0.05), K=(12.9, 0.05), parallax=(2.0, 0.1), N=2)
    assert isinstance(tree, ia.BasicStarModel)
    tmod = ia.TreeStarModel(ic, J=(13.3, 0.05), K=(12.9, 0.05), parallax=(2.0, 0.1), N=2)
    basic = ia.BinaryStarModel(ic, J=(13.3, 0.05), K=(12.9, 0.05), parallax=(2.0, 0.1))
    basic.set_bounds(distance=(0, 10000), mass=(0.1, 100.0))
    p6 = pars[:, :6]
    a, b = tmod.lnpost(p6), basic.lnpost(p6)
    fin = np.isfinite(b)
    assert fin.sum() > 1000 and np.allclose(a[fin], b[fin], rtol=1e-9, atol=1e-8)


def test_isotrack_model_vs_reference_golden():
    from tests.test_tree_cpu import _isotrack_objects
    import torch
    g, iso, track, obs = _isotrack_objects()
    mod = ia.IsoTrackModel(iso, track, **obs)
    p = g["pars"]
    fx.assert_close(mod.lnprior(p), g["lnprior"], RTOL, atol=ATOL, what="lnprior")
    fx.assert_close(mod.lnlike(p), g["lnlike"], RTOL, atol=1e-9, what="lnlike")
    fx.assert_close(mod.lnpost(p), g["lnpost"], RTOL, atol=1e-9, what="lnpost")
    k = int(np.flatnonzero(np.isfinite(g["lnpost"]))[0])
    assert np.isclose(mod.lnpost(p[k]), g["lnpost"][k], rtol=RTOL)
    assert mod.lnpost(torch.as_tensor(p, device="cuda")).is_cuda
    # the fit drivers work on the composed model too
    good = p[np.isfinite(g["lnpost"])]
    if len(good):
        mod.fit_mcmc(nwalkers=24, nburn=5, niter=5, p0=good[0], seed=1)
        assert len(mod.samples) == 120 and list(mod.samples.columns[:-1]) == list(mod.param_names)
    res = mod.fit_multinest(n_live_points=60, max_iter=200, seed=2)
    assert np.isfinite(res.logz) and len(mod.samples) >= 1 and np.isfinite(mod.evidence[0])
    # bounds / priors reach the two grid models; the closed-form age prior follows its bounds
    from isochrones_amd import priors
    before = mod.lnprior(good[:8])
    d0, (a_lo, a_hi) = mod.bounds("distance")[1], mod.bounds("age")
    mod.set_bounds(distance=(0, 3000))
    assert mod.bounds("distance") == (0, 3000) and mod._iso_model.bounds("distance") == (0, 3000)
    with pytest.raises(NotImplementedError):
        mod.set_prior(age=priors.FlatPrior((9, 10)))
    lo = float(np.min(good[:8, 2])) - 0.5
    mod.set_prior(age=priors.AgePrior(bounds=(lo, 10.15)))
    after = mod.lnprior(good[:8])
    want = np.log(np.log(10) / (10 ** 10.15 - 10 ** lo)) - np.log(np.log(10) / (10 ** a_hi - 10 ** a_lo))
    want += 3 * np.log(d0 / 3000.0)                 # the d^2 prior renormalises: 3 / hi^3
    assert np.allclose(after - before, want, rtol=1e-9, atol=1e-9)
    assert mod.bands == mod._track_model.bands and mod.labelstring == "single"


def test_tree_model_fits_and_quantile_errors():
    """The generic (observation-tree) model through both fit drivers, and the C-ABI argument checks of
    iso_chain_quantiles."""
    import ctypes as C
    import torch
    from isochrones_amd import _cabi, device as dev
    from tests.test_tree_cpu import build_notebook_tree
    ages = ia.grids.mist_log_ages()[60::2]
    ic = ia.synthetic_isochrone(bands=("J", "H", "K"), ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                                eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))
    mod = ia.StarModel(ic, obs=build_notebook_tree("x"), N=2, index=[0, 1], parallax=(2.0, 0.05), Teff=(5800, 100))
    assert isinstance(mod, ia.TreeStarModel)
    mod.fit_mcmc(nwalkers=40, nburn=20, niter=10, seed=2)
    s = mod.samples
    assert len(s) == 400 and list(s.columns[:-1]) == list(mod.param_names) and np.isfinite(s["lnprob"]).all()
    res = mod.fit_multinest(n_live_points=60, max_iter=400, seed=3)
    assert np.isfinite(res.logz) and res.niter < 400 + 6          # a macro-step retires n_live // 10 points at a time
    lz, err = mod.evidence
    assert lz == res.logz and err > 0 and len(mod.samples) >= 1
    # iso_chain_quantiles refuses bad levels / NULL; any chain length is served
    lib, ctx = _cabi.lib(), dev.context(0)
    x = torch.zeros(4, 8, 3, dtype=torch.float64, device="cuda")
    out = torch.zeros(1, 3, 1, dtype=torch.float64, device="cuda")
    q = (C.c_double * 1)(0.5)
    assert lib.iso_chain_quantiles(ctx, dev.ptr(x), 4, 1, 8, 3, q, 1, dev.ptr(out), None) == 0
    x2 = torch.zeros(2000, 8, 3, dtype=torch.float64, device="cuda")
    assert lib.iso_chain_quantiles(ctx, dev.ptr(x2), 2000, 1, 8, 3, q, 1, dev.ptr(out), None) == 0     # 16000 samples: streamed
    assert lib.iso_chain_quantiles(ctx, dev.ptr(x), 4, 1, 8, 3, (C.c_double * 1)(1.5), 1, dev.ptr(out), None) != 0
    assert lib.iso_chain_quantiles(ctx, None, 4, 1, 8, 3, q, 1, dev.ptr(out), None) != 0


def test_chain_layouts_row_major_and_parameter_major(monkeypatch):
    """iso_sampler_set_chain_layout / iso_chain_quantiles_layout: the sampler makes the same moves whichever way the
    chain is stored (row-major [step][row][param] = the C ABI's default, parameter-major [step][param][row] = what
    FusedEnsembleSampler uses), and the summaries of both layouts are bit-identical, in every kernel form."""
    import ctypes as C
    import torch
    from isochrones_amd import _cabi, device as dev
    from isochrones_amd.catalog import initial_positions
    ic = _small_track(("G", "BP", "RP"))
    cat, _ = synthetic_catalog(ic, 40, bands=["G", "BP", "RP"], seed=3, mag_unc=0.01)
    post = CatalogPosterior.from_catalog(cat, ic)
    W, T, D, S = 32, 60, 5, 40
    pos0, lnp0, failed = initial_positions(post, W, rng_seed=1)
    assert not bool(failed.any())
    lib = _cabi.lib()
    rows = S * W
    q = np.array([0.5, 0.16, 0.84])
    qp = q.ctypes.data_as(C.POINTER(C.c_double))
    res = {}
    for mode in ("stepwise", "persistent"):
        monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", mode)
        for layout in (_cabi.CHAIN_ROW_MAJOR, _cabi.CHAIN_PARAM_MAJOR):
            h = C.c_void_p()
            _cabi.check(lib.iso_sampler_create_catalog(post._h, W, 2.0, 77, C.byref(h)))
            _cabi.check(lib.iso_sampler_set_chain_layout(h, layout))
            pos, lnp = pos0.reshape(rows, D).clone(), lnp0.reshape(rows).clone()
            chain = torch.full((T, rows, D) if layout == _cabi.CHAIN_ROW_MAJOR else (T, D, rows), float("nan"),
                               dtype=torch.float64, device="cuda")
            clnp = torch.empty(T, rows, dtype=torch.float64, device="cuda")
            acc = torch.zeros(rows, dtype=torch.int32, device="cuda")
            _cabi.check(lib.iso_sampler_run(h, dev.ptr(pos), dev.ptr(lnp), T, dev.ptr(chain), dev.ptr(clnp), dev.ptr(acc), None))
            out = torch.empty(S, D, 3, dtype=torch.float64, device="cuda")
            _cabi.check(lib.iso_chain_quantiles_layout(dev.context(0), dev.ptr(chain), layout, T, S, W, D, qp, 3, dev.ptr(out), None))
            torch.cuda.synchronize()
            lib.iso_sampler_destroy(h)
            as_rows = chain if layout == _cabi.CHAIN_ROW_MAJOR else chain.permute(0, 2, 1)
            res[(mode, layout)] = (as_rows.cpu().numpy(), clnp.cpu().numpy(), acc.cpu().numpy(), out.cpu().numpy())
        assert lib.iso_sampler_set_chain_layout(None, 0) != 0
    ref = res[("stepwise", _cabi.CHAIN_ROW_MAJOR)]
    assert np.isfinite(ref[0]).all() and ref[2].sum() > 0.1 * rows * T
    for key, got in res.items():
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), key
    want = np.percentile(ref[0].reshape(T, S, W, D).transpose(1, 0, 2, 3).reshape(S, T * W, D), [50, 16, 84], axis=1)
    assert np.array_equal(ref[3], want.transpose(1, 2, 0))
    for mode_q in ("workgroup", "sort"):                       # the older summary kernels read both layouts too
        monkeypatch.setenv("ISOCHRONES_AMD_QUANTILES", mode_q)
        for layout in (_cabi.CHAIN_ROW_MAJOR, _cabi.CHAIN_PARAM_MAJOR):
            c = torch.as_tensor(ref[0] if layout == _cabi.CHAIN_ROW_MAJOR else np.ascontiguousarray(ref[0].transpose(0, 2, 1)),
                                device="cuda")
            out = torch.empty(S, D, 3, dtype=torch.float64, device="cuda")
            _cabi.check(lib.iso_chain_quantiles_layout(dev.context(0), dev.ptr(c), layout, T, S, W, D, qp, 3, dev.ptr(out), None))
            assert np.array_equal(out.cpu().numpy(), ref[3])
    h = C.c_void_p()
    _cabi.check(lib.iso_sampler_create_catalog(post._h, W, 2.0, 77, C.byref(h)))
    assert lib.iso_sampler_set_chain_layout(h, 7) != 0
    lib.iso_sampler_destroy(h)
    post.close()


def test_derived_samples_single_and_binary():
    """reference _make_samples (starmodel.py:1653-1707): per-component columns and combined magnitudes."""
    ages = ia.grids.mist_log_ages()[60::2]
    ic = ia.synthetic_isochrone(bands=("J", "K"), ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                                eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))
    truth = [380.0, 9.6, -0.1, 300.0, 0.1]
    mags = ic.interp_mag(truth, ["J", "K"])[3]
    one = ia.SingleStarModel(ic, J=(mags[0], 0.02), K=(mags[1], 0.02), parallax=(1000 / 300.0, 0.05))
    one.fit_mcmc(nwalkers=40, nburn=30, niter=10, seed=1)
    d = one.derived_samples
    assert len(d) == 400 and {"Teff", "logg", "mass", "J_mag", "K_mag", "parallax", "distance", "AV"} <= set(d.columns)
    assert np.allclose(d["parallax"], 1000.0 / one.samples["distance"])
    two = ia.BinaryStarModel(ic, J=(mags[0] - 0.4, 0.02), K=(mags[1] - 0.4, 0.02), parallax=(1000 / 300.0, 0.05))
    two.fit_mcmc(nwalkers=40, nburn=30, niter=10, seed=2)
    d2 = two.derived_samples
    want = {"eep_0", "eep_1", "age", "feh", "Teff_0", "Teff_1", "mass_0", "mass_1", "J_mag_0", "J_mag_1", "J_mag", "K_mag",
            "parallax", "distance", "AV", "lnprob"}
    assert want <= set(d2.columns) and "eep" not in d2.columns
    comb = -2.5 * np.log10(10 ** (-0.4 * d2["J_mag_0"]) + 10 ** (-0.4 * d2["J_mag_1"]))
    assert np.allclose(d2["J_mag"], comb) and np.all(d2["J_mag"] <= d2["J_mag_0"] + 1e-12)
    assert np.all(d2["mass_0"] >= d2["mass_1"] - 1e-9)          # eep_0 >= eep_1 at one age and composition
    # summaries over the samples (reference starmodel.py:1755-1841)
    assert set(one.physical_quantities) <= set(d.columns) and set(two.physical_quantities) <= set(d2.columns)
    assert one.observed_quantities == ["J_mag", "K_mag", "parallax"] and set(two.observed_quantities) <= set(d2.columns)
    assert np.isfinite(one.posterior_predictive) and one.posterior_predictive > 0 and two.posterior_predictive > 0   # short chains
    mp = one.map_pars
    assert mp.shape == (5,) and np.isclose(one.lnpost(mp), one.samples["lnprob"].max())


def test_sample_from_prior_reference_semantics():
    """reference starmodel.py:1716-1748 / priors.py:431-463: DataFrame of valid prior draws, EEPs resampled
    with the EEP-prior weights (integers within the EEP bounds), values=True gives the plain array."""
    import pandas as pd
    ic = _small_track(("G", "BP", "RP"))
    mod = ia.SingleStarModel(ic, G=(10.0, 0.05), parallax=(5.0, 0.2))
    df = mod.sample_from_prior(500, rng=np.random.default_rng(1))
    assert isinstance(df, pd.DataFrame) and list(df.columns) == list(mod.param_names) and len(df) == 500
    assert np.isfinite(mod.lnpost(df.values)).all()
    assert np.all(df["eep"] == np.round(df["eep"])) and df["eep"].between(*mod.bounds("eep")).all()
    # weights = AgePrior(log age) x d log age / d EEP with a prior flat in linear age: old (late) points dominate,
    # i.e. the draws are visibly not uniform in EEP
    assert (df["eep"] > 425).mean() > 0.7
    arr = mod.sample_from_prior(7, values=True, rng=np.random.default_rng(2))
    assert isinstance(arr, np.ndarray) and arr.shape == (7, 5)
    assert len(mod.sample_from_prior(0)) == 0
    assert mod.emcee_p0(16, rng=np.random.default_rng(3)).shape == (16, 5)


@pytest.mark.parametrize("N", [2, 3])
def test_fit_catalog_multiple_star_models(N):
    """Catalog fits with binary / triple models (reference: StarCatalog.iter_models(ic, N=2|3)): the MULTI kernels
    for NS = 2, 3 under the batched sampler recover the distance of synthetic unresolved binaries."""
    import pandas as pd
    ages = ia.grids.mist_log_ages()[60::2]
    ic = ia.synthetic_isochrone(bands=("J", "H", "K"), ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                                eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))
    rng = np.random.default_rng(3)
    rows = []
    for i in range(16):
        e0 = rng.uniform(300, 450); e1 = rng.uniform(250, e0); age = rng.uniform(9.2, 9.9); feh = rng.uniform(-0.5, 0.3)
        d = rng.uniform(100, 600)
        m0 = ic.interp_mag([e0, age, feh, d, 0.1], ["J", "H", "K"])[3]
        m1 = ic.interp_mag([e1, age, feh, d, 0.1], ["J", "H", "K"])[3]
        mags = -2.5 * np.log10(10 ** (-0.4 * m0) + 10 ** (-0.4 * m1))
        rows.append({"J_mag": mags[0], "J_mag_unc": 0.02, "H_mag": mags[1], "H_mag_unc": 0.02, "K_mag": mags[2],
                     "K_mag_unc": 0.02, "parallax": 1000 / d, "parallax_unc": 0.05, "true_d": d})
    df = pd.DataFrame(rows)
    cat = ia.StarCatalog(df.drop(columns=["true_d"]), props=["parallax"])
    res = fit_catalog(cat, ic, N=N, nwalkers=32, nburn=150, niter=60, seed=1)
    assert list(res.columns[:3]) == ["eep_0_median", "eep_0_p16", "eep_0_p84"] and res.shape == (16, 3 * (N + 4) + 3)
    assert res["ok"].mean() == 1.0
    assert np.all(res["eep_0_median"] >= res["eep_%d_median" % (N - 1)])
    rel = np.abs(res["distance_median"].values - df["true_d"].values) / df["true_d"].values
    assert np.median(rel) < 0.02


def test_catalog_rows_with_missing_bands():
    """A star without a measurement in some band: the catalog kernels skip that term (NaN observed magnitude),
    which is the posterior of the per-star model the reference would build without that band."""
    import torch
    ic = _small_track(("G", "BP", "RP"))
    cat, truth = synthetic_catalog(ic, 12, bands=["G", "BP", "RP"], seed=8, mag_unc=0.01)
    df = cat.df.copy()
    df.loc[df.index[0], "BP_mag"] = np.nan                     # also the first row: the template must skip it
    df.loc[df.index[4], "RP_mag_unc"] = np.nan
    df.loc[df.index[7], ["G_mag", "BP_mag"]] = np.nan
    cat2 = ia.StarCatalog(df, bands=["G", "BP", "RP"], props=list(cat.props))
    post = CatalogPosterior.from_catalog(cat2, ic, N=1)
    rng = np.random.default_rng(2)
    n = 4000
    lo = np.array([ic.model_grid.masses[0], 150, -1.0, 20.0, 0.0]); hi = np.array([ic.model_grid.masses[-1], 699, 0.5, 1500.0, 1.0])
    x = torch.as_tensor(rng.uniform(lo, hi, size=(n, 5)), device="cuda")
    models = list(cat2.iter_models(ic))
    assert models[0].bands == ["G", "RP"] and models[4].bands == ["G", "BP"] and models[7].bands == ["RP"]
    for s in (0, 4, 7, 2):
        sid = torch.full((n,), s, dtype=torch.int32, device="cuda")
        got = post.lnpost(x, sid).cpu().numpy()
        want = models[s].lnpost(x).cpu().numpy()
        fx.assert_close(got, want, 1e-11, atol=1e-11, what="star %d" % s)
    res = fit_catalog(cat2, ic, nwalkers=32, nburn=100, niter=50, seed=1)
    assert res["ok"].mean() == 1.0
    sliced = fit_catalog(cat2, ic, nwalkers=32, nburn=100, niter=50, seed=1, max_stars_per_batch=5)   # 5 + 5 + 2 stars
    assert sliced.shape == res.shape and sliced["ok"].mean() == 1.0
    assert np.allclose(sliced["distance_median"], res["distance_median"], rtol=0.2)


def test_catalog_columns_path_equals_descriptor_path():
    """iso_catalog_create_columns (template + per-star columns, constant blocks filled by a kernel) against
    iso_catalog_create (one host descriptor per star): same lnpost for every star, incl. stars without a
    parallax / with a negative one / without Teff, and a catalog-wide custom prior."""
    import pandas as pd
    import torch
    from isochrones_amd import priors as P
    rng = np.random.default_rng(12)
    ic = _small_track(("G", "RP"))
    n = 30
    df = pd.DataFrame({"G_mag": 10 + rng.random(n), "G_mag_unc": 0.01 + 0.01 * rng.random(n),
                       "RP_mag": 9.5 + rng.random(n), "RP_mag_unc": 0.02,
                       "parallax": 1 + 5 * rng.random(n), "parallax_unc": 0.05,
                       "Teff": 5000 + 1000 * rng.random(n), "Teff_unc": 80.0})
    df.loc[3, "parallax"] = np.nan
    df.loc[5, "parallax"] = -0.2
    df.loc[7, "Teff"] = np.nan
    for custom in (False, True):
        cat = ia.StarCatalog(df, bands=["G", "RP"], props=["parallax", "Teff"])
        if custom:
            cat.set_prior(feh=P.FlatPrior((-0.8, 0.3)), AV=P.PowerLawPrior(0.5, (0.0, 1.0)))
        a = CatalogPosterior.from_catalog(cat, ic)                                      # columns
        arr, template = CatalogPosterior.build_descs(cat, ic)
        b = CatalogPosterior(ic, _descs=arr, _template=template)                        # descriptors
        assert np.array_equal(a.bounds_hi, b.bounds_hi) and np.array_equal(a.bounds_lo, b.bounds_lo)
        assert np.array_equal(a.parallax, b.parallax, equal_nan=True)
        m = 3000
        lo = np.array([ic.model_grid.masses[0], 150, -1.0, 20.0, 0.0]); hi = np.array([ic.model_grid.masses[-1], 699, 0.5, 2500.0, 1.0])
        x = torch.as_tensor(rng.uniform(lo, hi, size=(m * n, 5)), device="cuda")
        sid = torch.arange(n, dtype=torch.int32, device="cuda").repeat_interleave(m)
        fx.assert_close(a.lnpost(x, sid).cpu().numpy(), b.lnpost(x, sid).cpu().numpy(), 1e-13, atol=1e-13, what="columns vs descs")
        assert np.isfinite(a.lnpost(x, sid).cpu().numpy()).sum() > m
        a.close(); b.close()


@pytest.mark.parametrize("mode", ["auto", "wave", "workgroup", "sort"])
def test_chain_quantiles_adversarial_inputs(mode, monkeypatch):
    """iso_chain_quantiles on hand-made chains (auto: the wave-per-pair selection kernels - the compile-time-shaped form
    for 12 / 25 / 50 / 100 full registers of 64 values with or without a partial one, the generic form otherwise - with
    their flagged hand-over; wave: the generic wave kernel everywhere; the workgroup selection kernel; the full LDS
    sort): smooth data, constant chains, heavy ties (overflow the selection lists: handed to the workgroup kernel / its
    sort), tiny and odd sample counts, more values than the wave kernel holds, infinities, both chain layouts; always
    numpy.quantile's numbers, bit for bit."""
    import ctypes as C
    import torch
    from isochrones_amd import _cabi, device as dev
    if mode != "auto":
        monkeypatch.setenv("ISOCHRONES_AMD_QUANTILES", mode)
    lib, ctx = _cabi.lib(), dev.context(0)
    rng = np.random.default_rng(21)
    qs = np.array([0.5, 0.16, 0.84, 0.0, 1.0, 0.999, 0.3333])
    for nsteps, S, W, D in ((100, 9, 32, 5), (37, 4, 16, 3), (1, 3, 2, 2), (255, 2, 32, 6), (3, 5, 1, 1), (104, 3, 32, 2),
                            (60, 3, 48, 2), (30, 2, 100, 3), (7, 2, 65, 2), (200, 5, 32, 3), (208, 2, 32, 2), (209, 2, 32, 2),
                            (100, 1031, 32, 5),
                            # the compile-time-shaped kernel: 12 / 25 / 50 / 100 full registers, with and without a tail, W | 64
                            (25, 3, 32, 3), (50, 4, 32, 3), (51, 3, 32, 2), (101, 3, 32, 3), (201, 3, 32, 2), (400, 3, 8, 2),
                            (100, 3, 64, 2), (1600, 3, 2, 2), (24, 3, 32, 2), (200, 3, 16, 3), (3201, 3, 1, 2),
                            # more than 8 192 values per pair: selection by refinement, streamed from the chain
                            # (k_chain_quantiles_big; 300 walkers x 100 iterations is the reference's default fit)
                            (100, 4, 300, 3), (129, 3, 64, 2), (300, 3, 100, 3), (1000, 3, 300, 2), (8193, 3, 1, 2)):
        x = rng.standard_normal((nsteps, S * W, D))
        if S > 1000:
            x[:, 5 * W:6 * W, 3] = np.exp(3 * x[:, 5 * W:6 * W, 3])      # long tail: most values share the first bins
        x[:, :W, 0] = 3.25                                        # ensemble 0, parameter 0: constant
        if D > 1:
            x[:, :W, 1] = rng.integers(0, 3, size=(nsteps, W))   # heavy ties
        if S > 1 and D > 2:
            x[:, W:2 * W, 2] = np.round(x[:, W:2 * W, 2], 1)      # moderate ties
        if S > 2:
            x[0, 2 * W, 0] = np.inf                               # one infinite sample
            x[-1, 2 * W, 0] = -1e300
        chain = torch.as_tensor(x, device="cuda")
        out = torch.zeros(S, D, qs.size, dtype=torch.float64, device="cuda")
        rc = lib.iso_chain_quantiles(ctx, dev.ptr(chain), nsteps, S, W, D, qs.ctypes.data_as(C.POINTER(C.c_double)), qs.size,
                                     dev.ptr(out), None)
        assert rc == 0
        got = out.cpu().numpy()
        flat = x.reshape(nsteps, S, W, D).transpose(1, 3, 0, 2).reshape(S, D, nsteps * W)
        with np.errstate(invalid="ignore"):
            want = np.moveaxis(np.quantile(flat, qs, axis=2), 0, 2)          # same levels, no x100/100 round trip
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin), (nsteps, S, W, D)
        assert np.array_equal(got[fin], want[fin]), (nsteps, S, W, D, np.abs(got[fin] - want[fin]).max())      # bit for bit
        assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)])
        # the same chain stored parameter-major [step][param][row] (what the sampler writes)
        chain_pm = torch.as_tensor(np.ascontiguousarray(x.transpose(0, 2, 1)), device="cuda")
        out2 = torch.zeros_like(out)
        rc = lib.iso_chain_quantiles_layout(ctx, dev.ptr(chain_pm), _cabi.CHAIN_PARAM_MAJOR, nsteps, S, W, D,
                                            qs.ctypes.data_as(C.POINTER(C.c_double)), qs.size, dev.ptr(out2), None)
        assert rc == 0 and np.array_equal(out2.cpu().numpy(), got, equal_nan=True), (nsteps, S, W, D, "parameter-major")


def test_fit_multinest_binary_model_respects_ordering():
    """Nested fit of a BinaryStarModel: the flat-box prior covers both orderings of the two EEPs, lnpost is -inf
    for eep_1 > eep_0 (starmodel.py:1618-1620), so every posterior sample must be ordered and about half of the
    box has no support."""
    ages = ia.grids.mist_log_ages()[60::2]
    ic = ia.synthetic_isochrone(bands=("J", "H", "K"), ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                                eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))
    truth = np.array([380.0, 330.0, 9.6, -0.1, 300.0, 0.1])
    m0 = ic.interp_mag([truth[0], *truth[2:]], ["J", "H", "K"])[3]
    m1 = ic.interp_mag([truth[1], *truth[2:]], ["J", "H", "K"])[3]
    mags = -2.5 * np.log10(10 ** (-0.4 * m0) + 10 ** (-0.4 * m1))
    mod = ia.BinaryStarModel(ic, J=(mags[0], 0.02), H=(mags[1], 0.02), K=(mags[2], 0.02), parallax=(1000 / 300.0, 0.05))
    res = mod.fit_multinest(n_live_points=300, seed=4)
    s = mod.samples
    assert np.isfinite(res.logz) and res.prior_fraction < 0.6
    assert np.all(s["eep_0"] >= s["eep_1"]) and np.isfinite(s["lnprob"]).all()
    assert abs(np.median(s["distance"]) - 300.0) < 30.0
    d = mod.derived_samples
    assert {"mass_0", "mass_1", "J_mag"} <= set(d.columns)


def test_save_load_round_trip_and_convenience_helpers(tmp_path):
    """reference StarModel.save_hdf / load_hdf (starmodel.py:1205-1317, 1843-1959), maxlike (:821-833),
    random_samples (:1055-1069), fit dispatch (:667-671): a fitted basic model and a fitted tree model come
    back from disk with the same measurements, priors, bounds, samples and lnpost."""
    import os
    from isochrones_amd import priors
    from tests.test_tree_cpu import build_notebook_tree
    ic = ia.get_ichrone("mist", bands=["J", "H", "K"])
    mod = ia.BinaryStarModel(ic, J=(9.6, 0.03), H=(9.2, 0.03), K=(9.1, 0.03), parallax=(8.0, 0.1), name="pair",
                             use_emcee=True)
    mod.set_prior(feh=priors.FlatPrior((-1.0, 0.4)), AV=priors.GaussianPrior(0.1, 0.05, bounds=(0, 1)))
    mod.set_bounds(eep=(200, 500))
    mod.fit(nwalkers=32, nburn=20, niter=10, seed=4)                        # use_emcee -> fit_mcmc
    f = str(tmp_path / "pair.npz")
    assert mod.save_hdf(f) == f and os.path.exists(f)
    with pytest.raises(IOError):
        mod.save_hdf(f)
    with pytest.raises(ImportError, match="pytables"):
        mod.save_hdf(str(tmp_path / "pair.h5"))
    back = ia.BinaryStarModel.load_hdf(f, ic=ic)
    assert type(back) is ia.BinaryStarModel and back.name == "pair" and back.kwargs == mod.kwargs
    assert back._bounds == mod._bounds and back.bounds("eep") == (200, 500)
    assert type(back._priors["feh"]) is priors.FlatPrior and back._priors["AV"].bounds == (0, 1)
    assert list(back.samples.columns) == list(mod.samples.columns)
    assert np.array_equal(back.samples.values, mod.samples.values)
    assert np.array_equal(back.derived_samples.values, mod.derived_samples.values, equal_nan=True)
    p = mod.samples[list(mod.param_names)].values[:50]
    assert np.array_equal(back.lnpost(p), mod.lnpost(p))
    assert len(back.random_samples(17, rng=np.random.default_rng(0))) == 17
    also = ia.BasicStarModel.load(f)                                        # grid rebuilt from the stored bands
    assert also.N == 2 and np.array_equal(also.lnpost(p), mod.lnpost(p))
    with pytest.raises(TypeError):
        ia.TreeStarModel.load(f, ic=ic)
    # maxlike: Nelder-Mead from the best prior draw does at least as well as that draw and every sample
    best = mod.maxlike(n_starts=2048, seed=1, options=dict(maxiter=400))
    assert np.isfinite(best.fun) and -best.fun >= np.max(mod.samples["lnprob"].values) - 5.0
    assert mod.prior("AV", 0.1) == mod._priors["AV"](0.1) and mod.mags["J"] == 9.6 and mod.directory == "."
    # tree model: nested fit (fit() without use_emcee), evidence survives the round trip
    tree = ia.TreeStarModel(ic, obs=build_notebook_tree("nb"), parallax=(2.0, 0.05), Teff=(5834.0, 100), name="nb")
    tree.set_prior(AV=priors.FlatPrior((0, 0.5)))
    tree.fit(n_live_points=60, max_iter=150, seed=2)
    g = str(tmp_path / "tree.npz")
    tree.save(g)
    tb = ia.TreeStarModel.load(g, ic=ic)
    assert tb.param_names == tree.param_names and tb.obs.leaf_labels == tree.obs.leaf_labels
    assert tb.evidence == tree.evidence and tb.labelstring == "binary" and tb.props == []
    q = tree.samples[list(tree.param_names)].values[:40]
    # the tree rebuilt from the stored rows may list same-resolution observations in another order: the same terms
    # summed in a different order, so equal to rounding, not bit for bit
    assert np.allclose(tb.lnpost(q), tree.lnpost(q), rtol=1e-12, atol=0)
    assert np.array_equal(tb.samples.values, tree.samples.values)
    cube = [0.2, 0.7, 0.5, 0.4, 0.3, 0.1]                      # EEPs out of order: mnest_prior sorts them (starmodel.py:644-656)
    tree.mnest_prior(cube)
    box = tree.prior_transform(np.array([0.7, 0.2, 0.5, 0.4, 0.3, 0.1]))
    assert np.allclose(cube, box) and cube[0] > cube[1]
    assert np.isclose(tree.mnest_loglike(cube), tree.lnpost(np.array(cube)))
    # the nested fit maps the cube the same way, so its samples are ordered and the whole cube carries prior mass
    e = tree.samples[list(tree.param_names)].values
    assert (e[:, 0] >= e[:, 1]).all()
    unfit = ia.SingleStarModel(ic, J=(9.6, 0.03))
    unfit.save(str(tmp_path / "unfit.npz"))
    assert ia.SingleStarModel.load(str(tmp_path / "unfit.npz"), ic=ic)._samples is None


def test_starfit_driver_on_an_ini_folder(tmp_path):
    """reference isochrones/starfit.py: star.ini folder -> model per multiplicity, fitted and stored next to
    the ini; a second call loads instead of refitting."""
    import os
    import shutil
    folder = tmp_path / "KOI-1"
    shutil.copytree(os.path.join(INI_DIR, "flat"), str(folder))
    mod = ia.starfit(str(folder), multiplicities=["single", "binary"], n_live_points=80, max_iter=120, seed=5,
                     feh_prior="flat")
    assert mod.N == 2 and mod.name == "KOI-1" and isinstance(mod._priors["feh"], ia.priors.FlatPrior)
    assert sorted(f for f in os.listdir(str(folder)) if f.endswith(".npz")) == \
        ["mist_starmodel_binary.npz", "mist_starmodel_single.npz"]
    logz = mod.evidence
    again = ia.starfit(str(folder), multiplicities=["binary"])
    assert again.evidence == logz and np.array_equal(again.samples.values, mod.samples.values)
    single = ia.SingleStarModel.load(str(folder / "mist_starmodel_single.npz"))
    assert single.N == 1 and set(single.kwargs) == {"J", "H", "K", "Teff", "parallax"}
    em = ia.starfit(str(folder), multiplicities=["single"], use_emcee=True, overwrite=True, nwalkers=24, nburn=10,
                    niter=5, seed=1)
    assert len(em.samples) == 120 and em.use_emcee
    with pytest.raises(ValueError, match="multiplicity"):
        ia.starfit(str(folder), multiplicities=["quadruple"])


def test_readme_quick_tour_runs(tmp_path, monkeypatch):
    """The README's quick tour, executed verbatim (stars/ = a copy of the ini fixtures)."""
    import os
    import re
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "README.md")).read()
    code = re.search(r"## Quick tour.*?```python\n(.*?)```", text, re.S).group(1)
    shutil.copytree(INI_DIR, str(tmp_path / "stars"))
    monkeypatch.chdir(tmp_path)
    ns = {}
    exec(compile(code, "README.md quick tour", "exec"), ns)
    assert len(ns["res"]) == 10_000 and np.isfinite(ns["mod"].evidence[0])
    assert ns["again"].kwargs == ns["mod"].kwargs and ns["lp"].is_cuda and ns["tree"].n_params == 11
    assert os.path.exists(str(tmp_path / "stars" / "flat" / "mist_starmodel_binary.npz"))


def test_catalog_write_ini_then_starfit_per_folder(tmp_path):
    """reference StarCatalog.write_ini (catalog.py:141-158) + scripts/batch_starfit: a catalog written as one ini
    folder per star, each folder fitted by starfit(); the per-folder models see the catalog's measurements."""
    import os
    ic = ia.get_ichrone("mist", bands=["G", "BP", "RP"], tracks=True)
    cat, truth = ia.synthetic_catalog(ic, 4, bands=["G", "BP", "RP"], seed=3)
    dirs = cat.write_ini(ic, root=str(tmp_path), nest_directories=False)
    assert len(dirs) == 4 and all(os.path.exists(os.path.join(d, "star.ini")) for d in dirs)
    for i, d in enumerate(dirs):
        want = cat.model(i, ic)
        mod = ia.starfit(d, multiplicities=["single"], ichrone=ic, n_live_points=60, max_iter=100, seed=i)
        assert mod.kwargs.keys() == want.kwargs.keys() and mod.name == os.path.basename(d)
        for k in want.kwargs:
            assert np.allclose(mod.kwargs[k], want.kwargs[k], rtol=0, atol=0)
        p = mod.samples[list(mod.param_names)].values[:16]
        assert np.array_equal(mod.lnpost(p), want.lnpost(p))
    # the list-file form, split like scripts/batch_starfit: line NR (1-based) -> task NR % P
    listfile = tmp_path / "stars.list"
    listfile.write_text("\n".join(os.path.basename(d) for d in dirs) + "\n# comment\nnot_a_folder\n")
    r0 = ia.batch_starfit(str(listfile), rank=0, world=2, multiplicities=["single"], ichrone=ic)
    r1 = ia.batch_starfit(str(listfile), rank=1, world=2, multiplicities=["single"], ichrone=ic)
    assert sorted(map(os.path.basename, r0)) == sorted([os.path.basename(dirs[1]), os.path.basename(dirs[3])])
    assert sorted(map(os.path.basename, r1)) == sorted([os.path.basename(dirs[0]), os.path.basename(dirs[2]), "not_a_folder"])
    assert isinstance(r1[os.path.join(str(tmp_path), "not_a_folder")], Exception)          # logged, batch carried on
    assert all(m.name == os.path.basename(f) for f, m in r0.items())
    again = cat.write_ini(ic, root=str(tmp_path), nest_directories=False)      # clobbers the folders (and their fits)
    assert again == dirs and not os.path.exists(os.path.join(dirs[0], "mist_starmodel_single.npz"))


def test_batch_starfit_device_batched_route(tmp_path):
    """batch_starfit(batched=True): the ini folders of a rank are sampled together (S stars x W walkers per
    launch) and every folder still gets its own stored model; the stored lnprob of every sample equals the
    per-folder model's own lnpost at that sample (catalog kernel == single-model kernel)."""
    import os
    ic = ia.get_ichrone("mist", bands=["G", "BP", "RP"], tracks=True)
    cat, truth = ia.synthetic_catalog(ic, 24, bands=["G", "BP", "RP"], seed=5)
    dirs = cat.write_ini(ic, root=str(tmp_path), nest_directories=False)
    # one star loses a band, one its parallax, one gets a keyword only the per-folder route knows
    from isochrones_amd import ini
    sc, _ = ini.read_ini(os.path.join(dirs[3], "star.ini"))
    sc.pop("BP"); ini.write_ini(os.path.join(dirs[3], "star.ini"), {k: ini.parse_value(v) for k, v in sc.items()})
    sc, _ = ini.read_ini(os.path.join(dirs[4], "star.ini"))
    sc.pop("parallax"); ini.write_ini(os.path.join(dirs[4], "star.ini"), {k: ini.parse_value(v) for k, v in sc.items()})
    with open(os.path.join(dirs[5], "star.ini"), "a") as f:
        f.write("maxAV = 0.5\n")
    res = ia.batch_starfit(dirs + [str(tmp_path / "nowhere")], batched=True, ichrone=ic, nwalkers=32, nburn=60,
                           niter=20, seed=9)
    assert isinstance(res[str(tmp_path / "nowhere")], Exception)
    assert len(res) == 25
    for i, d in enumerate(dirs):
        mod = res[d]
        assert isinstance(mod, ia.BasicStarModel), (d, mod)
        assert os.path.exists(os.path.join(d, "mist_starmodel_single.npz"))
        s = mod.samples
        assert len(s) == 32 * 20 and np.isfinite(s["lnprob"]).all() and "Teff" in s.columns and "G_mag" in s.columns
        p = s[list(mod.param_names)].values
        assert np.allclose(mod.lnpost(p), s["lnprob"].values, rtol=1e-12, atol=1e-12), d
    assert "BP" not in res[dirs[3]].kwargs and "parallax" not in res[dirs[4]].kwargs
    # the derived samples assembled from the slab-wide call equal the ones the model computes for itself
    fresh = ia.BasicStarModel.from_ini(ic, dirs[0])
    fresh._samples = res[dirs[0]].samples
    a, b = res[dirs[0]].derived_samples, fresh.derived_samples
    assert list(a.columns) == list(b.columns) and np.array_equal(a.values, b.values, equal_nan=True)
    assert res[dirs[5]].bounds("AV") == (0, 0.5)                       # fitted on its own, with its maxAV
    back = ia.SingleStarModel.load_hdf(os.path.join(dirs[0], "mist_starmodel_single.npz"), ic=ic)
    assert np.array_equal(back.samples.values, res[dirs[0]].samples.values)
    again = ia.batch_starfit(dirs, batched=True, ichrone=ic)          # everything exists: loaded, nothing refitted
    assert np.array_equal(again[dirs[1]].samples.values, res[dirs[1]].samples.values)
    # the sampled posteriors sit where the catalog's truth is (distance within 25 %)
    good = [abs(np.median(res[d].samples["distance"]) / truth["distance"].iloc[i] - 1) < 0.25 for i, d in enumerate(dirs)]
    assert np.mean(good) > 0.8
