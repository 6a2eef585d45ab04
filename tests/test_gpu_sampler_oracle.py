"""Oracle-anchored tests of the device-resident sampler at BASELINE.json's sizes.

cfg 4: 256 walkers x 5000 steps on the full-size MIST-shaped tables.  Every one of the 1.28 x 10^6 stretch
moves of the stored chain is replayed on the host (tests/_replay.py: same Philox counters, proposal rebuilt
from the stored previous state) and evaluated with the CPU oracle; the GPU's accept/reject decisions, stored
positions and stored lnprob values must be the oracle's.
cfg 5: a 10^4-star catalog at 32 walkers x 250 steps: oracle lnpost on a 2000-row subsample of the stored
chains, replay of the moves of a few hundred stars, device quantile summaries vs numpy, truth recovery.

Reference behaviour being checked: emcee.EnsembleSampler(nwalkers, npars, self.lnpost).run_mcmc
(isochrones/starmodel.py:886-972), one star per process in scripts/batch_starfit:60-62.
"""
import numpy as np
import pytest

import isochrones_amd as ia
from tests import _fixtures as fx
from tests import _replay

pytestmark = pytest.mark.gpu


def _oracle_fn(ic, descs):
    """lnpost_fn(block, pars) for _replay.replay: block b is evaluated with descs[b] by the CPU oracle."""
    oic = fx.make_oracle_ic(ic)

    def fn(blk, pars):
        out = np.empty(pars.shape[0])
        if len(descs) == 1:
            out[:] = oic.lnpost(descs[0], np.ascontiguousarray(pars.T), nthreads=16, parts=False)
            return out
        order = np.argsort(blk, kind="stable")
        bounds = np.searchsorted(blk[order], np.arange(len(descs) + 1))
        for b in range(len(descs)):
            sel = order[bounds[b]:bounds[b + 1]]
            if sel.size:
                out[sel] = oic.lnpost(descs[b], np.ascontiguousarray(pars[sel].T), nthreads=4, parts=False)
        return out
    return fn


def _cfg4_start(mod, nwalkers, seed=1):
    truth = np.array([1.0, 355.0, 0.0, 100.0, 0.1])
    rng = np.random.default_rng(seed)
    p0 = truth + np.array([0.01, 2.0, 0.02, 1.0, 0.02]) * rng.standard_normal((nwalkers, 5))
    p0[:, 4] = np.abs(p0[:, 4])
    return p0


@pytest.mark.parametrize("mode,nsteps", [("auto", 5000), ("stepwise", 700)])
def test_cfg4_every_move_of_the_fit_against_the_oracle(mode, nsteps, monkeypatch):
    """BASELINE configs[3]: 256 walkers x 5000 steps, full-size tables (persistent kernel = what `auto` picks for
    one ensemble; the launch-per-half-step kernel on a shorter run)."""
    import bench
    from isochrones_amd.sampler import FusedEnsembleSampler
    monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", mode)
    W, seed = 256, 2
    ic, mod = bench.build_model()
    assert ic.model_grid.interp.grid.shape == (15, 196, 1710, 18)
    p0 = _cfg4_start(mod, W)
    fn = _oracle_fn(ic, [mod.model_desc()])
    lnp0 = fn(np.zeros(W, dtype=int), p0)
    assert np.isfinite(lnp0).all()
    fs = FusedEnsembleSampler(mod, W, seed=seed)
    pos, lnp = fs.run_mcmc(p0, nsteps, lnprob0=lnp0, store=True)
    chain = fs.chain_steps.cpu().numpy()            # [T, W, 5]
    clnp = fs._lnprob.cpu().numpy()
    assert chain.shape == (nsteps, W, 5)
    st = _replay.replay(p0, lnp0, chain, clnp, W, 2.0, seed, 0, fn)
    assert st["moves"] == W * nsteps
    assert 0.2 < st["accepted"] / st["moves"] < 0.7
    assert st["near_ties"] <= 2                      # decisions within 1e-9 of the threshold
    assert st["max_lnp_rel"] < 1e-9
    # the returned state is the last stored step, and the acceptance counters count the accepted moves
    assert np.array_equal(pos.cpu().numpy(), chain[-1]) and np.array_equal(lnp.cpu().numpy(), clnp[-1])
    prev = np.concatenate([p0[None], chain[:-1]])
    prev_l = np.concatenate([lnp0[None], clnp[:-1]])
    moved = np.any(chain != prev, axis=2) | (clnp != prev_l)
    assert np.array_equal(fs.accepted.cpu().numpy(), moved.sum(axis=0).astype(np.int32))
    # the chain explores: every walker moved, and the posterior median is near the truth of the synthetic star
    assert (moved.sum(axis=0) > 0.1 * nsteps).all()
    # a second run_mcmc continues the counter stream (step0 = nsteps)
    pos2, lnp2 = fs.run_mcmc(pos, 40, lnprob0=lnp, store=True)
    c2, l2 = fs.chain_steps.cpu().numpy()[nsteps:], fs._lnprob.cpu().numpy()[nsteps:]
    st2 = _replay.replay(chain[-1], clnp[-1], c2, l2, W, 2.0, seed, nsteps, fn)
    assert st2["moves"] == 40 * W


def test_cfg3_binary_sampler_moves_against_the_oracle():
    """The NS = 2 / 6-band instantiation of the sampler kernel (BASELINE configs[2]'s model), full-size isochrone
    table, 128 walkers x 1500 steps."""
    from isochrones_amd.sampler import FusedEnsembleSampler
    bands = ("J", "H", "K", "BP", "RP", "G")
    ic = ia.synthetic_isochrone(bands=bands)
    truth = np.array([350.0, 300.0, 9.7, 0.0, 500.0, 0.2])
    a = ic.interp_mag([truth[0], truth[2], truth[3], truth[4], truth[5]], bands)[3]
    b = ic.interp_mag([truth[1], truth[2], truth[3], truth[4], truth[5]], bands)[3]
    tot = -2.5 * np.log10(10 ** (-0.4 * a) + 10 ** (-0.4 * b))
    unc = (0.02, 0.02, 0.02, 0.002, 0.002, 0.001)
    mod = ia.BinaryStarModel(ic, parallax=(2.0, 0.05), **{bd: (float(m), u) for bd, m, u in zip(bands, tot, unc)})
    W, T, seed = 128, 1500, 11
    rng = np.random.default_rng(5)
    p0 = truth + np.array([0.5, 0.5, 0.005, 0.005, 2.0, 0.005]) * rng.standard_normal((W, 6))
    p0[:, :2] = -np.sort(-p0[:, :2], axis=1)
    fn = _oracle_fn(ic, [mod.model_desc()])
    lnp0 = fn(np.zeros(W, dtype=int), p0)
    assert np.isfinite(lnp0).all()
    fs = FusedEnsembleSampler(mod, W, seed=seed)
    fs.run_mcmc(p0, T, lnprob0=lnp0, store=True)
    # lnlike reaches 1e5..1e7 here (sigma = 0.001 mag): 1e-9 relative, plus the absolute floor that a 1e-14 mag
    # rounding difference times (residual / sigma^2 ~ 1e6) produces
    st = _replay.replay(p0, lnp0, fs.chain_steps.cpu().numpy(), fs._lnprob.cpu().numpy(), W, 2.0, seed, 0, fn, lnp_atol=1e-7,
                        margin=1e-8)
    assert st["moves"] == W * T and st["accepted"] > 0.02 * st["moves"]
    assert st["near_ties"] <= 2


@pytest.mark.parametrize("kind", ["track", "iso"])
def test_asteroseismic_model_fits_on_the_fused_sampler_and_matches_the_oracle(kind, monkeypatch):
    """nu_max / delta_nu terms (reference starmodel.py:1603-1612, delta_nu with sigma = its value): the ASTERO
    instantiations of both sampler kernels, replayed move by move; fit_mcmc stays on the device path."""
    from isochrones_amd.sampler import FusedEnsembleSampler
    if kind == "track":
        ic = ia.synthetic_track(bands=("V", "K"))
        truth = np.array([1.0, 355.0, 0.0, 100.0, 0.1])
        width = np.array([0.01, 2.0, 0.02, 1.0, 0.02])
    else:
        ic = ia.synthetic_isochrone(bands=("V", "K"))
        truth = np.array([355.0, 9.6, 0.0, 100.0, 0.1])
        width = np.array([2.0, 0.01, 0.02, 1.0, 0.02])
    numax, dnu = (float(v) for v in ic.interp_value(truth, ["nu_max", "delta_nu"]))
    T, g, f, mags = ic.interp_mag(truth, ["V", "K"])
    mod = ia.SingleStarModel(ic, Teff=(float(T), 100), V=(float(mags[0]), 0.05), K=(float(mags[1]), 0.03),
                             nu_max=(numax, 0.03 * numax), delta_nu=(dnu, 0.02 * dnu))
    d = mod.model_desc()
    assert d.has_numax == 1 and d.has_dnu == 1
    W, seed = 64, 5
    rng = np.random.default_rng(2)
    p0 = truth + width * rng.standard_normal((W, 5))
    p0[:, 4] = np.abs(p0[:, 4])
    fn = _oracle_fn(ic, [d])
    lnp0 = fn(np.zeros(W, dtype=int), p0)
    assert np.isfinite(lnp0).all()
    # the asteroseismic terms matter for this posterior
    plain = ia.SingleStarModel(ic, Teff=(float(T), 100), V=(float(mags[0]), 0.05), K=(float(mags[1]), 0.03))
    assert np.max(np.abs(fn(np.zeros(W, dtype=int), p0) - _oracle_fn(ic, [plain.model_desc()])(np.zeros(W, dtype=int), p0))) > 1.0
    for mode, T_steps in (("persistent", 1200), ("stepwise", 300)):
        monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", mode)
        fs = FusedEnsembleSampler(mod, W, seed=seed)
        fs.run_mcmc(p0, T_steps, lnprob0=lnp0, store=True)
        st = _replay.replay(p0, lnp0, fs.chain_steps.cpu().numpy(), fs._lnprob.cpu().numpy(), W, 2.0, seed, 0, fn)
        assert st["moves"] == W * T_steps and st["accepted"] > 0.1 * st["moves"] and st["near_ties"] <= 2
        fs.close()
    monkeypatch.delenv("ISOCHRONES_AMD_SAMPLER")
    sampler = mod.fit_mcmc(nwalkers=64, nburn=200, niter=100, p0=truth, seed=3, fused=True)     # fused=True: no fallback
    assert isinstance(sampler, FusedEnsembleSampler) and sampler.chain.shape == (64, 100, 5)
    assert isinstance(mod.fit_mcmc(nwalkers=64, nburn=20, niter=20, p0=truth, seed=3), FusedEnsembleSampler)   # and by default
    lp = sampler.flatlnprobability.cpu().numpy()
    want = fn(np.zeros(lp.size, dtype=int), sampler.flatchain.cpu().numpy())
    fx.assert_close(lp, want, 1e-9, atol=1e-10, what="astero chain lnprob")


def _catalog(n_stars, seed=7):
    bands = ["G", "BP", "RP"]
    ic = ia.synthetic_track(bands=bands)            # full-size tables, as bench.py's catalog leg
    cat, truth = ia.synthetic_catalog(ic, n_stars, bands=bands, seed=seed, mag_unc=0.01)
    return ic, cat, truth


def test_cfg5_catalog_of_1e4_stars_against_the_oracle():
    """BASELINE configs[4] on one GPU: 10^4 stars, 32 walkers x (150 + 100) steps."""
    import torch
    from isochrones_amd.catalog import fit_stars_gpu, result_columns
    n_stars, W, nburn, niter = 10_000, 32, 150, 100
    ic, cat, truth = _catalog(n_stars)
    rows, chain, lnps = fit_stars_gpu(cat, ic, np.arange(n_stars), nwalkers=W, nburn=nburn, niter=niter, seed=11,
                                      return_chains=True)
    assert rows.shape == (n_stars, 3 * 5 + 3) and chain.shape == (n_stars, W, niter, 5)
    cols = result_columns(ic.param_names)
    ok = rows[:, cols.index("ok")] == 1
    assert ok.mean() > 0.995
    # (1) stored lnprob of 2000 random chain rows vs the oracle evaluated with that star's own model
    rng = np.random.default_rng(0)
    stars = rng.choice(np.flatnonzero(ok), 250, replace=False)
    oic = fx.make_oracle_ic(ic)
    worst = 0.0
    for s in stars:
        w = torch.as_tensor(rng.integers(0, W, 8), device=chain.device)
        t = torch.as_tensor(rng.integers(0, niter, 8), device=chain.device)
        p = chain[int(s), w, t].cpu().numpy()
        got = lnps[int(s), w, t].cpu().numpy()
        want = oic.lnpost(cat.model(int(s), ic).model_desc(), np.ascontiguousarray(p.T), parts=False)
        fx.assert_close(got, want, 1e-9, atol=1e-10, what="star %d chain lnprob" % s)
        worst = max(worst, float(np.max(np.abs(got - want) / np.maximum(1, np.abs(want)))))
    assert worst < 1e-9
    assert bool(torch.isfinite(lnps[torch.as_tensor(np.flatnonzero(ok), device=lnps.device)]).all())
    # (2) the device summaries are numpy's percentiles of the stored chains
    for s in stars[:40]:
        flat = chain[int(s)].reshape(-1, 5).cpu().numpy()
        q = np.percentile(flat, [50, 16, 84], axis=0).T.ravel()
        assert np.array_equal(rows[s, :15], q)
        assert rows[s, 15] == float(lnps[int(s)].max())
    # (3) truth recovery over the whole catalog (parallax 2 %, photometry 0.01 mag)
    d = rows[ok, cols.index("distance_median")]
    dt = truth["distance"].values[ok]
    assert np.median(np.abs(d - dt) / dt) < 0.03
    lo, hi = rows[ok, cols.index("distance_p16")], rows[ok, cols.index("distance_p84")]
    cover = np.mean((dt > lo) & (dt < hi))
    assert 0.45 < cover < 0.9                       # 68 % interval
    m, mt = rows[ok, cols.index("mass_median")], truth["mass"].values[ok]
    assert np.median(np.abs(m - mt) / mt) < 0.15
    acc = rows[ok, cols.index("acceptance")]
    assert 0.1 < np.median(acc) < 0.7


def test_cfg5_catalog_sampler_moves_against_the_oracle(monkeypatch):
    """The catalog (MULTI) instantiation at 10^4 ensembles in one run: all moves of 300 of the stars over 60
    steps are replayed against the oracle with each star's own model (both kernel forms)."""
    import torch
    from isochrones_amd.catalog import CatalogPosterior, initial_positions
    from isochrones_amd.sampler import FusedEnsembleSampler
    n_stars, W, T = 10_000, 32, 60
    ic, cat, _ = _catalog(n_stars, seed=8)
    post = CatalogPosterior.from_catalog(cat, ic)
    pos, lnp, failed = initial_positions(post, W, rng_seed=3)
    good = np.flatnonzero(~failed.cpu().numpy())
    if bool(failed.any()):                          # keep the batch rectangular, as fit_stars_gpu does
        pos[failed] = pos[int(good[0])]
        lnp[failed] = 0.0
    pick = np.sort(np.random.default_rng(1).choice(good, 300, replace=False))
    descs = [cat.model(int(s), ic).model_desc() for s in pick]
    fn = _oracle_fn(ic, descs)
    sel = torch.as_tensor(pick, device=pos.device)
    p_sel = pos[sel].reshape(-1, 5).cpu().numpy()
    l_sel = lnp[sel].reshape(-1).cpu().numpy()
    # the start points' lnpost (iso_catalog_lnpost, columns-built constant blocks) is the oracle's as well
    fx.assert_close(l_sel, fn(np.repeat(np.arange(300), W), p_sel), 1e-9, atol=1e-10, what="catalog start lnpost")
    for mode in ("stepwise", "persistent-dense"):
        monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", mode)
        fs = FusedEnsembleSampler(post, W, seed=21)
        fs.run_mcmc(pos, T, lnprob0=lnp, store=True)
        ch = fs.chain_steps.reshape(T, n_stars, W, 5)[:, sel].reshape(T, -1, 5).cpu().numpy()
        cl = fs._lnprob.view(T, n_stars, W)[:, sel].reshape(T, -1).cpu().numpy()
        st = _replay.replay(p_sel, l_sel, ch, cl, W, 2.0, 21, 0, fn, star_of_block=pick, lnp_atol=1e-10)
        assert st["moves"] == 300 * W * T and st["accepted"] > 0.1 * st["moves"]
        assert st["near_ties"] <= 2
        fs.close()
    post.close()


def test_independent_ensembles_of_one_model_in_one_launch():
    """iso_sampler_create_model_ensembles: E ensembles of the cfg-2 star advanced by the same launches.  Every move of
    every ensemble against the oracle (row = ensemble * W + walker keys the random numbers), ensemble 0 bit-identical to
    the single-ensemble sampler, differently started ensembles agree on the posterior (R-hat)."""
    import torch
    import bench
    from isochrones_amd.sampler import FusedEnsembleSampler
    ic, mod = bench.build_model()
    E, W, T, seed = 6, 64, 400, 5
    p0 = np.stack([_cfg4_start(mod, W, seed=10 + e) for e in range(E)])              # [E, W, 5]
    fn = _oracle_fn(ic, [mod.model_desc()])
    lnp0 = fn(np.zeros(E * W, dtype=int), p0.reshape(E * W, 5)).reshape(E, W)
    assert np.isfinite(lnp0).all()
    fs = FusedEnsembleSampler(mod, W, seed=seed, n_ensembles=E)
    pos, lnp = fs.run_mcmc(p0, T, lnprob0=lnp0, store=True)
    assert tuple(pos.shape) == (E, W, 5) and tuple(fs.chain.shape) == (E, W, T, 5) and tuple(fs.acceptance_fraction.shape) == (E, W)
    chain = fs.chain_steps.cpu().numpy()            # [T, E*W, 5]
    clnp = fs._lnprob.cpu().numpy()
    st = _replay.replay(p0.reshape(E * W, 5), lnp0.reshape(-1), chain, clnp, W, 2.0, seed, 0,
                        lambda blk, pars: fn(np.zeros(len(blk), dtype=int), pars))
    assert st["moves"] == E * W * T and st["near_ties"] <= 2 and st["max_lnp_rel"] < 1e-9
    # ensemble 0 = the plain single-ensemble sampler with the same seed and start
    one = FusedEnsembleSampler(mod, W, seed=seed)
    one.run_mcmc(p0[0], T, lnprob0=lnp0[0], store=True)
    assert torch.equal(one.chain, fs.chain[0]) and torch.equal(one.lnprobability, fs.lnprobability[0])
    # the ensembles sample the same posterior
    rhat = fs.gelman_rubin().cpu().numpy()
    assert rhat.shape == (5,) and np.all(rhat < 1.2), rhat
    q = fs.quantiles((0.5,)).cpu().numpy()           # [E, 5, 1]
    assert q.shape == (E, 5, 1)


@pytest.mark.parametrize("n_stars,nb", [(3, 9), (3, 11), (2, 11), (1, 11), (3, 6), (2, 9)])
def test_single_model_sampler_on_random_models_of_the_largest_shapes(n_stars, nb, monkeypatch):
    """Random isochrone models (the soak's generator: random observables, priors, prior keywords, table axes) of the
    shapes whose single-model kernels carry the most state, every stored move replayed against the oracle.
    Regression guard: a reordering of the model block's fields once made exactly the (3 stars, 9 bands) instantiation
    disagree with the oracle on most random models while the fixed-model tests and every other shape stayed clean -
    only the randomised soak (tests/soak/soak_sampler.py) saw it."""
    from isochrones_amd._cabi import IsoError
    from isochrones_amd.sampler import FusedEnsembleSampler
    from tests.soak import soak, soak_sampler
    monkeypatch.setenv("SOAK_KIND", "iso")
    monkeypatch.setenv("SOAK_NSTARS", str(n_stars))
    monkeypatch.setenv("SOAK_NB", str(nb))
    monkeypatch.setenv("ISOCHRONES_AMD_SAMPLER", "auto")
    rng = np.random.default_rng(1000 * n_stars + nb)
    done = 0
    for _ in range(60):
        cfg, ic, mod, axes, lo, hi = soak.build(rng)
        W = int(rng.choice([4, 30, 100, 256]))
        a = float(rng.choice([1.3, 2.0, 3.0]))
        sseed = int(rng.integers(0, 2 ** 40))
        try:
            fs = FusedEnsembleSampler(mod, W, a=a, seed=sseed)
        except IsoError:                 # a model the fused kernels do not take
            ic.release()
            continue
        p0 = soak_sampler.start_points(rng, mod, lo, hi, W, True)
        if p0 is None:
            ic.release()
            continue
        oic = fx.make_oracle_ic(ic)
        desc = mod.model_desc()

        def fn(blk, pars, oic=oic, desc=desc):
            return oic.lnpost(desc, np.ascontiguousarray(pars.T), nthreads=8, parts=False)
        lnp0 = fn(None, p0)
        assert np.isfinite(lnp0).all(), cfg
        fs.run_mcmc(p0, 12, lnprob0=lnp0, store=True)
        st = _replay.replay(p0, lnp0, fs.chain_steps.cpu().numpy(), fs._lnprob.cpu().numpy(), W, a, sseed, 0, fn,
                            lnp_atol=1e-7, margin=1e-8)
        assert st["near_ties"] <= 3, (st, cfg)
        fs.close()
        ic.release()
        done += 1
        if done >= 8:
            break
    assert done >= 4


@pytest.mark.parametrize("n_sys,n_stars,stdp", [(1, 800, 1), (2, 800, 0), (1, 1250, 0)])
def test_reference_shape_catalog_replayed_against_the_oracle(n_sys, n_stars, stdp):
    """The catalog the reference's `starfit` runs (starfit.py:86: MIST_Isochrone parametrisation; fit_mcmc's defaults
    nwalkers=300, starmodel.py:889-893) through `fit_stars_gpu`'s OWN route, with more stars than the uncapped kernel keeps
    resident (2 workgroups per CU), so that iso_sampler_run takes what it takes for the benchmark's reference-shape leg - the
    register-capped persistent kernel with workgroups of three waves (150 moves per half-step packed 64 + 64 + 22), for 800
    single stars (and for 10^4) with the default prior families compiled in, for the 1 250 stars one GPU of eight gets with
    the priors read at run time (the form that keeps that launch on the chip in one round).  Every move of 20 of the stars
    over the sampling run is rebuilt on the host and evaluated by the oracle with that star's own model (tests/_replay.py) -
    HIP against the oracle, not HIP against HIP; the same for a catalog of binaries (N = 2)."""
    import torch
    from isochrones_amd import _cabi
    from isochrones_amd.catalog import fit_stars_gpu
    W, nburn, niter = 300, 20, 30
    bands = ["G", "BP", "RP"]
    ic = ia.synthetic_isochrone(bands=bands)
    cat, _ = ia.synthetic_catalog(ic, n_stars, bands=bands, seed=17 + n_sys, mag_unc=0.01)
    rec = {}
    _cabi.trace_kernels(True)
    try:
        rows, chain, lnps = fit_stars_gpu(cat, ic, np.arange(n_stars), N=n_sys, nwalkers=W, nburn=nburn, niter=niter, seed=5,
                                          return_chains=True, replay_record=rec)
        names = _cabi.traced_kernels()
        plan = _cabi.last_sampler_plan()
    finally:
        _cabi.trace_kernels(False)
    D = n_sys + 4
    assert chain.shape == (n_stars, W, niter, D)
    # the launch the reference-shape leg of bench.py measures: register-capped persistent form, three-wave workgroups
    stretch = [k for k in names if k.startswith("k_stretch")]
    # (300 walkers = one ensemble per workgroup: single stars read their star's block through scalar loads - DENSE + UNI)
    want = ("k_stretch_persist<1, %d, 3, true, false, %s, %s>" % (n_sys, "true" if n_sys == 1 else "false", "true" if stdp else "false"))
    assert stretch == [want], stretch
    assert plan["persistent"] == 1 and plan["dense"] == 1 and plan["threads"] == 192, plan
    if n_sys == 1:          # (systems have the run-time-prior form only: the kernel's name says which one ran)
        assert plan["dense_stdp"] == stdp, plan
    good = np.flatnonzero(~rec["failed"].cpu().numpy())
    assert good.size >= 0.9 * n_stars
    pick = np.sort(np.random.default_rng(3).choice(good, 20, replace=False))
    descs = [cat.model(int(s), ic, N=n_sys).model_desc() for s in pick]
    fn = _oracle_fn(ic, descs)
    sel = torch.as_tensor(pick, device=chain.device)
    # start points (k_catalog_start at 300 walkers) and the ensembles burn-in left: the oracle's lnpost
    p_start = rec["start_pos"][sel].reshape(-1, D).cpu().numpy()
    fx.assert_close(rec["start_lnp"][sel].reshape(-1).cpu().numpy(), fn(np.repeat(np.arange(20), W), p_start), 1e-9, atol=1e-10,
                    what="start points")
    p0 = rec["pos"][sel].reshape(-1, D).cpu().numpy()
    l0 = rec["lnp"][sel].reshape(-1).cpu().numpy()
    fx.assert_close(l0, fn(np.repeat(np.arange(20), W), p0), 1e-9, atol=1e-10, what="state after burn-in")
    ch = chain[sel].permute(2, 0, 1, 3).reshape(niter, 20 * W, D).cpu().numpy()         # [T, B * W, D]
    cl = lnps[sel].permute(2, 0, 1).reshape(niter, 20 * W).cpu().numpy()
    st = _replay.replay(p0, l0, ch, cl, W, 2.0, rec["seed"], rec["step0"], fn, star_of_block=pick, lnp_atol=1e-9)
    assert st["moves"] == 20 * W * niter and st["accepted"] > 0.05 * st["moves"] and st["near_ties"] <= 2, st
    ic.release()
