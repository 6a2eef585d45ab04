"""Every kernel the build compiled, ticked off against the oracle.

libiso_hip.so is a table of template instantiations - (family, parametrisation, stars, bands, flags) - and the host picks
one per call.  Round 3 showed what an untested entry can hide (one sampler instantiation made wrong accept / reject
decisions while pytest was green).  This file enumerates the table from the build's own record
(isochrones_amd/csrc/libiso_hip.resources.json: one entry per kernel hipcc compiled), steers the library to every entry
through the public API, asks the library which kernel it actually launched (iso_debug_trace_kernels /
iso_debug_kernels) and checks the result against the CPU oracle:

  * batch kernels (k_lnpost_fast, k_lnpost_wide, k_lnpost, tree kernels): 4 096 seeded rows - wide over the bounds, a
    cluster near a solution, special values - lnpost / lnprior / lnlike vs the oracle, exact NaN / -inf pattern, 1e-9;
  * sampler kernels (k_stretch_half, k_stretch_persist, k_stretch_pair, and the any-model family k_stretch_tree /
    k_stretch_isotrack / k_stretch_wide): a short run whose every move is replayed on the host with the kernel's Philox
    stream and the oracle's lnpost (tests/_replay.py);
  * the interpolation / summary / set-up kernels: their own oracle or numpy equivalents.

The last test requires that the kernels seen by the tracer are exactly the kernels the build compiled: an entry nobody
can reach is dead weight to prune, an entry nobody tested is a hole.

Reference semantics: likelihood.py:40-147 (any N in {1,2,3} x any band list gives star_lnlike's number),
starmodel.py:538-542, 951-969, 1563-1635; observation.py:1181-1234; interp.py:252-392."""
import os
import re

import numpy as np
import pytest

import isochrones_amd as ia
from isochrones_amd import _cabi
from isochrones_amd.csrc import build as B
from tests import _fixtures as fx
from tests import _replay

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-10

#: kernel name -> test id that launched it (filled as the tests run; compared with the build's table at the end)
SEEN = {}
RAN = set()
ALL_BANDS = tuple(list(ia.grids.KNOWN_BANDS) + ["X%02d" % j for j in range(16)])


def table():
    return B.resource_table()


class traced:
    """with traced(test_id) as t: ... ; t.names = kernels launched inside."""

    def __init__(self, who):
        self.who = who

    def __enter__(self):
        _cabi.trace_kernels(True)
        return self

    def __exit__(self, *exc):
        self.names = _cabi.traced_kernels()
        _cabi.trace_kernels(False)
        for n in self.names:
            SEEN.setdefault(n, self.who)
        return False


class env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return False


# ---- small tables, models and samples -------------------------------------------------------------------------------
FEHS = np.array([-2.0, -1.0, -0.5, -0.25, 0.0, 0.25, 0.5])


def make_ic(kind, bands):
    bands = tuple(bands) or ("G",)                  # the BC table needs a column even if no band is observed
    if kind == "track":
        masses = ia.grids.mist_masses()[20:150:3]
        eeps = np.arange(200.0, 700.0)
        ic = ia.synthetic_track(bands=bands, fehs=FEHS, masses=masses, eeps=eeps, eep_bounds=(eeps[0], eeps[-1]),
                                limits=dict(mass=(masses[0], masses[-1]), feh=(-2.0, 0.5), age=(5, 10.13)))
        lo = np.array([masses[0], eeps[0], -2.0, 5.0, 0.0]); hi = np.array([masses[-1], eeps[-1], 0.5, 2000.0, 1.0])
        return ic, lo, hi
    ages = ia.grids.mist_log_ages()[40::3]
    eeps = np.arange(150.0, 700.0)
    ic = ia.synthetic_isochrone(bands=bands, ages=ages, fehs=FEHS, eeps=eeps, eep_bounds=(eeps[0], eeps[-1]),
                                limits=dict(age=(ages[0], ages[-1]), feh=(-2.0, 0.5)))
    return ic, np.array([eeps[0], ages[0], -2.0, 5.0, 0.0]), np.array([eeps[-1], ages[-1], 0.5, 2000.0, 1.0])


def truth_of(kind, ns):
    if kind == "track":
        return np.array([1.0, 355.0, 0.0, 300.0, 0.1])
    return np.array([380.0, 330.0, 300.0][:ns] + [9.6, -0.1, 300.0, 0.1])


def bounds_of(kind, ns, lo, hi):
    if kind == "track":
        return lo, hi
    return np.concatenate([[lo[0]] * ns, lo[1:]]), np.concatenate([[hi[0]] * ns, hi[1:]])


def prior_set(which, kind):
    """Two sets of NON-DEFAULT prior families, between them every family of ln_pdf's run-time switch in every slot
    (reference priors.py:235-381): the families the fused kernels read at run time when a model does not carry the
    reference's defaults.  Bounds wide enough for the test's samples."""
    P = ia.priors
    if which == "A":
        d = dict(mass=P.LogNormalPrior(0.0, 0.4), age=P.GaussianPrior(9.6, 0.3, bounds=(8.0, 10.13)), feh=P.FlatPrior((-1.5, 0.4)),
                 distance=P.GaussianPrior(300.0, 60.0, bounds=(1.0, 2000.0)), AV=P.PowerLawPrior(0.5, (0.0, 1.0)))
        orig = P.GaussianPrior(7.7, 0.8, bounds=(5.0, 10.13)) if kind == "track" else P.LogNormalPrior(-0.1, 0.5)      # (the synthetic tracks are young)
    else:
        d = dict(mass=P.PowerLawPrior(-2.35, (0.1, 10.0)), age=P.FlatLogPrior((8.0, 10.13)), feh=P.GaussianPrior(-0.2, 0.3, bounds=(-1.9, 0.5)),
                 distance=P.LogNormalPrior(5.7, 0.5), AV=P.GaussianPrior(0.2, 0.1, bounds=(0.0, 1.0)))
        orig = P.FlatLogPrior((5.0, 10.13)) if kind == "track" else P.PowerLawPrior(-2.35, (0.1, 10.0))
    direct = {k: v for k, v in d.items() if k != ("age" if kind == "track" else "mass")}      # the other one enters through the EEP term
    return direct, orig


def make_model(ic, kind, ns, bands, astero=False, custom_prior=False, spec=True, priors=None):
    truth = truth_of(kind, ns)
    prim = [truth[0]] + list(truth[ns:]) if kind == "iso" else list(truth)
    obs = {}
    if bands:
        mags = ic.interp_mag(prim, list(bands))[3]
        for j, b in enumerate(bands):
            obs[b] = (float(mags[j]) - 0.3 * (ns > 1), 0.02 + 0.01 * (j % 3))
    if spec:
        obs.update(Teff=(5700.0, 120.0), feh=(-0.1, 0.15))
    obs["parallax"] = (1000.0 / truth[-2], 0.05)
    if astero:
        nm, dn = ic.interp_value(prim[:3], ["nu_max", "delta_nu"])
        obs["nu_max"] = (float(nm), 0.05 * abs(float(nm)) + 1.0)
        obs["delta_nu"] = (float(dn), 0.05 * abs(float(dn)) + 0.1)
    mod = ia.BasicStarModel(ic, N=ns, **obs)
    if custom_prior:
        mod.set_prior(AV=ia.priors.FlatPrior((0.0, 0.8)))
    if priors:
        direct, orig = prior_set(priors, kind)
        mod.set_prior(**direct)
        mod._priors["eep"].orig_prior = orig
    return mod


def batch_rows(rng, kind, ns, lo, hi, n=4096):
    lo, hi = bounds_of(kind, ns, lo, hi)
    span = hi - lo
    x = rng.uniform(lo - 0.02 * span, hi + 0.02 * span, size=(n, lo.size))
    truth = truth_of(kind, ns)
    w = np.where(np.arange(lo.size) < (ns if kind == "iso" else 0), 8.0, 0.0) + np.array(
        ([0.05, 8.0, 0.05, 10.0, 0.03] if kind == "track" else [0.0] * ns + [0.05, 0.05, 10.0, 0.03]))
    x[: n // 2] = truth + w * rng.standard_normal((n // 2, lo.size))
    x[: n // 2, -1] = np.abs(x[: n // 2, -1])
    if kind == "iso" and ns > 1:
        x[: 3 * n // 4, :ns] = -np.sort(-x[: 3 * n // 4, :ns], axis=1)
    for j in range(lo.size):                       # bounds, specials
        x[n - 8 * (j + 1): n - 8 * (j + 1) + 6, j] = [lo[j], hi[j], np.nan, np.inf, -np.inf, np.nextafter(hi[j], np.inf)]
    return x


def check_batch(mod, oic, x, what, parts=True):
    w_post, w_prior, w_like = oic.lnpost(mod.model_desc(), np.ascontiguousarray(x.T), nthreads=8)
    assert np.isfinite(w_post).sum() > x.shape[0] // 20, what
    import torch
    fx.assert_close(mod.lnpost(torch.as_tensor(x, device="cuda")).cpu().numpy(), w_post, RTOL, atol=ATOL, what=what + " lnpost")
    if parts:
        fx.assert_close(mod.lnprior(x), w_prior, RTOL, atol=ATOL, what=what + " lnprior")
        fx.assert_close(mod.lnlike(x), w_like, RTOL, atol=ATOL, what=what + " lnlike")


def start_ball(rng, mod, kind, ns, W):
    truth = truth_of(kind, ns)
    w = np.array([0.01, 1.0, 0.01, 1.0, 0.01]) if kind == "track" else np.array([1.0] * ns + [0.01, 0.01, 1.0, 0.01])
    for _ in range(20):
        p = truth + w * rng.standard_normal((4 * W, truth.size))
        p[:, -1] = np.abs(p[:, -1])
        if kind == "iso" and ns > 1:
            p[:, :ns] = -np.sort(-p[:, :ns], axis=1)
        good = np.flatnonzero(np.isfinite(mod.lnpost(p)))
        if good.size >= W:
            return p[good[:W]]
    raise AssertionError("no start points")


def check_sampler(mod, oic, p0, W, steps, seed, what, n_ensembles=1, fn=None):
    from isochrones_amd.sampler import FusedEnsembleSampler
    if fn is None:
        desc = mod.model_desc()

        def fn(blk, pars):
            return oic.lnpost(desc, np.ascontiguousarray(pars.T), nthreads=8, parts=False)
    kw = dict(n_ensembles=n_ensembles) if n_ensembles > 1 else {}
    fs = FusedEnsembleSampler(mod, W, a=2.0, seed=seed, **kw)
    start = np.broadcast_to(p0, (n_ensembles,) + p0.shape).reshape(-1, p0.shape[1]).copy() if n_ensembles > 1 else p0
    lnp0 = fn(None, start)
    assert np.isfinite(lnp0).all(), what
    pin = start.reshape(n_ensembles, W, -1) if n_ensembles > 1 else start
    fs.run_mcmc(pin, steps, lnprob0=lnp0.reshape(n_ensembles, W) if n_ensembles > 1 else lnp0, store=True)
    chain = fs.chain_steps.cpu().numpy().reshape(steps, -1, p0.shape[1])
    st = _replay.replay(start, lnp0, chain, fs._lnprob.cpu().numpy().reshape(steps, -1), W, 2.0, seed, 0, fn, lnp_atol=1e-7, margin=1e-8)
    fs.close()
    assert st["moves"] == steps * W * n_ensembles and st["accepted"] > 0 and st["near_ties"] <= 2, (what, st)


def make_catalog(ic, kind, ns, bands, S, seed):
    from isochrones_amd.catalog import CatalogPosterior
    cat, _ = ia.synthetic_catalog(ic, S, bands=list(bands), seed=seed, mag_unc=0.02, with_parallax=True)
    return cat, CatalogPosterior.from_catalog(cat, ic, N=ns)


def check_catalog_batch(cat, post, ic, oic, ns, rng, what):
    import torch
    from isochrones_amd.catalog import initial_positions
    S = post.n_models
    pos, lnp, failed = initial_positions(post, 16, rng_seed=int(rng.integers(1 << 30)))
    assert not bool(failed.all()), what
    good = np.flatnonzero(~failed.cpu().numpy())
    D = post.n_params
    pars = pos[torch.as_tensor(good, device=pos.device)].reshape(-1, D)
    sid = torch.as_tensor(np.repeat(good, 16), dtype=torch.int32, device=pos.device)
    jitter = torch.as_tensor(rng.standard_normal(pars.shape) * np.array([2.0] * ns + [0.02, 0.02, 3.0, 0.01])[:D] if ic.kind != "track"
                             else rng.standard_normal(pars.shape) * np.array([0.02, 2.0, 0.02, 3.0, 0.01]), device=pos.device)
    pars = pars + jitter
    got = post.lnpost(pars.contiguous(), sid).cpu().numpy()
    p = pars.cpu().numpy()
    want = np.empty_like(got)
    for k in good:
        sel = np.flatnonzero(sid.cpu().numpy() == k)
        want[sel] = oic.lnpost(cat.model(int(k), ic, N=ns).model_desc(), np.ascontiguousarray(p[sel].T), nthreads=8, parts=False)
    # (the catalog kernels report what the batch kernel reports: NaN stays NaN)
    fx.assert_close(got, want, RTOL, atol=ATOL, what=what)
    return pos, lnp, good


def check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, steps, seed, what):
    import torch
    from isochrones_amd.sampler import FusedEnsembleSampler
    W = pos.shape[1]
    bad = np.setdiff1d(np.arange(post.n_models), good)
    if bad.size:
        pos[torch.as_tensor(bad, device=pos.device)] = pos[int(good[0])]
        lnp[torch.as_tensor(bad, device=pos.device)] = 0.0
    pick = good[: min(12, good.size)]
    descs = [cat.model(int(k), ic, N=ns).model_desc() for k in pick]

    def fn(blk, pars):
        out = np.empty(pars.shape[0])
        for b in range(len(descs)):
            sel = np.flatnonzero(blk == b)
            if sel.size:
                out[sel] = oic.lnpost(descs[b], np.ascontiguousarray(pars[sel].T), nthreads=8, parts=False)
        return out
    sel = torch.as_tensor(pick, device=pos.device)
    D = post.n_params
    p_sel = pos[sel].reshape(-1, D).cpu().numpy()
    l_sel = lnp[sel].reshape(-1).cpu().numpy()
    fs = FusedEnsembleSampler(post, W, a=2.0, seed=seed)
    fs.run_mcmc(pos.clone(), steps, lnprob0=lnp.clone(), store=True)
    S = post.n_models
    ch = fs.chain_steps.reshape(steps, S, W, D)[:, sel].reshape(steps, -1, D).cpu().numpy()
    cl = fs._lnprob.view(steps, S, W)[:, sel].reshape(steps, -1).cpu().numpy()
    st = _replay.replay(p_sel, l_sel, ch, cl, W, 2.0, seed, 0, fn, star_of_block=pick, lnp_atol=1e-7, margin=1e-8)
    fs.close()
    assert st["moves"] == steps * W * len(pick) and st["near_ties"] <= 2, (what, st)


# ---- the fused families: one test per (parametrisation, stars, bands) ---------------------------------------------------
SHAPES = [("track", 1), ("iso", 1), ("iso", 2), ("iso", 3)]
KIND_ID = {"track": 0, "iso": 1}


def expect(names, kernel, what):
    assert kernel in names, "%s: expected %s, the library launched %s" % (what, kernel, names)
    assert kernel in table(), "%s is not in the build's table" % kernel


@pytest.mark.parametrize("nb", list(range(13)))
@pytest.mark.parametrize("kind,ns", SHAPES)
def test_fused_families(kind, ns, nb):
    tid = "fused-%s%d-%d" % (kind, ns, nb)
    rng = np.random.default_rng(1000 * KIND_ID[kind] + 100 * ns + nb)
    K = KIND_ID[kind]
    bands = ia.grids.KNOWN_BANDS[:nb]
    ic, lo, hi = make_ic(kind, bands)
    oic = fx.make_oracle_ic(ic)
    x = batch_rows(rng, kind, ns, lo, hi)
    for astero in (False, True):
        a = "true" if astero else "false"
        with env(ISOCHRONES_AMD_PATH="auto", ISOCHRONES_AMD_STD_PRIORS=None):
            mod = make_model(ic, kind, ns, bands, astero=astero)
            assert mod.kernel_path() == "fused-packed"
            # batch kernel
            with traced(tid) as t:
                check_batch(mod, oic, x, "%s astero=%s" % (tid, a))
            expect(t.names, "k_lnpost_fast<%d, %d, %d, false, %s>" % (K, ns, nb, a), tid)
            p0 = start_ball(rng, mod, kind, ns, 16)
            if not astero:
                # the per-point callback: host rows in, host values out through the model's resident mailbox wave (one row in
                # the request line, several rows behind it) - the launch path's numbers bit for bit, and the oracle's
                __import__("time").sleep(0.005)          # a wave started by an earlier small call has left by now (idle 1 ms):
                with env(ISOCHRONES_AMD_MAILBOX=None), traced(tid) as t:      # this call starts - and names - a new one
                    one = mod.lnpost(p0[0])
                    few = mod.lnpost(x[:100])
                    pri = mod.lnprior(x[:100])
                expect(t.names, "k_mailbox_lnpost<%d, %d, %d>" % (K, ns, nb), tid)
                with env(ISOCHRONES_AMD_MAILBOX="0"):
                    assert one == mod.lnpost(p0[0])
                    assert np.array_equal(few, mod.lnpost(x[:100]), equal_nan=True) and np.array_equal(pri, mod.lnprior(x[:100]), equal_nan=True)
                w3 = oic.lnpost(mod.model_desc(), np.ascontiguousarray(x[:100].T), nthreads=4)
                fx.assert_close(few, w3[0], RTOL, atol=ATOL, what=tid + " mailbox lnpost")
                fx.assert_close(pri, w3[1], RTOL, atol=ATOL, what=tid + " mailbox lnprior")
            # step-wise sampler kernel (asteroseismic models have the persistent form only: asked for step-wise they take it)
            with env(ISOCHRONES_AMD_SAMPLER="stepwise"), traced(tid) as t:
                check_sampler(mod, oic, p0, 16, 10, 77 + nb, tid + " stepwise")
            expect(t.names, ("k_stretch_persist<%d, %d, %d, false, true, true, false>" if astero else "k_stretch_half<%d, %d, %d, false>") % (K, ns, nb), tid)
            # persistent, single model: default priors as compile-time constants / read at run time
            # (a single binary without asteroseismic terms takes the one-star-per-lane kernel unless told otherwise)
            pair = kind == "iso" and ns == 2 and not astero
            triple = kind == "iso" and ns == 3 and not astero
            with env(ISOCHRONES_AMD_SAMPLER="persistent", ISOCHRONES_AMD_STAR_LANES="0"), traced(tid) as t:
                check_sampler(mod, oic, p0, 16, 10, 78 + nb, tid + " persistent std priors")
            # (asteroseismic models: one persistent form, priors read at run time)
            expect(t.names, "k_stretch_persist<%d, %d, %d, false, %s, true, %s>" % (K, ns, nb, a, "false" if astero else "true"), tid)
            with env(ISOCHRONES_AMD_SAMPLER="persistent", ISOCHRONES_AMD_STD_PRIORS="0", ISOCHRONES_AMD_STAR_LANES="0"), traced(tid) as t:
                check_sampler(mod, oic, p0, 16, 10, 79 + nb, tid + " persistent run-time priors")
            expect(t.names, "k_stretch_persist<%d, %d, %d, false, %s, true, false>" % (K, ns, nb, a), tid)
            if pair:
                for W in ((16, 160) if nb <= 4 else (16, 48)):     # 8 / 24 / 80 moves per half-step: up to 16 per wave, or 32 (<= 4 bands)
                    pw = start_ball(rng, mod, kind, ns, W)
                    with env(ISOCHRONES_AMD_SAMPLER="persistent", ISOCHRONES_AMD_STAR_LANES=None), traced(tid) as t:
                        check_sampler(mod, oic, pw, W, 10, 178 + nb + W, tid + " one star per lane, std priors, W=%d" % W)
                    expect(t.names, "k_stretch_pair<%d>" % nb, tid)
            if triple:
                # one star per ROW of a wave (k_stretch_triple): 8 moves per half-step (half a row), 75 (two chunks of 64, the
                # second one partly filled), 150 (the reference's default 300 walkers: three chunks)
                for W in ((16, 128) if nb % 3 else (16, 100)):      # 8, 64, 50 moves per half-step (beyond 64 the plain form runs)
                    pw = start_ball(rng, mod, kind, ns, W)
                    with env(ISOCHRONES_AMD_SAMPLER="persistent", ISOCHRONES_AMD_STAR_LANES=None), traced(tid) as t:
                        check_sampler(mod, oic, pw, W, 6, 378 + nb + W, tid + " one star per row, std priors, W=%d" % W)
                    expect(t.names, "k_stretch_triple<%d>" % nb, tid)
                with env(ISOCHRONES_AMD_SAMPLER="persistent", ISOCHRONES_AMD_STAR_LANES=None), traced(tid) as t:
                    check_sampler(mod, oic, p0, 16, 6, 380 + nb, tid + " one star per row, three ensembles", n_ensembles=3)
                expect(t.names, "k_stretch_triple<%d>" % nb, tid)
            if not astero:
                # register-capped form with a single model (many ensembles of one star run it in rounds)
                # (default priors: the register-capped form with the families compiled in; read at run time when told so)
                with env(ISOCHRONES_AMD_SAMPLER="persistent-dense"), traced(tid) as t:
                    check_sampler(mod, oic, p0, 16, 8, 80 + nb, tid + " persistent dense, one model", n_ensembles=3)
                # (single stars with up to six bands; the other shapes have the run-time form only)
                expect(t.names, "k_stretch_persist<%d, %d, %d, true, false, false, %s>" % (K, ns, nb, "true" if (ns == 1 and nb <= 6) else "false"), tid)
                with env(ISOCHRONES_AMD_SAMPLER="persistent-dense", ISOCHRONES_AMD_DENSE_STDP="0"), traced(tid) as t:
                    check_sampler(mod, oic, p0, 16, 8, 81 + nb, tid + " persistent dense, one model, run-time priors", n_ensembles=3)
                expect(t.names, "k_stretch_persist<%d, %d, %d, true, false, false, false>" % (K, ns, nb), tid)
                if ns == 1:
                    # ONE ensemble per workgroup (what 258 and more walkers give; here: asked for): single stars read the
                    # star's block through scalar loads - the register-capped form with UNI
                    with env(ISOCHRONES_AMD_SAMPLER="persistent-dense", ISOCHRONES_AMD_PERSIST_GROUP="1"), traced(tid) as t:
                        check_sampler(mod, oic, p0, 16, 8, 82 + nb, tid + " persistent dense, one ensemble per workgroup", n_ensembles=3)
                    expect(t.names, "k_stretch_persist<%d, 1, %d, true, false, true, %s>" % (K, nb, "true" if nb <= 6 else "false"), tid)
                    with env(ISOCHRONES_AMD_SAMPLER="persistent-dense", ISOCHRONES_AMD_PERSIST_GROUP="1", ISOCHRONES_AMD_DENSE_STDP="0"), traced(tid) as t:
                        check_sampler(mod, oic, p0, 16, 8, 83 + nb, tid + " persistent dense, one ensemble per workgroup, run-time priors", n_ensembles=3)
                    expect(t.names, "k_stretch_persist<%d, 1, %d, true, false, true, false>" % (K, nb), tid)
            del mod
    # ---- the same table entries with NON-DEFAULT prior families in every slot: the arms of ln_pdf's run-time switch inside
    # the batch, step-wise, persistent (run-time priors), one-star-per-lane and register-capped kernels (the default-prior
    # passes above only ever execute the arms of the reference's defaults)
    which = "A" if nb % 2 == 0 else "B"
    for astero in (False, True):
        a = "true" if astero else "false"
        with env(ISOCHRONES_AMD_PATH="auto", ISOCHRONES_AMD_STD_PRIORS=None):
            mod = make_model(ic, kind, ns, bands, astero=astero, priors=which)
            w = "%s astero=%s priors %s" % (tid, a, which)
            with traced(tid) as t:
                check_batch(mod, oic, x, w)
            expect(t.names, "k_lnpost_fast<%d, %d, %d, false, %s>" % (K, ns, nb, a), tid)
            p0 = start_ball(rng, mod, kind, ns, 16)
            if not astero:
                with env(ISOCHRONES_AMD_SAMPLER="stepwise"), traced(tid) as t:
                    check_sampler(mod, oic, p0, 16, 10, 277 + nb, w + " stepwise")
                expect(t.names, "k_stretch_half<%d, %d, %d, false>" % (K, ns, nb), tid)
            # (the library sees that these are not the default families: the run-time form without being asked)
            with env(ISOCHRONES_AMD_SAMPLER="persistent", ISOCHRONES_AMD_STAR_LANES="0"), traced(tid) as t:
                check_sampler(mod, oic, p0, 16, 10, 278 + nb, w + " persistent")
            expect(t.names, "k_stretch_persist<%d, %d, %d, false, %s, true, false>" % (K, ns, nb, a), tid)
            if kind == "iso" and ns in (2, 3) and not astero:
                # (non-default priors: the one-star-per-lane / -per-row kernels exist for the default families only)
                with env(ISOCHRONES_AMD_SAMPLER="persistent", ISOCHRONES_AMD_STAR_LANES=None), traced(tid) as t:
                    check_sampler(mod, oic, p0, 16, 10, 279 + nb, w + " star lanes allowed")
                expect(t.names, "k_stretch_persist<%d, %d, %d, false, false, true, false>" % (K, ns, nb), tid)
            if not astero:
                with env(ISOCHRONES_AMD_SAMPLER="persistent-dense"), traced(tid) as t:
                    check_sampler(mod, oic, p0, 16, 8, 280 + nb, w + " persistent dense", n_ensembles=3)
                expect(t.names, "k_stretch_persist<%d, %d, %d, true, false, false, false>" % (K, ns, nb), tid)
            del mod
    if nb >= 1:      # catalog (MULTI) forms: per-row star index
        with env(ISOCHRONES_AMD_PATH="auto"):
            with traced(tid) as t:          # (the per-star blocks are built on the device: k_catalog_copy_template, k_catalog_fill)
                cat, post = make_catalog(ic, kind, ns, bands, 24, 5 + nb)
                pos, lnp, good = check_catalog_batch(cat, post, ic, oic, ns, rng, tid + " catalog batch")
            expect(t.names, "k_lnpost_fast<%d, %d, %d, true, false>" % (K, ns, nb), tid)
            expect(t.names, "k_catalog_start<%d, %d, %d>" % (K, ns, nb), tid)
            # resident catalog kernel: shared default priors as compile-time families / every prior read at run time
            with env(ISOCHRONES_AMD_SAMPLER="persistent", ISOCHRONES_AMD_STD_PRIORS=None), traced(tid) as t:
                check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, 8, 90 + nb, tid + " catalog persistent, default priors")
            expect(t.names, "k_stretch_persist<%d, %d, %d, false, false, false, true>" % (K, ns, nb), tid)
            with env(ISOCHRONES_AMD_SAMPLER="persistent", ISOCHRONES_AMD_STD_PRIORS="0"), traced(tid) as t:
                check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, 8, 93 + nb, tid + " catalog persistent, run-time priors")
            expect(t.names, "k_stretch_persist<%d, %d, %d, false, false, false, false>" % (K, ns, nb), tid)
            with env(ISOCHRONES_AMD_SAMPLER="persistent-dense"), traced(tid) as t:
                check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, 8, 91 + nb, tid + " catalog dense")
            expect(t.names, "k_stretch_persist<%d, %d, %d, true, false, false, %s>" % (K, ns, nb, "true" if (ns == 1 and nb <= 6) else "false"), tid)
            with env(ISOCHRONES_AMD_SAMPLER="persistent-dense", ISOCHRONES_AMD_DENSE_STDP="0"), traced(tid) as t:
                check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, 8, 94 + nb, tid + " catalog dense, run-time priors")
            expect(t.names, "k_stretch_persist<%d, %d, %d, true, false, false, false>" % (K, ns, nb), tid)
            if ns == 1:
                # a catalog with one ensemble per workgroup: every workgroup reads ITS star's block through scalar loads
                with env(ISOCHRONES_AMD_SAMPLER="persistent-dense", ISOCHRONES_AMD_PERSIST_GROUP="1"), traced(tid) as t:
                    check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, 8, 95 + nb, tid + " catalog dense, one star per workgroup")
                expect(t.names, "k_stretch_persist<%d, 1, %d, true, false, true, %s>" % (K, nb, "true" if nb <= 6 else "false"), tid)
                with env(ISOCHRONES_AMD_SAMPLER="persistent-dense", ISOCHRONES_AMD_PERSIST_GROUP="1", ISOCHRONES_AMD_DENSE_STDP="0"), traced(tid) as t:
                    check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, 8, 96 + nb, tid + " catalog dense, one star per workgroup, run-time priors")
                expect(t.names, "k_stretch_persist<%d, 1, %d, true, false, true, false>" % (K, nb), tid)
            with env(ISOCHRONES_AMD_SAMPLER="stepwise"), traced(tid) as t:
                check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, 6, 92 + nb, tid + " catalog stepwise")
            expect(t.names, "k_stretch_half<%d, %d, %d, false>" % (K, ns, nb), tid)
            post.close()
            # catalog-wide NON-DEFAULT priors (reference catalog.py:117-124): shared by the stars, read at run time
            direct, _ = prior_set(which, kind)
            cat.set_prior(**{k: v for k, v in direct.items() if k != "distance"})       # (the distance prior stays per star)
            post = __import__("isochrones_amd.catalog", fromlist=["CatalogPosterior"]).CatalogPosterior.from_catalog(cat, ic, N=ns)
            with traced(tid) as t:
                pos, lnp, good = check_catalog_batch(cat, post, ic, oic, ns, rng, tid + " catalog batch, priors " + which)
            expect(t.names, "k_lnpost_fast<%d, %d, %d, true, false>" % (K, ns, nb), tid)
            expect(t.names, "k_catalog_start<%d, %d, %d>" % (K, ns, nb), tid)
            with env(ISOCHRONES_AMD_SAMPLER="persistent"), traced(tid) as t:
                check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, 8, 290 + nb, tid + " catalog persistent, priors " + which)
            expect(t.names, "k_stretch_persist<%d, %d, %d, false, false, false, false>" % (K, ns, nb), tid)
            with env(ISOCHRONES_AMD_SAMPLER="persistent-dense"), traced(tid) as t:
                check_catalog_sampler(cat, post, ic, oic, ns, pos, lnp, good, 8, 291 + nb, tid + " catalog dense, priors " + which)
            expect(t.names, "k_stretch_persist<%d, %d, %d, true, false, false, false>" % (K, ns, nb), tid)
            post.close()
    ic.release()
    RAN.add(tid)


@pytest.mark.parametrize("kind,ns", SHAPES)
def test_band_tiled_family(kind, ns):
    tid = "wide-%s%d" % (kind, ns)
    rng = np.random.default_rng(7000 + 10 * KIND_ID[kind] + ns)
    bands = ALL_BANDS[:17]
    ic, lo, hi = make_ic(kind, bands)
    oic = fx.make_oracle_ic(ic)
    with env(ISOCHRONES_AMD_PATH="auto"):
        mod = make_model(ic, kind, ns, bands)
        with traced(tid) as t:
            check_batch(mod, oic, batch_rows(rng, kind, ns, lo, hi), tid)
        expect(t.names, "k_lnpost_wide<%d, %d>" % (KIND_ID[kind], ns), tid)
        # the any-model persistent sampler with the band-tiled evaluation
        with traced(tid) as t:
            check_sampler(mod, oic, start_ball(rng, mod, kind, ns, 16), 16, 10, 7100 + ns, tid + " sampler")
        expect(t.names, "k_stretch_wide<%d, %d>" % (KIND_ID[kind], ns), tid)
    ic.release()
    RAN.add(tid)


@pytest.mark.parametrize("nb", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("kind,ns", SHAPES)
def test_generic_family(kind, ns, nb):
    """k_lnpost<KIND, NS, 0>: the generic fall-back, band loop at run time (one form per shape since round 6)."""
    tid = "generic-%s%d-%d" % (kind, ns, nb)
    rng = np.random.default_rng(9000 + 100 * KIND_ID[kind] + 10 * ns + nb)
    bands = ia.grids.KNOWN_BANDS[:nb]
    K = KIND_ID[kind]
    with env(ISOCHRONES_AMD_PATH="generic"):
        ic, lo, hi = make_ic(kind, bands)
        oic = fx.make_oracle_ic(ic)
        mod = make_model(ic, kind, ns, bands, astero=(nb % 3 == 1))
        assert mod.kernel_path() == "generic"
        x = batch_rows(rng, kind, ns, lo, hi)
        with traced(tid) as t:
            check_batch(mod, oic, x, tid)
    expect(t.names, "k_lnpost<%d, %d, 0>" % (K, ns), tid)
    ic.release()
    RAN.add(tid)


# ---- observation-tree kernels ---------------------------------------------------------------------------------------------
def tree_model(ic, nb, leaves):
    """A tree with `nb` distinct bands and `leaves` stars: unresolved catalogue bands on 1-3 stars of one system; from four
    stars on an AO image splits a companion off (two systems)."""
    import isochrones_amd.observation as api
    bands = list(ia.grids.KNOWN_BANDS[:nb])
    t = api.ObservationTree(name="dispatch")
    for j, b in enumerate(bands):
        o = api.Observation("cat", b, 4.0)
        o.add_source(api.Source(10.0 + 0.2 * j, 0.02))
        t.add_observation(o)
    kw = dict(parallax=(3.0, 0.05), Teff=(5800.0, 150.0))
    if leaves <= 3:
        return ia.TreeStarModel(ic, obs=t, N=leaves, index=0, **kw)
    o = api.Observation("AO", bands[0], 0.1)
    o.add_source(api.Source(0.0, 0.02, separation=0.0, pa=0.0, relative=True, is_reference=True))
    o.add_source(api.Source(1.5, 0.03, separation=0.4, pa=30.0, relative=True))
    t.add_observation(o)
    N = {4: [2, 2], 5: [3, 2], 6: [3, 3]}[leaves]
    return ia.TreeStarModel(ic, obs=t, N=N, index=[0, 1], **kw)


def check_tree(mod, oic, rng, what):
    from oracle import oracle as orc
    names = list(mod.param_names)
    c = np.array([340.0 if nm.startswith("eep") else {"age": 9.6, "feh": 0.0, "distance": 330.0, "AV": 0.1}[nm.split("_")[0]] for nm in names])
    w = np.array([25.0 if nm.startswith("eep") else {"age": 0.15, "feh": 0.1, "distance": 30.0, "AV": 0.05}[nm.split("_")[0]] for nm in names])
    n = 4096
    x = c + w * rng.standard_normal((n, c.size))
    i = 0
    for s in mod.obs.systems:
        k = mod.obs.Nstars[s]
        x[: 3 * n // 4, i:i + k] = -np.sort(-x[: 3 * n // 4, i:i + k], axis=1)
        i += 4 + k
    for j in range(c.size):
        x[n - 6 * (j + 1): n - 6 * (j + 1) + 4, j] = [np.nan, np.inf, -np.inf, 0.0]
    w_post, w_prior, w_like = orc.tree_lnpost(oic, mod.tree_desc(), np.ascontiguousarray(x.T), nthreads=8)
    assert np.isfinite(w_post).sum() > n // 50, what
    fx.assert_close(mod.lnpost(x), w_post, RTOL, atol=ATOL, what=what + " lnpost")
    fx.assert_close(mod.lnprior(x), w_prior, RTOL, atol=ATOL, what=what + " lnprior")
    fx.assert_close(mod.lnlike(x), w_like, RTOL, atol=ATOL, what=what + " lnlike")
    # the per-point callback: one row at a time through the model's resident mailbox wave (k_mailbox_tree,
    # fast/tree_mailbox.h) - bit for bit what a one-row launch of the batch kernel gives, special values included.  (Against the
    # same row INSIDE a batch only to rounding: a node above one model star skips the reference's -2.5 log10(10^(-0.4 m)) = m
    # round trip unless some sample of its wavefront is beyond |m| = 700 mag - then the whole wave takes the logarithm, and
    # its other samples differ from the short cut by an ulp.  The special rows of x sit in the last waves of the batch.)
    rows = list(range(24)) + list(range(n - 6 * c.size, n - 6 * c.size + 12))
    batch = [np.asarray(f(x[rows])) for f in (mod.lnpost, mod.lnprior, mod.lnlike)]
    with env(ISOCHRONES_AMD_MAILBOX=None):
        one = [np.array([f(list(x[r])) for r in rows]) for f in (mod.lnpost, mod.lnprior, mod.lnlike)]
    with env(ISOCHRONES_AMD_MAILBOX="0"):
        launch = [np.array([f(list(x[r])) for r in rows]) for f in (mod.lnpost, mod.lnprior, mod.lnlike)]
    for a_, b_, c_ in zip(one, batch, launch):
        assert np.array_equal(a_, c_, equal_nan=True), what + " mailbox wave vs one-row launch"
        fx.assert_close(a_, b_, 1e-13, atol=1e-13, what=what + " mailbox wave vs batch")


def check_tree_sampler(mod, oic, rng, what, W=16, steps=10, seed=5):
    from oracle import oracle as orc
    names = list(mod.param_names)
    c = np.array([340.0 if nm.startswith("eep") else {"age": 9.6, "feh": 0.0, "distance": 330.0, "AV": 0.1}[nm.split("_")[0]] for nm in names])
    w = np.array([3.0 if nm.startswith("eep") else {"age": 0.02, "feh": 0.02, "distance": 3.0, "AV": 0.01}[nm.split("_")[0]] for nm in names])
    desc = mod.tree_desc()

    def fn(blk, pars):
        return orc.tree_lnpost(oic, desc, np.ascontiguousarray(pars.T), nthreads=8)[0]
    p0 = None
    for _ in range(20):
        x = c + w * rng.standard_normal((8 * W, c.size))
        i = 0
        for s in mod.obs.systems:
            k = mod.obs.Nstars[s]
            x[:, i:i + k] = -np.sort(-x[:, i:i + k], axis=1)
            i += 4 + k
        good = np.flatnonzero(np.isfinite(fn(None, x)))
        if good.size >= W:
            p0 = x[good[:W]]
            break
    assert p0 is not None, what + ": no start points"
    check_sampler(mod, oic, p0, W, steps, seed, what, fn=fn)


@pytest.mark.parametrize("nb", list(range(1, 17)))
def test_tree_families(nb):
    tid = "tree-%d" % nb
    rng = np.random.default_rng(11000 + nb)
    ic, lo, hi = make_ic("iso", ia.grids.KNOWN_BANDS[:nb])
    oic = fx.make_oracle_ic(ic)
    with env(ISOCHRONES_AMD_PATH="auto", ISOCHRONES_AMD_TREE_RUNTIME_LEAVES=None):
        for leaves in ((1, 2, 3, 4, 5) if nb <= 8 else (2, 5)):
            mod = tree_model(ic, nb, leaves)
            with traced(tid) as t:
                check_tree(mod, oic, rng, "%s leaves=%d" % (tid, leaves))
            nl = leaves if (leaves <= 4 and nb <= 8) else 0
            # (13-16 bands, round 6: the band-tiled runtime-leaf form, laid out for ISO_TREE_MAX_BANDS = 16 bands)
            expect(t.names, "k_lnpost_tree_fast<%d, %d>" % (nb if nb <= 12 else 16, nl), tid)
            if nl:      # the per-point callback's resident wave (register form only)
                expect(t.names, "k_mailbox_tree<%d, %d>" % (nb, nl), tid)
            with traced(tid) as t:
                check_tree_sampler(mod, oic, rng, "%s leaves=%d sampler" % (tid, leaves), seed=300 + 10 * nb + leaves)
            expect(t.names, "k_stretch_tree<%d, %d>" % (nb if nb <= 12 else 16, nl), tid)
    if nb == 3:       # the generic tree kernel (any shape; here by request)
        with env(ISOCHRONES_AMD_PATH="generic"):
            ic2, _, _ = make_ic("iso", ia.grids.KNOWN_BANDS[:nb])
            mod = tree_model(ic2, nb, 4)
            with traced(tid) as t:
                check_tree(mod, fx.make_oracle_ic(ic2), rng, tid + " generic")
            expect(t.names, "k_lnpost_tree", tid)
            ic2.release()
    ic.release()
    RAN.add(tid)


# ---- IsoTrackModel: both grids inside one persistent sampler kernel -------------------------------------------------------
@pytest.mark.parametrize("nb", list(range(13)))
def test_isotrack_family(nb):
    """k_stretch_isotrack<NB>: reference starmodel.py:2010-2104 composed from the oracle's two evaluations."""
    import math
    tid = "isotrack-%d" % nb
    rng = np.random.default_rng(15000 + nb)
    bands = ia.grids.KNOWN_BANDS[:nb]
    with env(ISOCHRONES_AMD_PATH="auto"):
        iso, _, _ = make_ic("iso", bands)
        track, _, _ = make_ic("track", bands)
        truth = np.array([355.0, 1.0, 9.6, 0.0, 300.0, 0.1])           # (eep, mass, age, feh, distance, AV)
        obs = dict(Teff=(5700.0, 150.0), feh=(0.0, 0.15), parallax=(1000.0 / truth[4], 0.05))
        if nb:
            mags = track.interp_mag([truth[1], truth[0], truth[3], truth[4], truth[5]], list(bands))[3]
            for j, b in enumerate(bands):
                obs[b] = (float(mags[j]), 0.05)
        mod = ia.IsoTrackModel(iso, track, **obs)
        oi, ot = fx.make_oracle_ic(iso), fx.make_oracle_ic(track)
        di, dt = mod._iso_model.model_desc(), mod._track_model.model_desc()
        lo, hi, lnorm = mod.age_prior_constants()

        def fn(blk, p):
            iso_p = np.column_stack([p[:, 0], p[:, 2], p[:, 3], p[:, 4], p[:, 5]])
            trk_p = np.column_stack([p[:, 1], p[:, 0], p[:, 3], p[:, 4], p[:, 5]])
            t = ot.lnpost(dt, trk_p.T.copy(), nthreads=8)
            i = oi.lnpost(di, iso_p.T.copy(), nthreads=8)
            age = p[:, 2]
            with np.errstate(all="ignore"):
                ln_age = np.where((age < lo) | (age > hi), -np.inf, lnorm + age * math.log(10))
                prior = t[1] + ln_age
                return np.where(np.isfinite(prior), prior + (i[2] + t[2]), -np.inf)
        W, p0 = 16, None
        for _ in range(20):
            x = truth + np.array([2.0, 0.01, 0.02, 0.01, 2.0, 0.01]) * rng.standard_normal((8 * W, 6))
            x[:, 5] = np.abs(x[:, 5])
            good = np.flatnonzero(np.isfinite(fn(None, x)))
            if good.size >= W:
                p0 = x[good[:W]]
                break
        assert p0 is not None, tid + ": no start points"
        # the batch composition (two fused launches + framework ops) agrees with the oracle's ...
        fx.assert_close(mod.lnpost(p0), fn(None, p0), RTOL, atol=1e-8, what=tid + " batch")
        # ... one point at a time - two per-point calls through the component models' resident waves and the age prior on the
        # host (IsoTrackModel._scalar_parts) - gives the batch's numbers (to rounding: the host adds where the batch's framework
        # kernel may fuse), an age outside the prior's support included
        rows = np.vstack([p0[:8], p0[:2] * [1, 1, 0, 1, 1, 1] + [0, 0, 20.0, 0, 0, 0]])
        for f in (mod.lnpost, mod.lnprior, mod.lnlike):
            fx.assert_close(np.array([f(list(r)) for r in rows]), np.asarray(f(rows)), 1e-13, atol=1e-13, what=tid + " one point at a time")
        # ... and so does every move of the resident sampler
        with traced(tid) as t:
            check_sampler(mod, None, p0, W, 10, 15100 + nb, tid + " sampler", fn=fn)
        expect(t.names, "k_stretch_isotrack<%d>" % nb, tid)
    iso.release()
    track.release()
    RAN.add(tid)


# ---- interpolation primitives -----------------------------------------------------------------------------------------------
def test_interpolation_families():
    tid = "interp"
    import torch
    rng = np.random.default_rng(12000)
    from oracle import oracle as orc
    # k_interp<2|3|4>: DFInterpolator over 2-, 3-, 4-axis tables
    for nd in (2, 3, 4):
        axes = [np.sort(rng.uniform(0, 10, n)) for n in (7, 9, 11, 5)[:nd]]
        grid = rng.standard_normal(tuple(a.size for a in axes) + (5,))
        dfi = ia.DFInterpolator.from_arrays(grid, axes, ["c%d" % j for j in range(5)])
        xs = [rng.uniform(a[0] - 0.2, a[-1] + 0.2, 5000) for a in axes]
        with traced(tid) as t:
            got = dfi(xs, ["c1", "c3", "c4"])
        want = orc.OracleTable(grid, axes).interp(xs, [1, 3, 4])
        fx.assert_close(got, want, 1e-11, atol=1e-12, what="k_interp<%d>" % nd)
        expect(t.names, "k_interp<%d>" % nd, tid)
        # ONE point: the context's resident service wave (kernels/k_service.h) - the launch path's numbers bit for bit, the
        # oracle's to rounding; points outside the table and NaN coordinates included
        __import__("time").sleep(0.005)                   # (a wave an earlier one-point call started has left: idle 1 ms)
        pts = [[float(x[i]) for x in xs] for i in range(40)] + [[float("nan")] + [float(x[0]) for x in xs[1:]]]
        with env(ISOCHRONES_AMD_MAILBOX=None), traced(tid) as t:
            one = np.array([dfi(p, ["c1", "c3", "c4"]) for p in pts])
            all5 = np.array([dfi(p) for p in pts[:8]])
        expect(t.names, "k_service", tid)
        with env(ISOCHRONES_AMD_MAILBOX="0"):
            assert np.array_equal(one, np.array([dfi(p, ["c1", "c3", "c4"]) for p in pts]), equal_nan=True)
            assert np.array_equal(all5, np.array([dfi(p) for p in pts[:8]]), equal_nan=True)
        fx.assert_close(one[:40], want[:40], 1e-11, atol=1e-12, what="service wave, %d-D table" % nd)
        assert np.isnan(one[40]).all()
    for kind in ("track", "iso"):
        K = KIND_ID[kind]
        for nb in range(1, 13):
            bands = ia.grids.KNOWN_BANDS[:nb]
            ic, lo, hi = make_ic(kind, bands)
            oic = fx.make_oracle_ic(ic)
            x = batch_rows(rng, kind, 1, lo, hi, n=40_000)
            x[:, 3] = np.abs(x[:, 3]) + 1.0
            wT, wg, wf, wm = oic.interp_mag(np.ascontiguousarray(x.T), [ic.bc_grid.interp.column_index[b] for b in bands], nthreads=8)
            with traced(tid) as t:
                T, g_, f, m = ic.interp_mag([x[:, j] for j in range(5)], list(bands))        # >= 32 768 rows: the packed form
            fx.assert_close(T, wT, RTOL, what="Teff"); fx.assert_close(m, wm, RTOL, atol=ATOL, what="mags %s %d" % (kind, nb))
            expect(t.names, "k_interp_mag_fast<%d, %d>" % (K, nb), tid)
            if nb in (1, 2, 5, 12):
                # one point at a time: interp_mag / interp_value through the resident service wave, alternating targets (the
                # wave restages its axes when the target changes) - bit for bit the one-point launch, and the oracle's numbers
                rows = [list(map(float, x[i])) for i in range(24)]
                mi = ic.model_grid.interp
                order = [2, 0, 1] if kind == "track" else [1, 2, 0]
                props = ["Teff", "logg", "Mbol"]
                with env(ISOCHRONES_AMD_MAILBOX=None), traced(tid) as t:
                    got_m = [ic.interp_mag(r, list(bands)) for r in rows]
                    got_v = np.array([ic.interp_value(r, props) for r in rows])
                expect(t.names, "k_service", tid)
                with env(ISOCHRONES_AMD_MAILBOX="0"):
                    ref_m = [ic.interp_mag(r, list(bands)) for r in rows]
                    ref_v = np.array([ic.interp_value(r, props) for r in rows])
                for a_, b_ in zip(got_m, ref_m):
                    assert all(np.array_equal(np.asarray(u), np.asarray(v), equal_nan=True) for u, v in zip(a_, b_))
                assert np.array_equal(got_v, ref_v, equal_nan=True)
                fx.assert_close(np.array([np.asarray(g4[3]) for g4 in got_m]), wm[:24], RTOL, atol=ATOL, what="service wave mags %s %d" % (kind, nb))
                fx.assert_close(np.array([g4[0] for g4 in got_m]), wT[:24], RTOL, what="service wave Teff")
                want_v = orc.OracleTable(mi.grid, mi.index_columns).interp([x[:24, order[0]], x[:24, order[1]], x[:24, order[2]]],
                                                                            [mi.column_index[c] for c in props])
                fx.assert_close(got_v, want_v, RTOL, atol=ATOL, what="service wave interp_value")
            if nb == 2:
                with traced(tid) as t:
                    T, g_, f, m = ic.interp_mag([x[:500, j] for j in range(5)], list(bands))  # small batch: column-parallel form
                fx.assert_close(m, wm[:500], RTOL, atol=ATOL, what="mags small batch")
                expect(t.names, "k_interp_mag<%d>" % K, tid)
                # wide pack of the model table (3-D tables, batches >= 32 768 rows) and the ragged-age EEP search
                cols = ["Teff", "logg", "Mbol"]
                with traced(tid) as t:
                    v = ic.interp_value([x[:, 0], x[:, 1], x[:, 2]], cols)
                mi = ic.model_grid.interp
                order = [2, 0, 1] if kind == "track" else [1, 2, 0]         # (mass, eep, feh) -> (feh, mass, eep); (eep, age, feh) -> (age, feh, eep)
                want = orc.OracleTable(mi.grid, mi.index_columns).interp([x[:, order[0]], x[:, order[1]], x[:, order[2]]],
                                                                          [mi.column_index[c] for c in cols], nthreads=8)
                fx.assert_close(v, want, RTOL, atol=ATOL, what="interp_value wide")
                expect(t.names, "k_interp3_wide<8>", tid)
                # one column: the four-pass form
                with traced(tid) as t:
                    v1 = ic.interp_value([x[:, 0], x[:, 1], x[:, 2]], cols[1:2])
                fx.assert_close(np.asarray(v1).reshape(-1), want[:, 1], RTOL, atol=ATOL, what="interp_value wide, one column")
                expect(t.names, "k_interp3_wide<4>", tid)
            ic.release()
    RAN.add(tid)


def test_eep_unit_cube_and_summary_families():
    tid = "misc"
    import torch
    rng = np.random.default_rng(13000)
    # k_interp_eep: (mass, age, feh) -> EEP on the ragged age table
    ic = ia.synthetic_track(bands=("G",))
    m = rng.uniform(0.3, 3.0, 2000); a = rng.uniform(8.0, 10.0, 2000); f = rng.uniform(-1.0, 0.4, 2000)
    with traced(tid) as t:
        e = ic.get_eep(m, a, f)
    expect(t.names, "k_interp_eep", tid)
    from oracle import oracle as orc
    # the reference's interp_eeps returns 1 + the fractional row index (its EEP axis starts at 1): this table's does too
    want = orc.interp_eep(a, f, m, np.asarray(ic.model_grid.fehs, float), np.asarray(ic.model_grid.masses, float), ic._age_grid, ic._array_lengths)
    assert np.isfinite(want).sum() > 500
    fx.assert_close(e, want + (float(ic.model_grid.interp.index_columns[2][0]) - 1.0), 1e-12, what="get_eep vs oracle")
    # ... and one star at a time through the resident service wave: the same numbers bit for bit
    __import__("time").sleep(0.005)
    with env(ISOCHRONES_AMD_MAILBOX=None), traced(tid) as t:
        e1 = np.array([ic.get_eep(float(m[i]), float(a[i]), float(f[i])) for i in range(60)])
    expect(t.names, "k_service", tid)
    assert np.array_equal(e1, np.asarray(e)[:60], equal_nan=True)
    with env(ISOCHRONES_AMD_MAILBOX="0"):
        assert np.array_equal(e1, np.array([ic.get_eep(float(m[i]), float(a[i]), float(f[i])) for i in range(60)]), equal_nan=True)
    # a catalog fit whose batch holds a star without start points (k_catalog_patch_failed keeps the batch rectangular on the
    # device): the rows of the host-checked path of rounds 1-5, bit for bit; the hopeless star's row is blank with ok = 0
    from isochrones_amd.catalog import fit_stars_gpu
    icc = ia.synthetic_track(bands=("G", "BP", "RP"))
    cat, _ = ia.synthetic_catalog(icc, 40, bands=["G", "BP", "RP"], seed=3, mag_unc=0.01)
    cat.measurements["G"][1][7] = 0.0                 # an uncertainty of zero: log(0) in every candidate's likelihood - no start point
    with traced(tid) as t:
        rows_lean = fit_stars_gpu(cat, icc, np.arange(40), nwalkers=32, nburn=20, niter=20, seed=4)
    expect(t.names, "k_catalog_patch_failed", tid)
    with env(ISOCHRONES_AMD_CATALOG_LEAN="0"):
        rows_host = fit_stars_gpu(cat, icc, np.arange(40), nwalkers=32, nburn=20, niter=20, seed=4)
    assert np.array_equal(rows_lean, rows_host, equal_nan=True)
    assert rows_lean[7, -1] == 0.0 and np.isnan(rows_lean[7, :-1]).all() and (rows_lean[np.arange(40) != 7, -1] == 1.0).all()
    icc.release()
    # k_unit_cube
    mod = ia.SingleStarModel(ic, Teff=(5770, 100), G=(10.0, 0.02))
    cube = rng.uniform(size=(1000, 5))
    with traced(tid) as t:
        got = mod.mnest_prior(torch.as_tensor(cube.copy(), device="cuda")).cpu().numpy()
    lo = np.array([mod.bounds(p)[0] for p in mod.param_names]); hi = np.array([mod.bounds(p)[1] for p in mod.param_names])
    assert np.array_equal(got, (hi - lo) * cube + lo)
    expect(t.names, "k_unit_cube", tid)
    ic.release()
    # chain summaries: every wave / exact / workgroup instantiation against numpy.quantile
    import ctypes as C
    from isochrones_amd import device as dev
    lib, ctx = _cabi.lib(), dev.context(0)
    q = np.array([0.5, 0.16, 0.84])
    shapes = {}
    for full in (12, 25, 50, 100):
        for tail in (False, True):
            Wk = 32
            steps = (full * 64 + (32 if tail else 0)) // Wk
            shapes["k_chain_quantiles_exact<%d, %s>" % (full, "true" if tail else "false")] = (steps, Wk, None)
    shapes["k_chain_quantiles_wave<52>"] = (37, 30, None)           # 1 110 values, W does not divide 64
    shapes["k_chain_quantiles_wave<104>"] = (170, 30, None)         # 5 100 values
    shapes["k_chain_quantiles_select"] = (250, 30, "workgroup")
    shapes["k_chain_quantiles"] = (100, 30, "sort")
    shapes["k_chain_quantiles_big"] = (100, 300, None)              # 30 000 values per pair: the reference's default fit
    for kernel, (steps, Wk, mode) in shapes.items():
        S, D = 40, 5
        host = rng.standard_normal((steps, D, S * Wk)).round(2)                                       # rounded: ties
        chain = torch.as_tensor(host, device="cuda")                                                  # parameter-major
        out = torch.empty(S, D, 3, dtype=torch.float64, device="cuda")
        with env(ISOCHRONES_AMD_QUANTILES=mode), traced(tid) as t:
            _cabi.check(lib.iso_chain_quantiles_layout(ctx, dev.ptr(chain), _cabi.CHAIN_PARAM_MAJOR, steps, S, Wk, D,
                                                       q.ctypes.data_as(C.POINTER(C.c_double)), 3, dev.ptr(out), None))
            torch.cuda.synchronize()
        want = np.quantile(host.reshape(steps, D, S, Wk).transpose(2, 1, 0, 3).reshape(S, D, -1), q, axis=2).transpose(1, 2, 0)
        assert np.array_equal(out.cpu().numpy(), want), kernel
        expect(t.names, kernel, tid)
    RAN.add(tid)


# ---- the table is closed -------------------------------------------------------------------------------------------------------
SETUP_KERNELS = {"k_pack_hot", "k_pack_bc", "k_pack_corners", "k_pack_wide", "k_catalog_copy_template", "k_catalog_fill", "k_signal_done"}


def test_every_compiled_kernel_was_launched_and_checked():
    """The kernels the tracer saw while the tests above ran = the kernels hipcc compiled for the library.  (The table-packing
    and catalog set-up kernels have no check of their own: every result above was computed from tables they laid out.)"""
    expected_tests = ({"fused-%s%d-%d" % (k, n, b) for k, n in SHAPES for b in range(13)} | {"wide-%s%d" % s for s in SHAPES}
                      | {"generic-%s%d-%d" % (k, n, b) for k, n in SHAPES for b in range(10)} | {"tree-%d" % b for b in range(1, 17)}
                      | {"isotrack-%d" % b for b in range(13)} | {"interp", "misc"})
    if RAN != expected_tests:
        pytest.skip("only part of this file ran (%d of %d enumeration tests): the closure check needs all of them"
                    % (len(RAN), len(expected_tests)))
    compiled = set(table())
    seen = set(SEEN)
    assert not (seen - compiled), "launched but not in the build's table: %s" % sorted(seen - compiled)[:10]
    missing = compiled - seen
    assert not missing, "%d compiled kernels were never launched by this file, e.g. %s" % (len(missing), sorted(missing)[:20])
    assert SETUP_KERNELS <= seen
    print("dispatch table: %d kernels compiled, %d launched and checked" % (len(compiled), len(seen)))
