"""Priors the device has no family for (reference: ``StarModel.set_prior`` takes ANY ``Prior`` object,
isochrones/starmodel.py:629-632; ``lnprior`` sums ``self._priors[prop].lnpdf(value)``, :1616-1635; the base class a user
subclasses defines ``_pdf`` / ``_lnpdf`` / ``distribution``, isochrones/priors.py:31-73).

The device evaluates a flat stand-in over such a prior's bounds and the model adds ``lnpdf`` on the host
(starmodel._HostPriorMixin).  Checked here through every calling form of a single-star model, a binary, an observation-tree
model and an IsoTrackModel against the SAME model with device priors, term by term:

    lnprior(host priors) = lnprior(defaults) - sum default.lnpdf(x) + sum user.lnpdf(x),   lnlike unchanged

plus the bounds / non-finite behaviour, and that such a model is fitted by the framework-op sampler (the resident kernels
refuse it, a catalog refuses it at ``set_prior``)."""
import math
import warnings

import numpy as np
import pytest

import isochrones_amd as ia
from isochrones_amd import priors as P

pytestmark = pytest.mark.gpu


class Triangle(P.Prior):
    """A user's prior in the reference's style: only ``_pdf`` (normalised by the base class's ``bounds`` setter)."""

    def __init__(self, bounds):
        self._norm = 1.0
        self.bounds = bounds

    def _pdf(self, x):
        lo, hi = self.bounds
        return (x - lo) + 0.05 * (hi - lo)


class ForeignExp:
    """Not derived from anything of ours: ``lnpdf`` / ``bounds`` / ``sample(n)`` (the reference's signature)."""

    def __init__(self, scale, hi):
        self.scale, self.bounds = scale, (0.0, hi)
        self._z = scale * (1 - math.exp(-hi / scale))

    def lnpdf(self, x):
        lo, hi = self.bounds
        return -math.inf if (x < lo or x > hi) else -x / self.scale - math.log(self._z)

    def sample(self, n):
        u = np.random.default_rng(5).random(n)
        return -self.scale * np.log(1 - u * (1 - math.exp(-self.bounds[1] / self.scale)))


def _ln(prior, x):
    return np.array([prior.lnpdf(float(v)) for v in np.atleast_1d(x)])


def _iso():
    ages = ia.grids.mist_log_ages()[60::2]
    return ia.synthetic_isochrone(bands=("J", "H", "K"), ages=ages, fehs=[-1.0, -0.5, 0.0, 0.5], eeps=np.arange(150.0, 700.0),
                                  eep_bounds=(150, 699), limits=dict(age=(ages[0], ages[-1]), feh=(-1.0, 0.5)))


def _rows(mod, n, rng, centre, width):
    x = np.asarray(centre) + np.asarray(width) * rng.standard_normal((n, len(centre)))
    if mod.N > 1:
        x[:, :mod.N] = -np.sort(-x[:, :mod.N], axis=1)
    return x


def _check_all_forms(mod, ref, x, host, tol=1e-12):
    """``host``: {parameter name: (column list, user prior, default prior)}."""
    import torch
    want_like = ref.lnlike(x)
    want_prior = ref.lnprior(x).copy()
    for cols, user, default in host.values():
        for c in cols:
            with np.errstate(invalid="ignore"):
                want_prior = want_prior - _ln(default, x[:, c]) + _ln(user, x[:, c])
    # a non-finite user term is the prior (and the posterior); rows the default prior already excluded stay excluded
    # (a row whose EEP term is NaN - off the grid - is NaN with either set of priors)
    dead = np.isneginf(ref.lnprior(x))
    for cols, user, _ in host.values():
        for c in cols:
            dead |= ~np.isfinite(_ln(user, x[:, c]))
    want_prior = np.where(dead, -np.inf, want_prior)
    ref_post = ref.lnpost(x)
    with np.errstate(invalid="ignore"):
        want_post = np.where(dead, -np.inf, np.where(np.isfinite(ref_post), want_prior + want_like, ref_post))
    assert np.isfinite(want_post).sum() > len(x) // 4 and dead.sum() > 0

    def same(got, want, what):
        got = np.asarray(got, dtype=float)
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin), what
        assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), what
        assert np.allclose(got[fin], want[fin], rtol=tol, atol=tol), (what, np.abs(got[fin] - want[fin]).max())

    xt = torch.as_tensor(x, device="cuda")
    for name, fn, want in (("lnpost", mod.lnpost, want_post), ("lnprior", mod.lnprior, want_prior), ("lnlike", mod.lnlike, want_like)):
        if name == "lnlike":            # (rows outside the model's bounds have no likelihood either way: compare the device's own)
            want = ref.lnlike(x)
            same(fn(x), want, name + " numpy")
            continue
        same(fn(x), want, name + " numpy")
        same(fn(xt).cpu().numpy(), want, name + " tensor")
        same([fn(list(r)) for r in x[:64]], want[:64], name + " one row at a time")
        same(fn(xt[3]).cpu().numpy()[None], want[3:4], name + " one tensor row")


def test_single_star_every_calling_form():
    ic = _iso()
    obs = dict(J=(9.6, 0.03), H=(9.2, 0.03), K=(9.1, 0.03), parallax=(8.0, 0.1), Teff=(5700, 100))
    ref = ia.SingleStarModel(ic, **obs)
    mod = ia.SingleStarModel(ic, **obs)
    tri = Triangle((-0.6, 0.4))
    far = ForeignExp(300.0, 240.0)                       # (narrower than the model's distance bounds: rows beyond are -inf)
    mod.set_prior(feh=tri, distance=far)
    assert mod.bounds("feh") == (-0.6, 0.4) and mod.bounds("distance") == (0.0, 240.0)
    # the reference model keeps its defaults, with the SAME bounds (the bounds test is the device's in both)
    ref.set_bounds(feh=(-0.6, 0.4), distance=(0.0, 240.0))
    rng = np.random.default_rng(3)
    x = _rows(mod, 4000, rng, [350.0, 9.6, -0.1, 125.0, 0.2], [40.0, 0.15, 0.35, 60.0, 0.15])
    host = {"feh": ([2], tri, ref._priors["feh"]), "distance": ([3], far, ref._priors["distance"])}
    _check_all_forms(mod, ref, x, host)
    # soa layout [n_params, N]
    import torch
    xt = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda")
    assert np.array_equal(mod.lnpost(xt, soa=True).cpu().numpy(), mod.lnpost(torch.as_tensor(x, device="cuda")).cpu().numpy(), equal_nan=True)
    # a prior swapped back restores the device path bit for bit
    mod.set_prior(feh=ia.priors.FehPrior(), distance=ia.priors.DistancePrior())
    mod.set_bounds(feh=(-0.6, 0.4), distance=(0.0, 240.0))
    assert np.array_equal(mod.lnpost(x), ref.lnpost(x), equal_nan=True)


def test_binary_and_isotrack():
    ic = _iso()
    obs = dict(J=(9.6, 0.03), H=(9.2, 0.03), K=(9.1, 0.03), parallax=(8.0, 0.1))
    ref, mod = ia.BinaryStarModel(ic, **obs), ia.BinaryStarModel(ic, **obs)
    tri = Triangle((0.0, 0.8))
    mod.set_prior(AV=tri)
    ref.set_bounds(AV=(0.0, 0.8))
    rng = np.random.default_rng(4)
    x = _rows(mod, 3000, rng, [380.0, 330.0, 9.6, -0.1, 125.0, 0.3], [30.0, 30.0, 0.15, 0.2, 15.0, 0.35])
    _check_all_forms(mod, ref, x, {"AV": ([5], tri, ref._priors["AV"])})

    from tests.test_tree_cpu import _isotrack_objects
    g, iso, track, o = _isotrack_objects()
    ref, mod = ia.IsoTrackModel(iso, track, **o), ia.IsoTrackModel(iso, track, **o)
    lo, hi = ref._track_model.bounds("feh")
    tri = Triangle((lo + 0.1 * (hi - lo), hi))
    mod.set_prior(feh=tri)
    mod.set_bounds(feh=tri.bounds)                       # (both component models: the isochrone-grid one only has bounds)
    ref.set_bounds(feh=tri.bounds)
    p = ref.sample_from_prior(1500, rng=np.random.default_rng(6))
    p = np.concatenate([p, p * (1 + 0.2 * rng.standard_normal(p.shape))])
    fcol = 3                                             # (eep, mass, age, feh, distance, AV)
    _check_all_forms(mod, ref, p, {"feh": ([fcol], tri, ref._track_model._priors["feh"])}, tol=1e-11)


def test_tree_model_every_system():
    import bench_configs
    ref, x = bench_configs.tree_model_and_samples(3000)
    mod, _ = bench_configs.tree_model_and_samples(16)
    tri = Triangle((0.0, 0.6))
    mod.set_prior(AV=tri)
    ref.set_bounds(AV=(0.0, 0.6))
    x[:, 5] = np.abs(x[:, 5]) * 3                        # (some rows beyond the prior's bounds)
    _check_all_forms(mod, ref, x, {"AV": (mod._host_columns("AV"), tri, ref._priors["AV"])}, tol=1e-11)
    assert mod._host_columns("AV") == [5] and mod._host_columns("age") == [2]


def test_fits_take_the_framework_sampler_and_the_kernels_refuse():
    from isochrones_amd.sampler import EnsembleSampler, FusedEnsembleSampler
    ic = _iso()
    mod = ia.SingleStarModel(ic, J=(9.6, 0.03), H=(9.2, 0.03), K=(9.1, 0.03), parallax=(8.0, 0.1))
    mod.set_prior(feh=Triangle((-0.6, 0.4)), distance=ForeignExp(300.0, 400.0))
    with pytest.raises(ValueError, match="host"):
        FusedEnsembleSampler(mod, 32)
    with pytest.raises(ValueError, match="host"):
        mod.fit_mcmc(nwalkers=32, nburn=5, niter=5, fused=True, seed=1)
    s = mod.fit_mcmc(nwalkers=32, nburn=30, niter=20, seed=1)
    assert isinstance(s, EnsembleSampler)
    df = mod.samples
    assert len(df) == 32 * 20 and np.isfinite(df["lnprob"]).all()
    assert df["feh"].between(-0.6, 0.4).all() and df["distance"].between(0.0, 400.0).all()
    # the stored lnprob is the model's lnpost (host terms included)
    cols = list(mod.param_names)
    assert np.allclose(mod.lnpost(df[cols].to_numpy()[:50]), df["lnprob"].to_numpy()[:50], rtol=1e-12, atol=1e-12)
    # nested fit: the callback form is lnpost
    res = mod.fit_multinest(n_live_points=100, seed=2)
    assert np.isfinite(res.logz)
    # a catalog is fitted inside the kernels: refused where the prior is set
    cat, _ = ia.synthetic_catalog(ic, 8, bands=["J", "H", "K"], seed=1)
    with pytest.raises(NotImplementedError, match="catalog"):
        cat.set_prior(feh=Triangle((-0.6, 0.4)))
    with pytest.raises(TypeError):
        mod.set_prior(AV=object())
