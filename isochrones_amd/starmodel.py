"""``BasicStarModel`` (+ Single/Binary/TripleStarModel): the posterior of an unresolved 1-3 star
system, evaluated by the fused HIP ``lnpost`` kernel.

Call surface follows the reference (isochrones/starmodel.py:1361-2007): constructor keywords
``Teff=(val, unc)``, ``logg=``, ``feh=``, ``parallax=``, ``nu_max=``, ``delta_nu=``, ``<band>=``,
``N``, ``eep_bounds``, ``maxAV``, ``max_distance``, ``halo_fraction``; ``param_names``,
``n_params``, ``bands``, ``spec_props``, ``bounds(prop)``, ``set_bounds``, ``set_prior``,
``lnpost(p)``, ``lnlike(p)``, ``lnprior(p)``, ``mnest_prior(cube, ndim, nparams)``,
``mnest_loglike``.  ``p`` may be

* one parameter vector  -> python float (what emcee / MultiNest callbacks expect),
* a numpy array ``[N, n_params]`` (emcee ``vectorize=True`` convention) -> numpy ``[N]``,
* a CUDA float64 tensor ``[N, n_params]`` or, with ``soa=True``, ``[n_params, N]`` -> CUDA ``[N]``.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
import threading

import numpy as np

from . import _cabi, device as dev
from .priors import (AgePrior, AVPrior, ChabrierPrior, DistancePrior, FehPrior, DEVICE_PRIOR_TYPES, check_host_prior,
                     flat_stand_in, is_host_prior, lnpdf_array)

logger = logging.getLogger("isochrones_amd")

_NOT_A_BAND = ("RA", "dec", "ra", "Dec", "maxAV", "parallax", "AV", "logg", "Teff", "feh", "density",
               "separation", "PA", "resolution", "relative", "N", "index", "id", "nu_max", "delta_nu")


def _prior_state(priors):
    """Identity + mutation counters of the prior objects a model's device constants were packed from (prior
    objects may be shared between models, as reference tests/test_likelihood.py does)."""
    eep = priors["eep"]
    return tuple((id(p), getattr(p, "_version", 0), tuple(p.bounds) if is_host_prior(p) else None)
                 for p in (priors["mass"], priors["age"], priors["feh"], priors["distance"], priors["AV"], eep.orig_prior)
                 ) + (id(eep), tuple(eep.bounds))


class EEPPrior:
    """Marker for the EEP prior: pdf(eep) = orig_prior(orig(eep)) * d(orig)/d(eep), evaluated on
    the device by interpolating (age, dt_deep) or (mass, dm_deep) (reference: priors.py:409-429)."""

    def __init__(self, ic, orig_prior, bounds=None, owner=None):
        import weakref
        self.ic = ic
        self._owners = weakref.WeakSet()        # models whose device constants depend on this object
        if owner is not None:
            self._owners.add(owner)
        self._orig_prior = orig_prior
        self.bounds = tuple(bounds) if bounds is not None else tuple(ic.eep_bounds)
        self.orig_par = ic.eep_replaces

    @property
    def bounds(self):
        return self._bounds

    @bounds.setter
    def bounds(self, new):
        from . import priors as _p
        self._bounds = tuple(new)
        _p.EPOCH[0] += 1

    # As in the reference, the EEP term keeps the prior object it was built with: set_prior() on the
    # parameter EEP replaces does not reach it (starmodel.py:1447 + :629-632); assigning this attribute does.
    @property
    def orig_prior(self):
        return self._orig_prior

    @orig_prior.setter
    def orig_prior(self, prior):
        if not isinstance(prior, DEVICE_PRIOR_TYPES):
            raise NotImplementedError("prior %r is not evaluable on the device" % (prior,))
        self._orig_prior = prior
        from . import priors as _p
        _p.EPOCH[0] += 1
        for owner in list(self._owners):
            owner._dirty()


def _draw(prior, n, rng):
    """``prior.sample(n, rng)``; a foreign prior object (the reference's signature, priors.py:69) takes ``sample(n)``."""
    try:
        return np.asarray(prior.sample(n, rng), dtype=float)
    except TypeError:
        return np.asarray(prior.sample(n), dtype=float)


class _HostPriorMixin:
    """Priors the device has no family for (reference: ``StarModel.set_prior`` takes any ``Prior`` object,
    starmodel.py:629-632; ``lnprior`` sums ``prior.lnpdf(value)`` over the parameters, :1616-1635).  The device evaluates a
    flat stand-in over such a prior's bounds (``priors.flat_stand_in``: the bounds test stays in the kernel); the model
    adds ``lnpdf(x) + log(width)`` of those parameters on the host to ``lnprior`` and ``lnpost`` - after the device's sum,
    so the last bits differ from the reference's left-to-right sum (1e-15 relative; the parity tests' 1e-9 holds).  A
    non-finite host term wins over whatever the device returned (the reference returns the prior when it is not finite).
    Models with host priors are fitted by the framework-op sampler: the resident kernels cannot call back."""

    def _host_columns(self, name):          # parameter columns the prior of `name` applies to
        raise NotImplementedError

    def _host_terms(self):
        terms = []
        for name in ("mass", "age", "feh", "distance", "AV"):
            pr = self._priors[name]
            if not is_host_prior(pr):
                continue
            cols = self._host_columns(name)
            if cols:
                lo, hi = flat_stand_in(pr).bounds
                terms.append((tuple(cols), pr, float(np.log(hi - lo))))
        return tuple(terms)

    def _packed_prior(self, pr):
        return flat_stand_in(pr).desc() if is_host_prior(pr) else pr.desc()

    @staticmethod
    def _host_lnprior(terms, x2):
        """Sum of the host terms for the rows of ``x2`` [n, n_params] (numpy)."""
        hp = np.zeros(x2.shape[0])
        for cols, pr, back in terms:
            for c in cols:
                hp = hp + (lnpdf_array(pr, x2[:, c]) + back)
        return hp

    @staticmethod
    def _host_combine(out, hp):
        """device lnprior / lnpost + host terms; a non-finite host term is the result."""
        with np.errstate(invalid="ignore"):
            return np.where(np.isfinite(hp), out + hp, hp)

    def _host_adjust_tensors(self, terms, pars, outs):
        """``outs``: (post, prior) CUDA tensors of the rows ``pars`` [n, n_params] (either may be None)."""
        import torch
        hp = torch.as_tensor(self._host_lnprior(terms, pars.detach().cpu().numpy()), device=pars.device)
        fin = torch.isfinite(hp)
        return tuple(None if o is None else torch.where(fin, o + hp, hp) for o in outs)


class _ConvenienceMixin:
    """Small helpers of the reference's StarModel that only touch results or delegate to lnpost:
    ``prior`` (starmodel.py:634), ``maxlike`` (:821-833), ``random_samples`` (:1055-1069), persistence
    (:1205-1317, 1843-1959; container format in isochrones_amd/persist.py)."""

    def prior(self, prop, val, **kwargs):
        return self._priors[prop](val, **kwargs)

    def lnpost_polychord(self, theta):
        """PolyChord's callback form of ``lnpost``: (log-posterior, derived parameters) with no derived parameters
        (reference: starmodel.py:703-705).  ``theta`` may also be a batch [N, n_params]."""
        return self.lnpost(theta), [0.0]

    @property
    def directory(self):
        return getattr(self, "_directory", None) or "."

    def maxlike(self, p0=None, n_starts=4096, seed=0, **kwargs):
        """A (local) optimum of lnpost: Nelder-Mead on -lnpost like the reference, started from ``p0`` or —
        the batched evaluation makes this cheap — from the best of ``n_starts`` prior draws."""
        import scipy.optimize
        if p0 is None:
            cand = self.sample_from_prior(int(n_starts), rng=np.random.default_rng(seed))
            if hasattr(cand, "columns"):
                cand = cand[list(self.param_names)].values
            cand = np.ascontiguousarray(cand, dtype=float)
            lp = np.asarray(self.lnpost(cand))
            p0 = cand[int(np.nanargmax(np.where(np.isfinite(lp), lp, -np.inf)))]
        kwargs.setdefault("method", "Nelder-Mead")

        def fn(p):
            v = -self.lnpost(p)
            return v if np.isfinite(v) else 1e300
        return scipy.optimize.minimize(fn, np.asarray(p0, dtype=float), **kwargs)

    def random_samples(self, n, rng=None):
        """``n`` rows drawn (with replacement) from the posterior samples."""
        rng = rng or np.random.default_rng()
        samples = self.samples
        return samples.iloc[rng.integers(len(samples), size=int(n))].reset_index(drop=True)

    def save(self, filename, overwrite=True):
        from . import persist
        return persist.save_model(self, filename, overwrite=overwrite)

    @classmethod
    def load(cls, filename, ic=None, name=None):
        from . import persist
        return persist.load_model(cls, filename, ic=ic, name=name)

    def save_hdf(self, filename, path="", overwrite=False, append=False):
        from . import persist
        return persist.save_hdf(self, filename, path=path, overwrite=overwrite, append=append)

    @classmethod
    def load_hdf(cls, filename, path="", name=None, ic=None):
        from . import persist
        if not str(filename).endswith(".npz"):
            raise ImportError("reading the reference's HDF5 layout needs pytables, which is not installed; "
                              "models saved here are .npz containers")
        return persist.load_model(cls, filename, ic=ic, name=name)


def _run_mcmc_fit(model, nwalkers, nburn, niter, p0, seed, fused, n_ensembles=1):
    """Burn in, reset, sample with the device-resident sampler (``fused`` None: whenever the library has a kernel for the
    model's shape; False: the framework-op :class:`EnsembleSampler`, a debugging aid that evaluates lnpost through the
    batch kernels; True: raise if there is no device-resident form).  Reference: fit_mcmc_old, starmodel.py:889-972."""
    import torch
    from .sampler import EnsembleSampler, FusedEnsembleSampler
    rng = np.random.default_rng(seed)
    npars = model.n_params
    if p0 is None:
        p0 = model.sample_from_prior(nwalkers * n_ensembles, rng=rng)
    else:
        centre = np.asarray(p0, dtype=float)
        p0 = rng.normal(size=(nwalkers * n_ensembles, npars)) * 0.01 + centre[None, :]
        bad = ~np.isfinite(model.lnpost(p0))
        p0[bad] = centre                       # a perturbed walker that left the support starts on the point itself
    p0 = np.asarray(p0, dtype=float)
    sampler = None
    host = getattr(model, "_host_terms", None)
    if host is not None and host():
        if fused:
            raise ValueError("a model with priors evaluated on the host cannot run the resident sampler (fused=True)")
        logger.warning("priors %s are evaluated on the host: fit by the framework-op sampler", [type(t[1]).__name__ for t in host()])
        fused = False
    if fused is None or fused:
        try:
            sampler = FusedEnsembleSampler(model, nwalkers, seed=int(rng.integers(2 ** 62)), n_ensembles=n_ensembles)
        except _cabi.IsoError as e:
            # only "the library has no resident kernel for this shape" is a reason to change samplers (the framework-op
            # one draws different random numbers and is an order of magnitude slower); a HIP / memory failure is an error
            if fused or e.rc not in (None, _cabi.ERR_INVALID):
                raise
            import warnings
            warnings.warn("fit_mcmc: no device-resident sampler for this model shape (%s); using the framework-op "
                          "EnsembleSampler, whose random numbers differ" % e, RuntimeWarning, stacklevel=3)
    if sampler is None:
        if n_ensembles != 1:
            raise ValueError("n_ensembles > 1 needs the device-resident sampler")
        sampler = EnsembleSampler(nwalkers, npars, model.lnpost, seed=int(rng.integers(2 ** 62)),
                                  device=torch.device("cuda", dev.current_device()))
    if n_ensembles > 1:
        p0 = p0.reshape(n_ensembles, nwalkers, npars)
    pos, prob = sampler.run_mcmc(p0, nburn, store=False)
    sampler.reset()
    sampler.run_mcmc(pos, niter, lnprob0=prob)
    return sampler


class _NestedFitMixin:
    """``fit_multinest`` / ``evidence`` for any model with ``param_names``, ``bounds(par)`` and a
    batched ``lnpost``: the reference hands ``mnest_loglike`` (= lnpost) and the flat-box ``mnest_prior``
    to pymultinest (starmodel.py:717-802); here the same likelihood and prior drive the batched nested
    sampler in isochrones_amd/nested.py, every proposal batch being one fused lnpost launch."""

    def fit_multinest(self, n_live_points=1000, basename=None, verbose=False, refit=False, overwrite=False,
                      test=False, evidence_tolerance=0.5, seed=0, batched=True, **kwargs):
        """``batched=True`` (default) retires n_live_points // 10 live points per macro-step with a vectorised host
        loop; ``batched=False`` is the classic one-point-per-iteration loop.  Same integral, same result object."""
        from .nested import nested_sample, nested_sample_batched
        names = list(self.param_names)
        lo = np.array([self.bounds(nm)[0] for nm in names], dtype=float)
        hi = np.array([self.bounds(nm)[1] for nm in names], dtype=float)
        run_kwargs = dict(nlive=int(n_live_points), tol=float(evidence_tolerance), seed=seed)
        allowed = ("enlarge", "max_batch", "max_calls", "max_iter") + (("remove",) if batched else ("batch",))
        run_kwargs.update({k: v for k, v in kwargs.items() if k in allowed})
        if test:
            print("nested_sample() with the following kwargs: {}".format(run_kwargs))
            return None
        run = nested_sample_batched if batched else nested_sample
        run_kwargs["propose"] = self._device_proposer(lo, hi, seed)
        transform = getattr(self, "_fit_transform", None) or getattr(self, "mnest_transform", None)
        if transform is not None:
            run_kwargs["transform"] = transform                # the cube -> parameter map is not the plain box
        res = run(lambda th: self.lnpost(np.ascontiguousarray(th)), lo, hi, **run_kwargs)
        self._nested = res
        self._samples = None
        self._fit_kind = "nested"
        if verbose:
            print("logZ = %.3f +/- %.3f  (%d iterations, %d lnpost evaluations, efficiency %.3f)"
                  % (res.logz, res.logz_err, res.niter, res.ncall, res.efficiency))
        if basename is not None:                     # MultiNest's equal-weight posterior file: params..., loglike
            import os
            folder = os.path.dirname(os.path.abspath(basename))
            os.makedirs(folder, exist_ok=True)
            x, ll = res.equal_weight_samples(rng=np.random.default_rng(seed))
            np.savetxt("{}post_equal_weights.dat".format(basename), np.column_stack([x, ll]))
        return res

    use_emcee = False

    def fit(self, **kwargs):
        """``fit_mcmc`` if the model was built with ``use_emcee=True``, else ``fit_multinest`` (reference:
        starmodel.py:667-671)."""
        if self.use_emcee:
            return self.fit_mcmc(**kwargs)
        return self.fit_multinest(**kwargs)

    def _device_proposer(self, lo, hi, seed):
        """Proposals of the nested sampler on the device: uniform draws in the bounding ellipsoid (or the unit
        cube), the flat-box transform, lnpost and the likelihood-threshold test all stay in HBM; only the
        accepted points travel back."""
        import torch
        device = torch.device("cuda", dev.current_device())
        lo_t = torch.as_tensor(lo, dtype=torch.float64, device=device)
        span_t = torch.as_tensor(np.asarray(hi) - np.asarray(lo), dtype=torch.float64, device=device)
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed) + 0x5EED)
        d = len(lo)

        def propose(mean, A, want, threshold):
            if mean is None:
                u = torch.rand(want, d, generator=gen, device=device, dtype=torch.float64)
            else:
                z = torch.randn(want, d, generator=gen, device=device, dtype=torch.float64)
                r = torch.rand(want, generator=gen, device=device, dtype=torch.float64) ** (1.0 / d)
                z = z * (r / z.norm(dim=1))[:, None]
                u = torch.as_tensor(mean, device=device) + z @ torch.as_tensor(A, device=device).T
                u = u[((u >= 0.0) & (u <= 1.0)).all(dim=1)]
            if u.shape[0] == 0:
                return np.empty((0, d)), np.empty(0), want
            ll = self.lnpost(self._cube_to_pars_device(u, lo_t, span_t))
            ok = torch.isfinite(ll) & (ll > threshold)
            return u[ok].cpu().numpy(), ll[ok].cpu().numpy(), want

        return propose

    def _cube_to_pars_device(self, u, lo_t, span_t):
        """mnest_prior on a CUDA batch; the flat box for models whose ``mnest_prior`` is one (BasicStarModel,
        reference starmodel.py:1637-1640)."""
        return lo_t + u * span_t

    @property
    def evidence(self):
        """(log evidence, its error) of the last nested fit (reference starmodel.py:813-819)."""
        if getattr(self, "_nested", None) is None:
            if getattr(self, "_loaded_evidence", None) is not None:
                return self._loaded_evidence
            raise AttributeError("fit_multinest must be run to access the evidence")
        return (self._nested.logz, self._nested.logz_err)

    def _nested_frame(self):
        import pandas as pd
        x, ll = self._nested.equal_weight_samples(rng=np.random.default_rng(0))
        df = pd.DataFrame(x, columns=list(self.param_names))
        df["lnprob"] = ll
        return df


class BasicStarModel(_NestedFitMixin, _ConvenienceMixin, _HostPriorMixin):
    def __init__(self, ic, eep_bounds=None, name="", directory=".", N=1, maxAV=None, max_distance=None,
                 halo_fraction=None, ra=None, dec=None, obs=None, use_emcee=False, **kwargs):
        self._ic = ic
        self.eep_bounds = tuple(eep_bounds) if eep_bounds is not None else tuple(ic.eep_bounds)
        self.name = str(name)
        self._directory = str(directory)
        self.use_emcee = bool(use_emcee)
        self.ra, self.dec = ra, dec
        if N not in (1, 2, 3):
            raise ValueError("N must be 1, 2 or 3")
        if N > 1 and ic.eep_replaces == "age":
            raise ValueError("Can only fit mulitple stars with IsochroneInterpolator!")
        self.N = N
        if ic.eep_replaces == "age":
            self.mass_index, self.feh_index, self.distance_index, self.AV_index = 0, 2, 3, 4
        else:
            self.age_index = N
            self.feh_index, self.distance_index, self.AV_index = N + 1, N + 2, N + 3

        self.kwargs = {}
        for k, v in kwargs.items():
            try:
                val, unc = v
                if not (np.isnan(val) or np.isnan(unc)):
                    self.kwargs[k] = (np.float64(val), np.float64(unc))
            except TypeError:
                logger.warning("kwarg {}={} ignored!".format(k, v))

        self._priors = {"mass": ChabrierPrior(), "feh": FehPrior(), "age": AgePrior(),
                        "distance": DistancePrior(), "AV": AVPrior()}
        self._priors["eep"] = EEPPrior(ic, self._priors[ic.eep_replaces], bounds=eep_bounds, owner=self)
        self._bounds = {"mass": None, "feh": None, "age": None,
                        "distance": self._priors["distance"].bounds, "AV": self._priors["AV"].bounds,
                        "eep": self._priors["eep"].bounds}
        for par in ("mass", "feh", "age"):   # snap to the table's limits (starmodel.py:1459-1460)
            self.bounds(par)
        if maxAV is not None:
            self.set_bounds(AV=(0, maxAV))
        if max_distance is not None:
            self.set_bounds(distance=(0, max_distance))
        elif "parallax" in kwargs:
            value, unc = kwargs["parallax"]
            if value > 0:
                self.set_bounds(distance=(0, 1.0 / value * 2000))
            elif value < 0:
                self.set_bounds(distance=(0, 1.0 / np.abs(unc) * 2000))
        if halo_fraction is not None:
            self._priors["feh"] = FehPrior(halo_fraction=halo_fraction)
        self._handles, self._handle_ic, self._handle_state = {}, {}, {}

    # -- star.ini files (reference: StarModel.from_ini, starmodel.py:248-436; write_ini, 1485-1499) ------
    @staticmethod
    def ini_keywords(path):
        """The constructor keywords a ``star.ini`` file describes (measurements, ``ra`` / ``dec``, ``N``, ``maxAV`` ...)."""
        from . import ini
        scalars, sections = ini.read_ini(path)
        kw = {}
        for k, v in scalars.items():
            kw[k] = ini.parse_value(v)
        for r in ini.observation_rows(sections):
            if r["relative"] or r["separation"] != 0.0:
                raise ValueError("%s describes resolved companions ([%s]); use TreeStarModel.from_ini"
                                 % (path, r["name"]))
            kw[r["band"]] = (r["mag"], r["e_mag"])
        ra = kw.pop("RA", kw.pop("ra", None))
        dec = kw.pop("dec", kw.pop("Dec", None))
        if ra is not None:
            kw["ra"] = ra
        if dec is not None:
            kw["dec"] = dec
        if "N" in kw:
            kw["N"] = int(kw["N"])
        return kw

    @classmethod
    def from_ini(cls, ic, folder=".", ini_file="star.ini", **kwargs):
        """A model from a ``star.ini`` file: every ``key = value, uncertainty`` line becomes a measurement
        keyword.  Sections of plain, unresolved photometry (``[twomass]`` holding only bands) are read as
        keywords too; sections describing resolved companions need :class:`TreeStarModel`."""
        import os
        path = ini_file if os.path.isabs(ini_file) else os.path.join(folder, ini_file)
        kw = cls.ini_keywords(path)
        kw.update(kwargs)
        if kw.get("N") is None:
            kw.pop("N", None)
        kw.setdefault("name", os.path.basename(os.path.abspath(folder)))
        return cls(ic, directory=os.path.abspath(folder), **kw)

    def write_ini(self, root="."):
        """``<root>/<name>/star.ini`` holding the measurements (and ra/dec), readable by :meth:`from_ini`."""
        import os
        from . import ini
        path = os.path.join(root, self.name)
        os.makedirs(path, exist_ok=True)
        scalars = {}
        if self.ra is not None and self.dec is not None:
            scalars["ra"], scalars["dec"] = self.ra, self.dec
        for k, v in self.kwargs.items():
            scalars[k] = (float(v[0]), float(v[1]))
        ini.write_ini(os.path.join(path, "star.ini"), scalars)
        return os.path.join(path, "star.ini")

    # -- description ------------------------------------------------------------------------
    @property
    def ic(self):
        return self._ic

    @property
    def labelstring(self):
        return {1: "single", 2: "binary", 3: "triple"}[self.N]

    @property
    def param_names(self):
        base = tuple(self.ic.param_names)
        if self.N == 1:
            return base
        return tuple(["eep_%d" % i for i in range(self.N)] + list(base[1:]))

    @property
    def n_params(self):
        return len(self.param_names)

    @property
    def bands(self):
        return [k for k in self.kwargs if k in self.ic.bc_grid.bands]

    @property
    def props(self):
        return [k for k in self.kwargs if k in _NOT_A_BAND]

    @property
    def mags(self):
        return {b: self.kwargs[b][0] for b in self.bands}

    param_description = property(lambda self: self.param_names)

    def prior_transform(self, cube):
        """Unit cube -> parameters, out of place (reference: StarModel.prior_transform, starmodel.py:615-627)."""
        cube = np.asarray(cube, dtype=float)
        lo = np.array([self.bounds(p)[0] for p in self.param_names])
        hi = np.array([self.bounds(p)[1] for p in self.param_names])
        return (hi - lo) * cube + lo

    @property
    def spec_props(self):
        return [self.kwargs.get(k, (np.nan, np.nan)) for k in ("Teff", "logg", "feh")]

    def bounds(self, prop):
        if prop in ("eep_0", "eep_1", "eep_2"):
            prop = "eep"
        if self._bounds[prop] is None:
            if prop not in ("mass", "feh", "age"):
                raise ValueError("Unknown property {}".format(prop))
            lo, hi = self.ic.model_grid.get_limits(prop)
            self._bounds[prop] = (lo, hi)
            self._priors[prop].bounds = (lo, hi)
            self._dirty()
        return self._bounds[prop]

    def set_bounds(self, **kwargs):
        for k, v in kwargs.items():
            if len(v) != 2:
                raise ValueError("Must provide (min, max)")
            self._bounds[k] = tuple(v)
            self._priors[k].bounds = tuple(v)
        self._dirty()

    def set_prior(self, **kwargs):
        """reference: StarModel.set_prior (starmodel.py:629-632).  ``eep=`` takes another model's EEP prior
        object (tests/test_likelihood.py:19-20 shares one between two models)."""
        for prop, prior in kwargs.items():
            if prop == "eep":
                if not isinstance(prior, EEPPrior):
                    raise NotImplementedError("the EEP prior must be an EEPPrior (orig_prior x d orig / d EEP)")
                prior._owners.add(self)
            elif is_host_prior(prior):
                check_host_prior(prior, prop)       # evaluated on the host (_HostPriorMixin)
            self._priors[prop] = prior
            self._bounds[prop] = tuple(prior.bounds)
        self._dirty()

    def _host_columns(self, name):
        names = self.param_names
        return [names.index(name)] if name in names else []     # (the parameter EEP replaces has no column)

    def model_desc(self) -> _cabi.IsoModelDesc:
        """Pack observations + prior constants into the C-ABI descriptor (host only)."""
        d = _cabi.IsoModelDesc()
        d.n_stars = self.N
        bands = self.bands
        if len(bands) > _cabi.ISO_MAX_BANDS:
            raise ValueError("at most %d bands" % _cabi.ISO_MAX_BANDS)
        d.n_bands = len(bands)
        ci = self.ic.bc_grid.interp.column_index
        for j, b in enumerate(bands):
            d.bc_cols[j] = ci[b]
            d.mag_val[j], d.mag_unc[j] = self.kwargs[b]
        for j, (val, unc) in enumerate(self.spec_props):
            d.spec_val[j], d.spec_unc[j] = val, unc
        if "parallax" in self.kwargs:
            d.has_parallax = 1
            d.plx_val, d.plx_unc = self.kwargs["parallax"]
        if "nu_max" in self.kwargs:
            if -1 in self.ic._astero_cols:
                raise ValueError("model table has no nu_max/delta_nu columns")
            d.has_numax = 1
            d.numax_val, d.numax_unc = self.kwargs["nu_max"]
            if "delta_nu" in self.kwargs:
                d.has_dnu = 1
                # the reference passes the *value* as the uncertainty (starmodel.py:1612)
                d.dnu_val = self.kwargs["delta_nu"][0]
                d.dnu_unc = self.kwargs["delta_nu"][0]
        for name in ("mass", "age", "feh", "distance", "AV"):
            # the parameter EEP replaces only enters through the EEP term, with the EEP prior's own object
            pr = self._priors["eep"].orig_prior if name == self.ic.eep_replaces else self._priors[name]
            setattr(d, "prior_" + name, self._packed_prior(pr))
        d.eep_lo, d.eep_hi = self._priors["eep"].bounds
        for j, par in enumerate(self.param_names):
            d.bound_lo[j], d.bound_hi[j] = self.bounds(par)
        return d

    # -- device -----------------------------------------------------------------------------
    def _dirty(self):
        for h in getattr(self, "_handles", {}).values():
            _cabi.lib().iso_model_destroy(h)
        self._handles = {}
        self._handle_ic = {}
        self._handle_state = {}
        self._scalar_cache = None

    def _scalar_call(self, p, which):
        """lnpost / lnprior / lnlike of ONE host row as a float: the per-point callback of emcee / MultiNest
        (reference starmodel.py:797,952,966).  Everything a call needs besides the numbers - the model handle, a
        parameter buffer, an output buffer, their addresses, the C entry point - is kept between calls and revalidated
        by two integer comparisons (no prior object anywhere was mutated since; the interpolator was not rebound), so the
        wrapper adds about a microsecond to the C call (whose resident mailbox wave answers without a launch)."""
        from . import priors as _p
        from .interp import TABLE_EPOCH
        # the cache (and with it the parameter / output buffers) is per THREAD: ctypes drops the GIL inside the C call, so
        # two threads sharing one pair of buffers would overwrite each other's rows (threaded emcee, a pool around
        # mnest_loglike); the C side serialises callers of one model on its own mutex
        tls = self.__dict__.get("_scalar_tls")
        if tls is None:
            tls = self.__dict__.setdefault("_scalar_tls", threading.local())
        c = getattr(tls, "c", None)
        if (c is None or c[0] != _p.EPOCH[0] or c[1] != self.ic._generation or c[8] != TABLE_EPOCH[0]
                or c[9] is not self._scalar_cache):
            device = dev.current_device()
            h = self.handle(device)                      # the full check (prior objects' versions, interpolator, tables)
            if self._scalar_cache is None:
                self._scalar_cache = object()            # token: _dirty() / a rebuilt handle drops every thread's cache
            buf, out = np.empty(self.n_params), np.empty(3)
            c = tls.c = (_p.EPOCH[0], self.ic._generation, h, buf, out, buf.ctypes.data,
                         tuple(out.ctypes.data + 8 * k for k in range(3)), _cabi.lib().iso_lnpost_host,
                         TABLE_EPOCH[0], self._scalar_cache, self._host_terms())
        c[3][:] = p                                      # (a row of the wrong length raises here)
        a = c[6]
        if c[10]:                                        # priors evaluated on the host (_HostPriorMixin): all parts, then add
            rc = c[7](c[2], c[5], 1, a[0], a[1], a[2])
            if rc:
                _cabi.check(rc)
            hp = self._host_lnprior(c[10], c[3][None, :])
            parts = (float(self._host_combine(c[4][0:1], hp)[0]), float(self._host_combine(c[4][1:2], hp)[0]), float(c[4][2]))
            return parts if which is None else parts[which]
        if which is None:                                # all three parts in one call: (lnpost, lnprior, lnlike)
            rc = c[7](c[2], c[5], 1, a[0], a[1], a[2])
            if rc:
                _cabi.check(rc)
            return float(c[4][0]), float(c[4][1]), float(c[4][2])
        rc = c[7](c[2], c[5], 1, a[0] if which == 0 else None, a[1] if which == 1 else None, a[2] if which == 2 else None)
        if rc:
            _cabi.check(rc)
        return float(c[4][which])

    def handle(self, device=None):
        if device is None:
            device = dev.current_device()
        ich = self.ic.handle(device)
        h = self._handles.get(device)
        state = _prior_state(self._priors)
        if h is not None and (self._handle_ic.get(device) != self.ic._generation or self._handle_state.get(device) != state):
            _cabi.lib().iso_model_destroy(h)   # the interpolator was rebound / a shared prior object changed: rebuild
            self._handles.pop(device, None)
            self._scalar_cache = None
            h = None
        if h is None:
            if -1 in self.ic._prior_cols:
                raise ValueError("model table lacks the (%s, d%s_deep) columns the EEP prior needs"
                                 % (self.ic.eep_replaces, "t" if self.ic.eep_replaces == "age" else "m"))
            desc = self.model_desc()
            h = C.c_void_p()
            _cabi.check(_cabi.lib().iso_model_create(ich, C.byref(desc), C.byref(h)))
            self._handles[device] = h
            self._handle_ic[device] = self.ic._generation
            self._handle_state[device] = _prior_state(self._priors)       # packing may have snapped bounds
        return h

    def kernel_path(self, device=None):
        """'generic' | 'fused-compact' | 'fused-packed': the kernel family that evaluates this model on `device`."""
        return ("generic", "fused-compact", "fused-packed")[_cabi.lib().iso_model_kernel_path(self.handle(device))]

    def __del__(self):
        try:
            self._dirty()
        except Exception:
            pass

    def evaluate_device(self, pars, soa=False, parts=False):
        """pars: CUDA float64 tensor, [N, n_params] (or [n_params, N] if soa).
        Returns lnpost[N] or (lnpost, lnprior, lnlike)."""
        if pars.dim() != 2 or pars.dtype.itemsize != 8 or not pars.dtype.is_floating_point:
            raise ValueError("pars must be a 2-D float64 tensor")
        npar = self.n_params
        if soa:
            if pars.shape[0] != npar:
                raise ValueError("expected [%d, N]" % npar)
            n, stride_n, stride_p = pars.shape[1], 1, pars.shape[1]
        else:
            if pars.shape[1] != npar:
                raise ValueError("expected [N, %d]" % npar)
            n, stride_n, stride_p = pars.shape[0], npar, 1
        device = pars.device.index
        pars = pars.contiguous()
        post = dev.empty_f64((n,), device)
        prior = dev.empty_f64((n,), device) if parts else None
        like = dev.empty_f64((n,), device) if parts else None
        if n:
            _cabi.check(_cabi.lib().iso_lnpost(self.handle(device), dev.ptr(pars), stride_n, stride_p, n,
                                               dev.ptr(post), dev.ptr(prior), dev.ptr(like),
                                               dev.stream_ptr(device)))
            terms = self._host_terms()
            if terms:
                post, prior = self._host_adjust_tensors(terms, pars.t() if soa else pars, (post, prior))
        return (post, prior, like) if parts else post

    def _evaluate(self, p, which, soa=False):
        tp = type(p)
        if ((tp is list or tp is tuple or (tp is np.ndarray and p.ndim == 1)) and len(p) == self.n_params
                and not isinstance(p[0], (list, tuple, np.ndarray))):     # (a list of n_params ROWS is a batch)
            return self._scalar_call(p, which)
        if dev.is_tensor(p) and p.is_cuda:
            import torch
            single = p.dim() == 1
            pp = p.double()[None, :] if single else p.double()
            out = self.evaluate_device(pp, soa=soa and not single, parts=which != 0)
            out = out if which == 0 else out[which]
            return out[0] if single else out
        arr = np.ascontiguousarray(p, dtype=np.float64)
        single = arr.ndim == 1
        a2 = arr[None, :] if single else arr
        device = dev.current_device()
        if not (soa and not single):
            # host arrays: one C call.  Sampler-callback sizes go through the pinned, device-mapped staging buffer (one
            # launch, completion flag), large batches through the chunked upload / download pipeline of iso_lnpost_host
            if a2.shape[1] != self.n_params:
                raise ValueError("expected [N, %d]" % self.n_params)
            n = a2.shape[0]
            out = np.empty(n)
            ptrs = [None, None, None]
            ptrs[which] = out.ctypes.data
            rc = _cabi.lib().iso_lnpost_host(self.handle(device), a2.ctypes.data, n, *ptrs)
            if rc:
                _cabi.check(rc)
            if which != 2:
                terms = self._host_terms()
                if terms:
                    out = self._host_combine(out, self._host_lnprior(terms, a2))
            return float(out[0]) if single else out
        out = self.evaluate_device(dev.to_device_f64(a2, device), soa=soa and not single, parts=which != 0)
        out = (out if which == 0 else out[which]).cpu().numpy()
        return float(out[0]) if single else out

    def lnpost(self, p, soa=False):
        return self._evaluate(p, 0, soa)

    def lnprior(self, p, soa=False):
        return self._evaluate(p, 1, soa)

    def lnlike(self, p, soa=False):
        return self._evaluate(p, 2, soa)

    # -- nested-sampling style API (reference: starmodel.py:1637-1645) -----------------------
    def mnest_prior(self, cube, ndim=None, nparams=None):
        """Unit cube -> parameter bounds, in place.  ``cube`` may be a 1-D sequence (reference
        behaviour, handled on the host: it is 5-7 multiply-adds) or a CUDA tensor [N, n_params]
        (transformed by the device kernel)."""
        if dev.is_tensor(cube) and cube.is_cuda:
            c2 = cube if cube.dim() == 2 else cube[None, :]
            if not c2.is_contiguous() or c2.dtype.itemsize != 8:
                raise ValueError("cube must be a contiguous float64 tensor")
            n = c2.shape[0]
            _cabi.check(_cabi.lib().iso_unit_cube(self.handle(c2.device.index), dev.ptr(c2), self.n_params, 1, n,
                                                  dev.stream_ptr(c2.device.index)))
            return cube
        for i, par in enumerate(self.param_names):
            lo, hi = self.bounds(par)
            cube[i] = (hi - lo) * cube[i] + lo
        return cube

    def mnest_loglike(self, cube, ndim=None, nparams=None):
        return self.lnpost(cube)

    # -- start points -----------------------------------------------------------------------
    def sample_from_prior(self, n, values=False, require_valid=True, rng=None, max_tries=50):
        """Draws from the priors as a DataFrame (``values=True``: a plain [n, n_params] array), reference
        starmodel.py:1716-1748: the non-EEP parameters from their prior objects, every EEP by weighted
        resampling of uniform integer EEPs with weight orig_prior(orig(eep)) x d orig / d EEP
        (``EEP_prior.sample``, priors.py:431-463; the weights come from one batched ``interp_value``); rows
        with a non-finite lnpost are redrawn (``require_valid``; one batched lnpost per round)."""
        import pandas as pd
        rng = rng or np.random.default_rng()
        names = list(self.param_names)
        if n == 0:
            return np.empty((0, len(names))) if values else pd.DataFrame(columns=names)
        eep_names = [nm for nm in names if nm.startswith("eep")]
        orig_par = self.ic.eep_replaces
        deriv = "dt_deep" if orig_par == "age" else "dm_deep"

        def draw(m):
            cols = {nm: _draw(self._priors[nm], m, rng) for nm in names if nm not in eep_names}
            lo, hi = self._priors["eep"].bounds
            for nm in eep_names:
                cand = rng.integers(int(np.ceil(lo)), max(int(np.floor(hi)), int(np.ceil(lo)) + 1), size=m).astype(float)
                pars = [cols["mass"], cand, cols["feh"]] if orig_par == "age" else [cand, cols["age"], cols["feh"]]
                v = np.atleast_2d(self.ic.interp_value(pars, [deriv, orig_par]))
                op = self._priors["eep"].orig_prior
                fin = np.isfinite(v[:, 1])
                w = np.where(fin, op.pdf_array(np.where(fin, v[:, 1], 0.0)), 0.0) * v[:, 0]
                w = np.where(np.isfinite(w) & (w > 0), w, 0.0)
                cols[nm] = cand[rng.choice(m, size=m, p=w / w.sum())] if w.sum() > 0 else cand
            x = np.column_stack([cols[nm] for nm in names])
            if self.N > 1:   # eep_0 >= eep_1 >= eep_2
                x[:, :self.N] = -np.sort(-x[:, :self.N], axis=1)
            return x

        out = draw(n)
        if require_valid:
            for _ in range(max_tries):
                bad = ~np.isfinite(self.lnpost(out))
                if not bad.any():
                    break
                out[bad] = draw(max(int(bad.sum()), 2))[: int(bad.sum())]
        return out if values else pd.DataFrame(out, columns=names)


    # -- MCMC (reference: fit_mcmc_old, starmodel.py:889-972; emcee replaced by the on-device
    #    stretch-move sampler in isochrones_amd/sampler.py) ------------------------------------
    def emcee_p0(self, nwalkers, rng=None):
        return self.sample_from_prior(nwalkers, values=True, rng=rng, require_valid=True)

    def fit_mcmc(self, nwalkers=300, nburn=200, niter=100, p0=None, initial_burn=None, ninitial=50, seed=None,
                 fused=None, **kwargs):
        """Burn in, reset, sample (reference: fit_mcmc_old, starmodel.py:889-972).  ``initial_burn``: run
        ``ninitial`` iterations from the prior draws first and restart every walker in a 0.1 % ball around the best
        point found (the reference's re-initialisation).  ``fused`` selects the single-kernel sampler (default: used
        whenever the model is on the corner-packed fast path, else the framework-op sampler, which evaluates lnpost
        through the same HIP kernels)."""
        import torch
        from .sampler import EnsembleSampler, FusedEnsembleSampler
        rng = np.random.default_rng(seed)
        npars = self.n_params
        device = torch.device("cuda", dev.current_device())

        host_priors = bool(self._host_terms())
        if host_priors:
            if fused:
                raise ValueError("a model with priors evaluated on the host cannot run the resident sampler (fused=True)")
            logger.warning("priors %s are evaluated on the host: fit by the framework-op sampler",
                           [type(t[1]).__name__ for t in self._host_terms()])

        def make_sampler():
            if (fused is None or fused) and not host_priors:
                try:
                    return FusedEnsembleSampler(self, nwalkers, seed=int(rng.integers(2 ** 62)))
                except _cabi.IsoError as e:
                    # (as _run_mcmc_fit: only "no resident kernel for this shape" changes samplers; a HIP / memory failure is an error)
                    if fused or e.rc not in (None, _cabi.ERR_INVALID):
                        raise
                    import warnings
                    warnings.warn("fit_mcmc: no device-resident sampler for this model shape (%s); using the framework-op "
                                  "EnsembleSampler, whose random numbers differ" % e, RuntimeWarning, stacklevel=3)
            return EnsembleSampler(nwalkers, npars, self.lnpost, seed=int(rng.integers(2 ** 62)), device=device)

        if p0 is None:
            p0 = self.emcee_p0(nwalkers, rng=rng)
            if initial_burn:
                first = make_sampler()
                first.run_mcmc(p0, ninitial)
                flat = first.flatlnprobability
                best = first.flatchain[int(torch.argmax(flat))].cpu().numpy()
                ball = best * (1 + rng.normal(size=p0.shape) * 0.001)
                bad = ~np.isfinite(self.lnpost(ball))
                ball[bad] = best                       # a perturbed walker that left the support restarts on the point
                p0 = ball
        else:
            p0 = rng.normal(size=(nwalkers, npars)) * 0.01 + np.asarray(p0, dtype=float)[None, :]
        sampler = make_sampler()
        pos, prob = sampler.run_mcmc(p0, nburn, store=False)
        sampler.reset()
        sampler.run_mcmc(pos, niter, lnprob0=prob)
        self._sampler = sampler
        self._samples = None
        self._fit_kind = "mcmc"
        return sampler

    fit_mcmc_old = fit_mcmc

    @property
    def sampler(self):
        if getattr(self, "_sampler", None) is None:
            raise AttributeError("MCMC must be run to access sampler")
        return self._sampler

    @property
    def samples(self):
        """Posterior samples as a DataFrame: the sampled parameters + lnprob (+ every model column
        and band magnitude of a single star via ``ic(...)``, reference starmodel.py:1653-1714)."""
        import pandas as pd
        if getattr(self, "_samples", None) is None:
            if getattr(self, "_fit_kind", "mcmc") == "nested":
                df = self._nested_frame()
                chain = df[list(self.param_names)].values
            else:
                chain = self.sampler.flatchain.cpu().numpy()
                df = pd.DataFrame(chain, columns=list(self.param_names))
                df["lnprob"] = self.sampler.flatlnprobability.cpu().numpy()
            if self.N == 1:
                derived = self.ic(*[chain[:, j] for j in range(5)])
                for c in derived.columns:
                    if c not in df.columns:
                        df[c] = derived[c].values
            self._samples = df
        return self._samples


    @property
    def derived_samples(self):
        """Derived quantities of the posterior samples (reference: ``_make_samples``, starmodel.py:1653-1707):
        one star: every model column and band magnitude from ``ic(*params)``; two / three stars: the sampled
        parameters, each component's columns with ``_0`` / ``_1`` / ``_2`` suffixes and the combined
        magnitudes; plus ``parallax``, ``distance``, ``AV``."""
        import pandas as pd
        if getattr(self, "_derived_samples", None) is not None and self._derived_for is self.samples:
            return self._derived_samples
        df = self.samples
        if self.N == 1:
            out = self.ic(*[df[c].values for c in self.param_names])
        else:
            out = df[list(self.param_names) + ["lnprob"]].copy()
            for k in range(self.N):
                comp = self.ic(*[df[c].values for c in ("eep_%d" % k, "age", "feh", "distance", "AV")])
                keep = [c for c in comp.columns if c not in ("eep", "age")]
                comp = comp[keep].rename(columns={c: "%s_%d" % (c, k) for c in keep if c not in ("distance", "AV")})
                out = pd.concat([out, comp.drop(columns=[c for c in ("distance", "AV") if c in comp.columns])], axis=1)
            for b in self.ic.bands:
                flux = sum(10 ** (-0.4 * out["%s_mag_%d" % (b, k)].values) for k in range(self.N))
                out[b + "_mag"] = -2.5 * np.log10(flux)
        out["parallax"] = 1000.0 / df["distance"].values
        out["distance"] = df["distance"].values
        out["AV"] = df["AV"].values
        self._derived_samples, self._derived_for = out, df
        return out


    # -- summaries over the samples (reference: starmodel.py:1755-1841) -----------------------------
    @property
    def physical_quantities(self):
        if self.N == 1:
            return ["mass", "radius", "age", "Teff", "logg", "feh", "distance", "AV"]
        cols = []
        for q in ("mass", "radius"):
            cols += ["%s_%d" % (q, k) for k in range(self.N)] if self.N == 3 else []
        if self.N == 2:
            cols = ["mass_0", "radius_0", "mass_1", "radius_1", "Teff_0", "Teff_1", "logg_0", "logg_1"]
        else:
            cols = sum([["mass_%d" % k, "radius_%d" % k] for k in range(3)], []) + \
                ["Teff_%d" % k for k in range(3)] + ["logg_%d" % k for k in range(3)]
        return cols + ["age", "feh", "distance", "AV"]

    @property
    def observed_quantities(self):
        cols = ["{}_mag".format(b) for b in self.bands]
        if self.N == 1:
            return cols + list(self.props)
        return cols + [p if p in self.derived_samples.columns else "{}_0".format(p) for p in self.props]

    @property
    def posterior_predictive(self):
        """Mean chi^2 per observable of the derived samples against the observations."""
        d = self.derived_samples
        chisq = 0
        for b in self.bands:
            val, unc = self.kwargs[b]
            chisq = chisq + (val - d["{}_mag".format(b)]) ** 2 / unc ** 2
        for pname, col in zip(self.props, self.observed_quantities[len(self.bands):]):
            val, unc = self.kwargs[pname]
            chisq = chisq + (val - d[col]) ** 2 / unc ** 2
        return float(np.mean(chisq)) / (len(self.bands) + len(self.props))

    @property
    def map_pars(self):
        s = self.samples
        return s.loc[s["lnprob"].idxmax(), list(self.param_names)].values.astype(float)


class SingleStarModel(BasicStarModel):
    def __init__(self, *args, **kwargs):
        kwargs["N"] = 1
        super().__init__(*args, **kwargs)


class BinaryStarModel(BasicStarModel):
    def __init__(self, *args, **kwargs):
        kwargs["N"] = 2
        super().__init__(*args, **kwargs)


class TripleStarModel(BasicStarModel):
    def __init__(self, *args, **kwargs):
        kwargs["N"] = 3
        super().__init__(*args, **kwargs)



# ==========================================================================================
# generic (observation-tree) model — "next" row f4
# ==========================================================================================
class TreeStarModel(_NestedFitMixin, _ConvenienceMixin, _HostPriorMixin):
    """The reference's generic ``StarModel`` (isochrones/starmodel.py:63-661): photometry organised
    in an :class:`~isochrones_amd.observation.ObservationTree` (resolved and blended sources,
    relative photometry, several physical systems), evaluated on the device from the flattened
    tree.  Isochrone parametrisation only, as the reference.  Parameter vector: for every system
    its EEPs (descending) followed by age, feh, distance, AV (``param_names``).

    Keyword measurements (``Teff=(v, e)``, ``logg=``, ``feh=``, ``parallax=``, ``AV=``, ``<band>=``)
    are added to the tree exactly like the reference's ``_build_obs`` / ``_add_properties``; a
    keyword of the form ``Teff_1=(v, e)`` addresses leaf ``0_1``."""

    def __init__(self, ic, obs=None, N=1, index=0, name="", eep_bounds=None, maxAV=None, max_distance=None,
                 use_emcee=False, **kwargs):
        import re
        from .observation import Observation, ObservationTree, Source
        if ic.eep_replaces != "mass":
            raise NotImplementedError("Prior not implemented for evolution track grids")
        self._ic = ic
        self.name = name
        self.use_emcee = bool(use_emcee)
        if obs is None:
            obs = ObservationTree()
            for k, v in kwargs.items():
                if k in ic.bands:
                    v = (v, np.nan) if np.size(v) != 2 else v
                    o = Observation("", k, 99)           # bogus resolution, as the reference
                    o.add_source(Source(v[0], v[1]))
                    o._set_reference()
                    obs.add_observation(o)
            obs.define_models(ic, N=N, index=index)
            add_props = True
        else:
            add_props = len(obs.model_nodes()) == 0
            if add_props:
                obs.define_models(ic, N=N, index=index)
        self.obs = obs
        if add_props:
            for k, v in kwargs.items():
                if k in ic.bands or k in ("maxAV", "max_distance"):
                    continue
                if k == "parallax":
                    obs.add_parallax(v)
                elif k == "AV":
                    obs.add_AV(v)
                elif k in ("Teff", "logg", "feh", "density"):
                    obs.add_spectroscopy(**{k: v})
                elif re.search(r"_", k):
                    m = re.search(r"^(\w+)_(\w+)$", k)
                    obs.add_spectroscopy(label="0_{}".format(m.group(2)), **{m.group(1): v})
        self._priors = {"mass": ChabrierPrior(), "feh": FehPrior(), "age": AgePrior(),
                        "distance": DistancePrior(), "AV": AVPrior()}
        self._priors["eep"] = EEPPrior(ic, self._priors["mass"], bounds=eep_bounds, owner=self)
        self._bounds = {"mass": None, "feh": None, "age": None, "distance": self._priors["distance"].bounds,
                        "AV": self._priors["AV"].bounds, "eep": self._priors["eep"].bounds}
        for par in ("feh", "age"):           # the reference snaps these to the table on first use (the generic
            self.bounds(par)                 # StarModel never asks for the mass bounds: its Chabrier prior keeps (0.1, 100))
        if maxAV is not None:
            self.set_bounds(AV=(0, maxAV))
        if max_distance is not None:
            self.set_bounds(distance=(0, max_distance))
        self._handles, self._handle_ic, self._handle_state = {}, {}, {}

    ic = property(lambda self: self._ic)

    # -- star.ini files (reference: starmodel.py:229-436) ---------------------------------------
    @staticmethod
    def get_bands(inifile):
        from . import ini
        return ini.get_bands(inifile)

    @classmethod
    def from_ini(cls, ic, folder=".", ini_file="star.ini", **kwargs):
        """A model from a ``star.ini`` file (format: the reference's docstring, starmodel.py:249-314).
        Top-level lines are measurement keywords (``Teff = 5770, 80``), ``maxAV``, ``N``, ``index`` or
        ``obsfile`` (a csv with the columns of :meth:`ObservationTree.from_df`); every ``[section]`` is one
        instrument's photometry, resolved companions carrying ``separation_<tag>`` / ``PA_<tag>`` /
        ``<band>_<tag>``.  Without sections the bands apply to all model stars together."""
        import os
        import pandas as pd
        from . import ini
        from .observation import ObservationTree
        path = ini_file if os.path.isabs(ini_file) else os.path.join(folder, ini_file)
        scalars, sections = ini.read_ini(path)
        kw = {k: ini.parse_value(v) for k, v in scalars.items()}
        for k in ("RA", "dec", "ra", "Dec"):
            kw.pop(k, None)
        obs = None
        if sections:
            rows = ini.observation_rows(sections)
            obs = ObservationTree.from_df(pd.DataFrame(rows, columns=["name", "band", "resolution", "relative",
                                                                       "separation", "pa", "mag", "e_mag"]))
        obsfile = kw.pop("obsfile", None)
        if obsfile is not None:
            obsfile = obsfile if os.path.isabs(obsfile) else os.path.join(folder, obsfile)
            obs = ObservationTree.from_df(pd.read_csv(obsfile))
        for k in ("N", "index"):
            if k in kw:
                kw[k] = [int(x) for x in kw[k]] if isinstance(kw[k], list) else int(kw[k])
        for k, v in list(kw.items()):
            if isinstance(v, list) and k not in ("N", "index"):
                kw[k] = tuple(v)
        kw.update(kwargs)
        if kw.get("N") is None:
            kw.pop("N", None)
        kw.setdefault("name", os.path.basename(os.path.abspath(folder)))
        new = cls(ic, obs=obs, **kw)
        new._directory = os.path.abspath(folder)
        return new

    def print_ascii(self, fout=None, p=None):
        return self.obs.print_ascii(fout=fout, p=p)

    def convert_pars_to_eep(self, pars):
        """A mass-based parameter vector (pre-2.0 layout: masses where the EEPs go) with every mass replaced
        by the EEP that reaches it at that age and [Fe/H] (reference: starmodel.py:443-454)."""
        pardict = self.obs.p2pardict(list(pars))
        for star, sp in pardict.items():
            sp[0] = float(self.ic.get_eep(sp[0], sp[1], sp[2], accurate=True))
        return self.obs.pardict2p(pardict)

    @property
    def param_names(self):
        return tuple(self.obs.param_description)

    param_description = param_names

    @property
    def labelstring(self):
        """'single' / 'binary' / 'triple' for one system, joined by '-' for several (reference: starmodel.py:163-177)."""
        names = {1: "single", 2: "binary", 3: "triple"}
        N = self.obs.Nstars
        return "-".join(names.get(N[s], "{}stars".format(N[s])) for s in self.obs.systems)

    @property
    def mags(self):
        return {n.observation.band: n.source.mag for n in self.obs.obs_nodes()}

    @property
    def props(self):
        """Measured properties beyond Teff / logg / feh (reference: starmodel.py:201-206)."""
        found = {k for v in self.obs.spectroscopy.values() for k in v}
        return sorted(found - {"Teff", "logg", "feh"})

    def _slots(self, reference_order):
        """(system, number of stars) in the order the parameter slots are walked.  The parameter vector itself is laid
        out by ascending system index (``param_names``, what ``lnpost`` reads: observation.py:1116-1130, :1146-1154).
        The reference's ``prior_transform`` / ``mnest_prior`` walk ``obs.Nstars.items()`` instead (starmodel.py:618,
        :646) - the order in which the tree first meets each system - which is a different slot layout whenever the
        systems are met out of order AND hold different numbers of stars; there the reference scales slots with the
        bounds of other parameters.  ``reference_order=True`` reproduces that walk."""
        N = self.obs.Nstars
        return list(N.items()) if reference_order else [(s, N[s]) for s in self.obs.systems]

    def _box(self, cube, reference_order, sort_eeps):
        cube = np.asarray(cube, dtype=float)
        pars = cube * 0
        i = 0
        lo_e, hi_e = self._bounds["eep"]
        for _, n in self._slots(reference_order):
            pars[..., i:i + n] = (hi_e - lo_e) * cube[..., i:i + n] + lo_e
            if sort_eeps and n > 1:
                pars[..., i:i + n] = -np.sort(-pars[..., i:i + n], axis=-1)
            for j, par in enumerate(("age", "feh", "distance", "AV")):
                lo, hi = self.bounds(par)
                pars[..., i + n + j] = (hi - lo) * cube[..., i + n + j] + lo
            i += 4 + n
        return pars

    def mnest_transform(self, cube):
        """Unit cube -> parameters exactly as the reference's ``StarModel.mnest_prior`` maps them (starmodel.py:644-656):
        the flat box, every system's EEPs put in descending order, slots walked as the reference walks them (see
        :meth:`_slots`).  ``cube`` [..., n_params]; out of place."""
        return self._box(cube, True, True)

    def _fit_transform(self, cube):
        """The same map on the parameter vector's own layout: what this build's nested sampler uses, so that a fit also
        works for the trees on which the reference's slot walk goes astray (identical to :meth:`mnest_transform`
        everywhere else)."""
        return self._box(cube, False, True)

    def _cube_to_pars_device(self, u, lo_t, span_t):
        import torch
        pars = lo_t + u * span_t
        i = 0
        for s in self.obs.systems:
            n = self.obs.Nstars[s]
            if n > 1:
                pars[:, i:i + n] = torch.sort(pars[:, i:i + n], dim=1, descending=True).values
            i += 4 + n
        return pars

    def mnest_prior(self, cube, ndim=None, nparams=None):
        """Unit cube -> parameters, in place: pymultinest's prior callback (reference: starmodel.py:644-656)."""
        out = self.mnest_transform(np.array([cube[i] for i in range(self.n_params)], dtype=float))
        for i in range(self.n_params):
            cube[i] = out[i]

    def mnest_loglike(self, cube, ndim=None, nparams=None):
        return self.lnpost(np.array([cube[i] for i in range(self.n_params)], dtype=float))

    @property
    def n_params(self):
        return len(self.param_names)

    @property
    def bands(self):
        return [b for b in self.ic.bc_grid.bands if b in set(self.obs.bands)]

    def bounds(self, prop):
        if prop not in self._bounds:              # parameter names carry system / star suffixes: eep_0_1, age_0, ...
            prop = prop.split("_")[0]
        if self._bounds[prop] is None:
            if prop not in ("mass", "feh", "age"):
                raise ValueError("Unknown property {}".format(prop))
            lo, hi = self.ic.model_grid.get_limits(prop)
            self._bounds[prop] = (lo, hi)
            self._priors[prop].bounds = (lo, hi)
            self._dirty()
        return self._bounds[prop]

    def set_prior(self, **kwargs):
        """reference: StarModel.set_prior (starmodel.py:629-632); as there, the EEP term keeps the mass prior it
        was built with (assign ``_priors["eep"].orig_prior`` to change it)."""
        for prop, prior in kwargs.items():
            if prop == "eep":
                if not isinstance(prior, EEPPrior):
                    raise NotImplementedError("the EEP prior must be an EEPPrior (orig_prior x d orig / d EEP)")
                prior._owners.add(self)
            elif is_host_prior(prior):
                if prop == "mass":
                    raise NotImplementedError("the mass prior of a tree model only enters through the EEP term")
                check_host_prior(prior, prop)       # evaluated on the host (_HostPriorMixin)
            self._priors[prop] = prior
            self._bounds[prop] = tuple(prior.bounds)
        self._dirty()

    def _host_columns(self, name):
        """Every system carries its own age / feh / distance / AV behind its EEPs (parameter layout: per system the
        N_s EEPs, then age, feh, distance, AV)."""
        if name == "mass":
            return []
        k = ("age", "feh", "distance", "AV").index(name)
        cols, at = [], 0
        for n_s in self.obs.program(self.bands)["n_stars"]:
            cols.append(at + n_s + k)
            at += n_s + 4
        return cols

    def set_bounds(self, **kwargs):
        for k, v in kwargs.items():
            if len(v) != 2:
                raise ValueError("Must provide (min, max)")
            self._bounds[k] = tuple(v)
            self._priors[k].bounds = tuple(v)
        self._dirty()

    def tree_desc(self) -> _cabi.IsoTreeDesc:
        """Pack the flattened tree + prior constants into the C-ABI record (host only)."""
        bands = self.bands
        prog = self.obs.program(bands)
        d = _cabi.IsoTreeDesc()
        if len(prog["systems"]) > _cabi.TREE_MAX_SYSTEMS or len(prog["leaf_system"]) > _cabi.TREE_MAX_LEAVES:
            raise ValueError("at most %d systems / %d model stars" % (_cabi.TREE_MAX_SYSTEMS, _cabi.TREE_MAX_LEAVES))
        if len(bands) > _cabi.TREE_MAX_BANDS or len(prog["terms"]) > _cabi.TREE_MAX_TERMS:
            raise ValueError("at most %d bands / %d observation nodes" % (_cabi.TREE_MAX_BANDS, _cabi.TREE_MAX_TERMS))
        if len(prog["spec"]) > _cabi.TREE_MAX_SPEC or len(prog["limits"]) > _cabi.TREE_MAX_SPEC:
            raise ValueError("too many spectroscopic constraints")
        d.n_systems, d.n_leaves, d.n_bands = len(prog["systems"]), len(prog["leaf_system"]), len(bands)
        d.n_terms, d.n_spec, d.n_limits = len(prog["terms"]), len(prog["spec"]), len(prog["limits"])
        for i, n in enumerate(prog["n_stars"]):
            d.n_stars[i] = n
        for i, (s, j) in enumerate(zip(prog["leaf_system"], prog["leaf_slot"])):
            d.leaf_system[i], d.leaf_slot[i] = s, j
        ci = self.ic.bc_grid.interp.column_index
        for i, b in enumerate(bands):
            d.bc_cols[i] = ci[b]
        for i, t in enumerate(prog["terms"]):
            e = d.terms[i]
            e.band, e.relative, e.mask, e.ref_mask = t["band"], t["relative"], t["mask"], t["ref_mask"]
            e.mag, e.unc, e.ref_mag = t["mag"], t["unc"], t["ref_mag"]
        for i, t in enumerate(prog["spec"]):
            d.spec[i].leaf, d.spec[i].prop, d.spec[i].a, d.spec[i].b = t["leaf"], t["prop"], t["val"], t["unc"]
        for i, t in enumerate(prog["limits"]):
            d.limits[i].leaf, d.limits[i].prop, d.limits[i].a, d.limits[i].b = t["leaf"], t["prop"], t["lo"], t["hi"]
        for s, (v, e) in prog["parallax"].items():
            d.has_plx[s], d.plx_val[s], d.plx_unc[s] = 1, v, e
        for s, (v, e) in prog["AV"].items():
            d.has_av[s], d.av_val[s], d.av_unc[s] = 1, v, e
        for j, prop in enumerate(("age", "feh", "distance", "AV")):
            d.bound_lo[j], d.bound_hi[j] = self.bounds(prop)        # also snaps feh/age priors to the table
        for name in ("mass", "age", "feh", "distance", "AV"):
            pr = self._priors["eep"].orig_prior if name == "mass" else self._priors[name]
            setattr(d, "prior_" + name, self._packed_prior(pr))
        d.eep_lo, d.eep_hi = self._priors["eep"].bounds
        return d

    def _dirty(self):
        for h in getattr(self, "_handles", {}).values():
            _cabi.lib().iso_tree_model_destroy(h)
        self._handles = {}
        self._handle_ic = {}
        self._handle_state = {}

    def handle(self, device=None):
        if device is None:
            device = dev.current_device()
        ich = self.ic.handle(device)
        h = self._handles.get(device)
        state = _prior_state(self._priors)
        if h is not None and (self._handle_ic.get(device) != self.ic._generation or self._handle_state.get(device) != state):
            _cabi.lib().iso_tree_model_destroy(h)
            self._handles.pop(device, None)
            h = None
        if h is None:
            desc = self.tree_desc()
            h = C.c_void_p()
            _cabi.check(_cabi.lib().iso_tree_model_create(ich, C.byref(desc), C.byref(h)))
            self._handles[device] = h
            self._handle_ic[device] = self.ic._generation
            self._handle_state[device] = _prior_state(self._priors)
        return h

    def __del__(self):
        try:
            self._dirty()
        except Exception:
            pass

    def evaluate_device(self, pars, parts=False):
        """pars: CUDA float64 [N, n_params] -> lnpost [N] or (lnpost, lnprior, lnlike)."""
        npar = self.n_params
        if pars.dim() != 2 or pars.shape[1] != npar or pars.dtype.itemsize != 8 or not pars.dtype.is_floating_point:
            raise ValueError("expected a float64 [N, %d] tensor" % npar)
        device = pars.device.index
        pars = pars.contiguous()
        n = pars.shape[0]
        post = dev.empty_f64((n,), device)
        prior = dev.empty_f64((n,), device) if parts else None
        like = dev.empty_f64((n,), device) if parts else None
        if n:
            _cabi.check(_cabi.lib().iso_tree_lnpost(self.handle(device), dev.ptr(pars), npar, 1, n, dev.ptr(post),
                                                    dev.ptr(prior), dev.ptr(like), dev.stream_ptr(device)))
            terms = self._host_terms()
            if terms:
                post, prior = self._host_adjust_tensors(terms, pars, (post, prior))
        return (post, prior, like) if parts else post

    def _scalar_call(self, p, which):
        """lnpost / lnprior / lnlike of ONE host row as a float - how emcee / MultiNest drive a generic StarModel
        (reference starmodel.py:538-542, 797, 952): handle, buffers and addresses are kept per thread and revalidated by
        integer comparisons (as BasicStarModel._scalar_call); the C call is answered by the model's resident mailbox wave."""
        from . import priors as _p
        from .interp import TABLE_EPOCH
        tls = self.__dict__.get("_scalar_tls")
        if tls is None:
            tls = self.__dict__.setdefault("_scalar_tls", threading.local())
        c = getattr(tls, "c", None)
        if (c is None or c[0] != _p.EPOCH[0] or c[1] != self.ic._generation or c[8] != TABLE_EPOCH[0]
                or c[9] is not self._handles.get(c[10])):
            device = dev.current_device()
            h = self.handle(device)
            buf, out = np.empty(self.n_params), np.empty(3)
            c = tls.c = (_p.EPOCH[0], self.ic._generation, h, buf, out, buf.ctypes.data,
                         tuple(out.ctypes.data + 8 * k for k in range(3)), _cabi.lib().iso_tree_lnpost_host, TABLE_EPOCH[0], h, device,
                         self._host_terms())
        c[3][:] = p
        a = c[6]
        if c[11] and which != 2:                         # priors evaluated on the host (_HostPriorMixin)
            rc = c[7](c[2], c[5], 1, a[0] if which == 0 else None, a[1] if which == 1 else None, None)
            if rc:
                _cabi.check(rc)
            return float(self._host_combine(c[4][which:which + 1], self._host_lnprior(c[11], c[3][None, :]))[0])
        rc = c[7](c[2], c[5], 1, a[0] if which == 0 else None, a[1] if which == 1 else None, a[2] if which == 2 else None)
        if rc:
            _cabi.check(rc)
        return float(c[4][which])

    def _evaluate(self, p, which):
        tp = type(p)
        if ((tp is list or tp is tuple or (tp is np.ndarray and p.ndim == 1)) and len(p) == self.n_params
                and not isinstance(p[0], (list, tuple, np.ndarray))):
            return self._scalar_call(p, which)
        if dev.is_tensor(p) and p.is_cuda:
            single = p.dim() == 1
            out = self.evaluate_device(p.double()[None, :] if single else p.double(), parts=which != 0)
            out = out if which == 0 else out[which]
            return out[0] if single else out
        arr = np.ascontiguousarray(p, dtype=np.float64)
        single = arr.ndim == 1
        a2 = arr[None, :] if single else arr
        device = dev.current_device()
        if a2.ndim == 2 and a2.shape[0] <= 65536:
            # host arrays of sampler-callback size: one C call through the context's pinned staging buffer
            if a2.shape[1] != self.n_params:
                raise ValueError("expected [N, %d]" % self.n_params)
            n = a2.shape[0]
            out = np.empty(n)
            dp = C.POINTER(C.c_double)
            ptrs = [None, None, None]
            ptrs[which] = out.ctypes.data_as(dp)
            _cabi.check(_cabi.lib().iso_tree_lnpost_host(self.handle(device), a2.ctypes.data_as(dp), n, *ptrs))
            if which != 2:
                terms = self._host_terms()
                if terms:
                    out = self._host_combine(out, self._host_lnprior(terms, a2))
            return float(out[0]) if single else out
        out = self.evaluate_device(dev.to_device_f64(a2, device), parts=which != 0)
        out = (out if which == 0 else out[which]).cpu().numpy()
        return float(out[0]) if single else out

    def lnpost(self, p):
        return self._evaluate(p, 0)

    def lnprior(self, p):
        return self._evaluate(p, 1)

    def lnlike(self, p):
        return self._evaluate(p, 2)


    def prior_transform(self, cube):
        """Unit cube -> parameters (reference: starmodel.py:615-627, its slot walk included: see :meth:`_slots`)."""
        return self._box(cube, True, False)

    # -- fits (reference: StarModel.fit_mcmc / fit_multinest, starmodel.py:717-972) --------------
    def sample_from_prior(self, n, rng=None, max_tries=200):
        """[n, n_params] uniform draws from the parameter box with a finite lnpost (EEPs of a system
        in descending order, as the reference's mnest_prior sorts them)."""
        rng = rng or np.random.default_rng()

        def draw(m):
            return self._fit_transform(rng.random((m, self.n_params)))

        out = draw(n)
        for _ in range(max_tries):
            bad = ~np.isfinite(self.lnpost(out))
            if not bad.any():
                return out
            out[bad] = draw(int(bad.sum()))
        raise RuntimeError("could not find %d starting points with a finite lnpost" % n)

    emcee_p0 = sample_from_prior

    def fit_mcmc(self, nwalkers=300, nburn=200, niter=100, p0=None, seed=None, fused=None, n_ensembles=1, **kwargs):
        """Stretch-move ensemble resident on the device: burn-in and sampling are one persistent launch each
        (``k_stretch_tree``: proposal, tree lnpost and accept step of every iteration inside the kernel, positions in
        LDS).  ``fused=False`` selects the framework-op sampler around the batch kernel (debugging aid; also what runs
        when the tree has no device-resident form - tables off the corner-packed path)."""
        self._sampler = _run_mcmc_fit(self, nwalkers, nburn, niter, p0, seed, fused, n_ensembles)
        self._samples = None
        self._fit_kind = "mcmc"
        return self._sampler

    fit_mcmc_old = fit_mcmc

    @property
    def sampler(self):
        if getattr(self, "_sampler", None) is None:
            raise AttributeError("MCMC must be run to access sampler")
        return self._sampler

    @property
    def samples(self):
        """Posterior samples (sampled parameters + lnprob) of the last fit."""
        import pandas as pd
        if getattr(self, "_samples", None) is None:
            if getattr(self, "_fit_kind", "mcmc") == "nested":
                self._samples = self._nested_frame()
            else:
                df = pd.DataFrame(self.sampler.flatchain.cpu().numpy(), columns=list(self.param_names))
                df["lnprob"] = self.sampler.flatlnprobability.cpu().numpy()
                self._samples = df
        return self._samples


class _StarModelMeta(type):
    def __instancecheck__(cls, obj):
        return isinstance(obj, (BasicStarModel, TreeStarModel))

    def __subclasscheck__(cls, sub):
        return issubclass(sub, (BasicStarModel, TreeStarModel))


class StarModel(metaclass=_StarModelMeta):
    """The reference's name.  Called, it is a factory: an :class:`ObservationTree` (or a ``star.ini`` with
    sections) gives the generic tree model, plain keyword measurements of one unresolved 1-3 star system give
    :class:`BasicStarModel` (the reference pins both to the same numbers, tests/test_likelihood.py).  The class
    methods of the reference's ``StarModel`` hang off it too (``from_ini``, ``get_bands``, ``load_hdf``)."""

    def __new__(cls, ic, obs=None, **kwargs):
        if obs is not None:
            return TreeStarModel(ic, obs=obs, **kwargs)
        return BasicStarModel(ic, **kwargs)

    from_ini = TreeStarModel.from_ini
    get_bands = staticmethod(TreeStarModel.get_bands)

    @staticmethod
    def load_hdf(filename, path="", name=None, ic=None):
        from . import persist
        return persist.load_model(None, filename, ic=ic, name=name)

    load = load_hdf


class IsoTrackModel(_NestedFitMixin):
    """The reference's experimental model that asks one star to agree with *both* grids
    (isochrones/starmodel.py:2010-2104): parameters (eep, mass, age, feh, distance, AV); the
    likelihood is the sum of the isochrone-grid likelihood at (eep, age, feh, d, AV) and the
    track-grid likelihood at (mass, eep, feh, d, AV), the parallax term counted once; the prior is
    the track model's prior plus the age prior.  Composed on the device from two fused-kernel
    evaluations (no new kernel)."""

    param_names = ("eep", "mass", "age", "feh", "distance", "AV")

    def __init__(self, iso, track, **kwargs):
        self.iso, self.track = iso, track
        no_plx = {k: v for k, v in kwargs.items() if k != "parallax"}
        self._track_model = BasicStarModel(track, **kwargs)        # owns the priors and the parallax term
        self._iso_model = BasicStarModel(iso, **no_plx)
        self._priors = self._track_model._priors
        self.kwargs = self._track_model.kwargs

    ic = property(lambda self: self.track)
    n_params = 6
    labelstring = "single"
    name = property(lambda self: self._track_model.name)
    bands = property(lambda self: self._track_model.bands)
    props = property(lambda self: self._track_model.props)
    spec_props = property(lambda self: self._track_model.spec_props)
    mags = property(lambda self: self._track_model.mags)
    param_description = param_names

    def bounds(self, prop):
        return self._track_model.bounds(prop)

    def set_bounds(self, **kwargs):
        self._track_model.set_bounds(**kwargs)
        self._iso_model.set_bounds(**kwargs)

    def set_prior(self, **kwargs):
        """Priors live in the track-grid model (mass through its EEP term, feh, distance, AV); the age prior is
        evaluated here and has to stay an :class:`AgePrior` (its bounds may change)."""
        if "age" in kwargs and not isinstance(kwargs["age"], AgePrior):
            raise NotImplementedError("IsoTrackModel evaluates the age prior in closed form: pass an AgePrior")
        self._track_model.set_prior(**kwargs)       # (a prior evaluated on the host joins through the track model's evaluate_device)

    def prior(self, prop, val, **kwargs):
        return self._priors[prop](val, **kwargs)

    def _host_terms(self):
        return self._track_model._host_terms()      # (they join lnprior through the track model's evaluations)

    def age_prior_constants(self):
        """(lo, hi, lnorm) of the closed-form age prior: ln p(age) = lnorm + age ln 10 inside [lo, hi]."""
        lo, hi = self._priors["age"].bounds
        return float(lo), float(hi), math.log(math.log(10.0) / (10.0 ** hi - 10.0 ** lo))

    def _split(self, p):
        import torch
        iso_p = torch.stack([p[:, 0], p[:, 2], p[:, 3], p[:, 4], p[:, 5]], dim=1).contiguous()
        trk_p = torch.stack([p[:, 1], p[:, 0], p[:, 3], p[:, 4], p[:, 5]], dim=1).contiguous()
        return iso_p, trk_p

    def evaluate_device(self, p):
        """p: CUDA float64 [N, 6] -> (lnpost, lnprior, lnlike) CUDA tensors."""
        import torch
        iso_p, trk_p = self._split(p)
        _, t_prior, t_like = self._track_model.evaluate_device(trk_p, parts=True)
        _, _, i_like = self._iso_model.evaluate_device(iso_p, parts=True)
        age = p[:, 2]
        lo, hi, lnorm = self.age_prior_constants()
        ln_age = lnorm + age * math.log(10.0)
        ln_age = torch.where((age < lo) | (age > hi), torch.full_like(age, -float("inf")), ln_age)
        lnprior = t_prior + ln_age
        lnlike = i_like + t_like
        lnpost = torch.where(torch.isfinite(lnprior), lnprior + lnlike, torch.full_like(lnprior, -float("inf")))
        return lnpost, lnprior, lnlike

    def _scalar_parts(self, p):
        """(lnpost, lnprior, lnlike) of ONE host row - the per-point callback of emcee / MultiNest (reference
        starmodel.py:2069-2104: lnprior from the track model and the age prior, lnlike = isochrone-grid + track-grid terms):
        two per-point calls through the component models' resident mailbox waves and the closed-form age prior on the host,
        instead of two batch launches and a dozen framework operations on one-row tensors."""
        eep, mass, age, feh, dist, av = (float(x) for x in p)
        _, t_prior, t_like = self._track_model._scalar_call((mass, eep, feh, dist, av), None)
        i_like = self._iso_model._scalar_call((eep, age, feh, dist, av), 2)
        lo, hi, lnorm = self.age_prior_constants()
        ln_age = -math.inf if (age < lo or age > hi) else lnorm + age * math.log(10.0)
        lnprior = t_prior + ln_age
        lnlike = i_like + t_like
        return (lnprior + lnlike if math.isfinite(lnprior) else -math.inf), lnprior, lnlike

    def _evaluate(self, p, which):
        tp = type(p)
        if ((tp is list or tp is tuple or (tp is np.ndarray and p.ndim == 1)) and len(p) == 6
                and not isinstance(p[0], (list, tuple, np.ndarray))):
            return self._scalar_parts(p)[which]
        if dev.is_tensor(p) and p.is_cuda:
            single = p.dim() == 1
            out = self.evaluate_device(p.double()[None, :] if single else p.double())[which]
            return out[0] if single else out
        arr = np.asarray(p, dtype=float)
        single = arr.ndim == 1
        out = self.evaluate_device(dev.to_device_f64(arr[None, :] if single else arr, dev.current_device()))[which]
        out = out.cpu().numpy()
        return float(out[0]) if single else out

    def lnpost(self, p):
        return self._evaluate(p, 0)

    def lnprior(self, p):
        return self._evaluate(p, 1)

    def lnlike(self, p):
        return self._evaluate(p, 2)

    # -- fits: the same drivers as the other models (fit_multinest / evidence come from the mixin) ----
    def sample_from_prior(self, n, rng=None, max_tries=200):
        """[n, 6] uniform draws from the parameter box with a finite lnpost."""
        rng = rng or np.random.default_rng()
        lo = np.array([self.bounds(nm)[0] for nm in self.param_names], dtype=float)
        hi = np.array([self.bounds(nm)[1] for nm in self.param_names], dtype=float)
        out = rng.uniform(lo, hi, size=(n, 6))
        for _ in range(max_tries):
            bad = ~np.isfinite(self.lnpost(out))
            if not bad.any():
                return out
            out[bad] = rng.uniform(lo, hi, size=(int(bad.sum()), 6))
        raise RuntimeError("could not find %d starting points with a finite lnpost" % n)

    emcee_p0 = sample_from_prior

    def fit_mcmc(self, nwalkers=300, nburn=200, niter=100, p0=None, seed=None, fused=None, n_ensembles=1, **kwargs):
        """Device-resident stretch-move ensemble (``k_stretch_isotrack``: both grids' evaluations inside one
        persistent kernel); ``fused=False``: the framework-op sampler around :meth:`evaluate_device`."""
        self._sampler = _run_mcmc_fit(self, nwalkers, nburn, niter, p0, seed, fused, n_ensembles)
        self._samples, self._fit_kind = None, "mcmc"
        return self._sampler

    fit_mcmc_old = fit_mcmc

    @property
    def samples(self):
        import pandas as pd
        if getattr(self, "_samples", None) is None:
            if getattr(self, "_fit_kind", "mcmc") == "nested":
                self._samples = self._nested_frame()
            else:
                if getattr(self, "_sampler", None) is None:
                    raise AttributeError("a fit must be run to access samples")
                df = pd.DataFrame(self._sampler.flatchain.cpu().numpy(), columns=list(self.param_names))
                df["lnprob"] = self._sampler.flatlnprobability.cpu().numpy()
                self._samples = df
        return self._samples
