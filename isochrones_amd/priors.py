"""Host-side prior descriptions.

On the GPU the priors are evaluated inside the fused ``lnpost`` kernel; the host only owns the
per-model *constants* (bounds, normalisations) and packs them into ``iso_prior`` records
(include/isochrones_amd.h).  Class names and the meaning of ``bounds`` follow the reference
(isochrones/priors.py) so that ``StarModel._priors`` / ``set_prior`` / ``set_bounds`` read the
same.  The families the reference ships (its ``BasicStarModel`` defaults plus the simple bounded
ones) are evaluated on the device.  Any OTHER ``Prior`` object — a user's subclass that defines
``_pdf`` (and optionally ``_lnpdf`` / ``distribution``) the way the reference's base class expects
(priors.py:31-73) — is a *host prior*: the device sees a flat stand-in over its bounds and the model
adds ``lnpdf`` of that parameter on the host (``starmodel._HostPriorMixin``); such a model is fitted
by the framework-op sampler, not by the resident kernels.

Normalisation constants are obtained exactly the way the reference obtains them
(``scipy.integrate.quad`` over the same integrand: priors.py:42-45 for ``FehPrior``,
:171-203 for the broken Chabrier IMF) so that they agree to the last bit; closed forms are
used as a cross-check in the tests and as a fallback when scipy is unavailable.
"""
from __future__ import annotations

import math

import numpy as np

from . import _cabi

_ROOT_2PI = float(np.sqrt(2 * np.pi))
_LN10 = float(np.log(10))

try:  # scipy is part of the image; keep a closed-form fallback anyway
    from scipy.integrate import quad as _quad
except Exception:  # pragma: no cover
    _quad = None


def _ncdf(z):
    return 0.5 * math.erfc(-z / math.sqrt(2.0))


#: bumped whenever ANY prior object (or a model's EEP prior) is mutated: a model's cached scalar-call state
#: (starmodel.BasicStarModel._scalar_call) is valid as long as this number has not moved
EPOCH = [0]


class Prior:
    """Common surface: ``bounds``, ``pdf(x)``, ``lnpdf(x)``, ``__call__`` (= pdf), ``desc()``."""

    kind = 0
    bounded = 0
    _bounds = (-np.inf, np.inf)
    _version = 0          # bumped on every mutation: models sharing a prior object notice (starmodel._prior_state)

    @property
    def bounds(self):
        return self._bounds

    @bounds.setter
    def bounds(self, new):
        self._bounds = (float(new[0]), float(new[1]))
        self._version += 1
        EPOCH[0] += 1
        if self.kind == 0 and getattr(self, "_pdf", None) is not None and _quad is not None:
            self._norm = _quad(self._pdf, *self._bounds)[0]          # (reference priors.py:42-45)
        self._rebuild()

    def _rebuild(self):
        pass

    def _raw(self, x):
        # a user's subclass in the reference's style defines _pdf; _norm is the integral over its bounds (priors.py:42-60)
        f = getattr(self, "_pdf", None)
        if f is None:
            raise NotImplementedError
        return f(x) / getattr(self, "_norm", 1.0)

    def pdf(self, x):
        lo, hi = self.bounds
        if x < lo or x > hi:
            return 0.0
        return self._raw(x)

    __call__ = pdf

    def pdf_array(self, x):
        """``pdf`` of every element of an array (what start-point generation evaluates by the thousand); the same values
        as ``pdf`` element by element: 0 outside the bounds, NaN for NaN."""
        x = np.asarray(x, dtype=float)
        lo, hi = self.bounds
        with np.errstate(all="ignore"):
            raw = np.asarray(self._raw_array(x), dtype=float)
        return np.where((x < lo) | (x > hi), 0.0, raw)

    def _raw_array(self, x):
        return np.array([self._raw(float(v)) for v in x.ravel()]).reshape(x.shape)

    def lnpdf(self, x):
        own = getattr(self, "_lnpdf", None)          # (the reference's hook, priors.py:62-67: no bounds test in front of it)
        if own is not None and self.kind == 0:
            return own(x)
        if self.bounded:
            lo, hi = self.bounds
            if x < lo or x > hi:
                return -np.inf
        p = self.pdf(x)
        return float(np.log(p)) if p else -np.inf

    def sample(self, n, rng=None):
        """The reference's base class draws from ``self.distribution`` when a subclass has one (priors.py:69-73); a prior
        with finite bounds and a pdf is drawn by rejection under its largest value on a grid (start points only need
        draws that cover the support)."""
        dist = getattr(self, "distribution", None)
        if dist is not None:
            return np.asarray(dist.rvs(n), dtype=float)
        lo, hi = self.bounds
        if not (np.isfinite(lo) and np.isfinite(hi)) or getattr(self, "_pdf", None) is None:
            raise NotImplementedError
        rng = rng or np.random.default_rng()
        top = 1.05 * max(self.pdf(float(v)) for v in np.linspace(lo, hi, 2049))
        out = np.empty(0)
        while out.size < n:
            x = rng.uniform(lo, hi, size=max(2 * (n - out.size), 64))
            keep = rng.uniform(0.0, top, size=x.size) < np.array([self.pdf(float(v)) for v in x])
            out = np.concatenate([out, x[keep]])
        return out[:n]

    # the reference's self-checks (priors.py:74-104), used by its tests/test_priors.py
    def test_integral(self):
        from scipy.integrate import quad
        lo, hi = self.bounds
        pts = [b for b in getattr(self, "breakpoints", ()) if lo < b < hi] or None
        assert np.isclose(1.0, quad(self.pdf, lo, hi, points=pts, limit=200)[0], rtol=1e-6)

    def test_sampling(self, n=100000, rng=None):
        """Histogram of ``sample(n)`` against the bin-averaged pdf: every bin with more than 50
        draws within 6 sigma (Poisson), the reference's criterion."""
        from scipy.integrate import quad
        x = self.sample(n, rng)
        rng_ = None if not np.all(np.isfinite(self.bounds)) else self.bounds
        hn, b = np.histogram(x, range=rng_)
        h = hn / (hn.sum() * np.diff(b))
        pdf = np.array([quad(self.pdf, lo, hi, limit=200)[0] / (hi - lo) for lo, hi in zip(b[:-1], b[1:])])
        ok = hn > 50
        resid = np.abs(pdf[ok] - h[ok]) / pdf[ok] * np.sqrt(hn[ok])
        assert resid.max() < 6, resid

    def _fields(self):
        return {}

    def desc(self) -> _cabi.IsoPrior:
        d = _cabi.IsoPrior()
        d.kind = self.kind
        d.bounded = self.bounded
        d.lo, d.hi = self.bounds
        for k, v in self._fields().items():
            setattr(d, k, float(v))
        return d


class FlatPrior(Prior):
    kind = _cabi.PRIOR_FLAT
    bounded = 1

    def __init__(self, bounds):
        self._bounds = (float(bounds[0]), float(bounds[1]))

    def _raw(self, x):
        lo, hi = self.bounds
        return 1.0 / (hi - lo)

    def _raw_array(self, x):
        return np.full(x.shape, self._raw(0.0))

    def sample(self, n, rng=None):
        rng = rng or np.random.default_rng()
        lo, hi = self.bounds
        return rng.random(n) * (hi - lo) + lo


class FlatLogPrior(Prior):
    """pdf flat in 10**x (x is a log10 quantity)."""
    kind = _cabi.PRIOR_FLATLOG
    bounded = 1

    def __init__(self, bounds):
        self._bounds = (float(bounds[0]), float(bounds[1]))

    def _raw(self, x):
        lo, hi = self.bounds
        return _LN10 * 10 ** x / (10 ** hi - 10 ** lo)

    _raw_array = _raw

    def sample(self, n, rng=None):
        rng = rng or np.random.default_rng()
        lo, hi = self.bounds
        return np.log10(rng.random(n) * (10 ** hi - 10 ** lo) + 10 ** lo)


class PowerLawPrior(Prior):
    kind = _cabi.PRIOR_POWERLAW
    bounded = 1

    def __init__(self, alpha, bounds):
        self.alpha = float(alpha)
        self._bounds = (float(bounds[0]), float(bounds[1]))

    def _C(self):
        lo, hi = self.bounds
        a1 = 1 + self.alpha
        return a1 / (hi ** a1 - lo ** a1)

    def _raw(self, x):
        return self._C() * x ** self.alpha

    _raw_array = _raw

    def lnpdf(self, x):
        lo, hi = self.bounds
        if x < lo or x > hi:
            return -np.inf
        with np.errstate(divide="ignore"):
            return float(np.log(self._C()) + self.alpha * np.log(x))

    def sample(self, n, rng=None):
        rng = rng or np.random.default_rng()
        lo, hi = self.bounds
        a1 = 1 + self.alpha
        u = rng.random(n)
        return (u * (hi ** a1 - lo ** a1) + lo ** a1) ** (1 / a1)

    def _fields(self):
        return dict(a=self.alpha)


class GaussianPrior(Prior):
    kind = _cabi.PRIOR_GAUSS

    def __init__(self, mean, sigma, bounds=None):
        self.mean, self.sigma = float(mean), float(sigma)
        self.bounded = 0
        self.norm, self.lognorm = 1.0, 0.0
        if bounds is not None:
            self.bounded = 1
            self._bounds = (float(bounds[0]), float(bounds[1]))
            self.norm = self._mass_inside()
            # a truncation window that holds no probability mass gives norm = 0: -inf, as numpy's log in the reference
            # (priors.py:246-247), not an exception
            self.lognorm = math.log(self.norm) if self.norm > 0 else -math.inf

    def _mass_inside(self):
        a, b = (self._bounds[0] - self.mean) / self.sigma, (self._bounds[1] - self.mean) / self.sigma
        return _ncdf(b) - _ncdf(a)

    def _rebuild(self):
        """Bounds assigned after construction (``prior.bounds = ...``, ``model.set_bounds``): the prior is bounded
        from then on - lnpdf is -inf outside, on the host and in the device descriptor - as the reference's
        ``BoundedPrior.lnpdf`` (priors.py:131-140).  The normalisation stays the one of the constructor, and, as the
        reference's bounds setter does with its integral test (priors.py:123-129), bounds under which the density no
        longer integrates to one are refused."""
        self.bounded = 1
        if not np.isclose(self._mass_inside() / self.norm if self.norm > 0 else np.inf, 1.0):
            raise ValueError("Problem setting bounds to {}; integral test failed.".format(self._bounds))

    def _raw(self, x):
        z = (x - self.mean) / self.sigma
        if self.norm == 0:
            return math.inf
        return math.exp(-(z * z) / 2.0) / _ROOT_2PI / self.sigma / self.norm

    def _raw_array(self, x):
        z = (x - self.mean) / self.sigma
        if self.norm == 0:
            return np.full(x.shape, np.inf)
        return np.exp(-(z * z) / 2.0) / _ROOT_2PI / self.sigma / self.norm

    def lnpdf(self, x):
        if self.bounded and (x < self._bounds[0] or x > self._bounds[1]):
            return -np.inf
        z = (x - self.mean) / self.sigma
        return -(z * z) / 2.0 - math.log(_ROOT_2PI) - math.log(self.sigma) - self.lognorm

    def sample(self, n, rng=None):
        rng = rng or np.random.default_rng()
        out = rng.standard_normal(n) * self.sigma + self.mean
        if self.bounded:
            lo, hi = self._bounds
            bad = (out < lo) | (out > hi)
            while bad.any():
                out[bad] = rng.standard_normal(int(bad.sum())) * self.sigma + self.mean
                bad = (out < lo) | (out > hi)
        return out

    def _fields(self):
        return dict(a=self.mean, b=self.sigma, c=self.lognorm)


class LogNormalPrior(Prior):
    kind = _cabi.PRIOR_LOGNORMAL

    def __init__(self, mu, sigma):
        self.mu, self.sigma = float(mu), float(sigma)
        self.scale = float(np.exp(mu))
        self._bounds = (0.0, np.inf)

    def _raw(self, x):
        s = self.sigma
        with np.errstate(all="ignore"):          # x = 0: inf * 0 = nan, as numpy gives the reference - not an exception
            y = np.float64(x) / self.scale if np.isscalar(x) else x / self.scale
            return (1.0 / _ROOT_2PI) / (s * y) * np.exp(-0.5 * (np.log(y) / s) ** 2) / self.scale

    _raw_array = _raw

    def lnpdf(self, x):
        s = self.sigma
        y = x / self.scale
        return float(np.log(1.0 / _ROOT_2PI) - (np.log(s) + np.log(y)) - 0.5 * (np.log(y) / s) ** 2 - self.mu)

    def sample(self, n, rng=None):
        rng = rng or np.random.default_rng()
        return np.exp(rng.standard_normal(n) * self.sigma + self.mu)

    def _fields(self):
        return dict(a=self.mu, b=self.sigma)


class ChabrierPrior(Prior):
    """Chabrier (2003) IMF: log-normal below the breakpoint, power law above, continuous at the
    breakpoint and normalised over ``bounds`` (reference: priors.py:143-232, 514-519)."""
    kind = _cabi.PRIOR_CHABRIER

    def __init__(self, bounds=(0.1, 100.0), mu=math.log(0.079), sigma=0.69 * math.log(10),
                 alpha=-2.35, breakpoint=1.0, powerlaw_bounds=(1.0, 100.0)):
        self.low = LogNormalPrior(mu, sigma)
        self.high = PowerLawPrior(alpha, powerlaw_bounds)
        self.breakpoint = float(breakpoint)
        self.breakpoints = (self.breakpoint,)
        self._bounds = (float(bounds[0]), float(bounds[1]))
        self._rebuild()

    _norm_cache = {}

    def _rebuild(self):
        lo, hi = self._bounds
        bp = self.breakpoint
        key = (lo, hi, bp, self.low.mu, self.low.sigma, self.high.alpha, self.high.bounds)
        if key in ChabrierPrior._norm_cache:      # catalogs build thousands of identical priors
            self.norms = ChabrierPrior._norm_cache[key].copy()
            self.lognorms = np.log(self.norms)
            return
        ratio = self.high(bp) / self.low(bp)
        if _quad is not None:
            tot = (_quad(lambda x: self.low(x) / 1.0, lo, bp, limit=200)[0]
                   + _quad(lambda x: self.high(x) / ratio, bp, hi, limit=200)[0])
        else:  # pragma: no cover
            tot = self.closed_form_total(ratio)
        self.norms = np.array([1.0, ratio]) * tot
        self.lognorms = np.log(self.norms)
        ChabrierPrior._norm_cache[key] = self.norms.copy()

    def closed_form_total(self, ratio=None):
        lo, hi = self._bounds
        bp = self.breakpoint
        if ratio is None:
            ratio = self.high(bp) / self.low(bp)
        s, mu = self.low.sigma, self.low.mu
        low_mass = _ncdf((math.log(bp) - mu) / s) - (_ncdf((math.log(lo) - mu) / s) if lo > 0 else 0.0)
        plo, phi = self.high.bounds
        a1 = 1 + self.high.alpha
        top = min(hi, phi)
        high_mass = (top ** a1 - max(bp, plo) ** a1) / (phi ** a1 - plo ** a1) if top > bp else 0.0
        return low_mass + high_mass / ratio

    def _raw(self, x):
        if x < self.breakpoint:
            return self.low(x) / self.norms[0]
        return self.high(x) / self.norms[1]

    def _raw_array(self, x):
        return np.where(x < self.breakpoint, self.low.pdf_array(x) / self.norms[0], self.high.pdf_array(x) / self.norms[1])

    def lnpdf(self, x):
        # no bounds test on this path in the reference (BrokenPrior._lnpdf)
        if x < self.breakpoint:
            return self.low.lnpdf(x) - self.lognorms[0]
        return self.high.lnpdf(x) - self.lognorms[1]

    def sample(self, n, rng=None):
        rng = rng or np.random.default_rng()
        lo, hi = self._bounds
        out = np.empty(n)
        filled = 0
        # rejection from a log-uniform proposal — adequate for start-point generation
        grid = np.geomspace(max(lo, 1e-3), hi, 400)
        pmax = float(np.max(self.pdf_array(grid) * grid))
        while filled < n:
            m = max(2 * (n - filled), 64)
            x = np.exp(rng.uniform(np.log(max(lo, 1e-3)), np.log(hi), m))
            keep = rng.random(m) * pmax * 1.05 < self.pdf_array(x) * x
            take = x[keep][: n - filled]
            out[filled:filled + take.size] = take
            filled += take.size
        return out

    def _fields(self):
        return dict(a=self.low.mu, b=self.low.sigma, c=self.high.alpha, d=self.breakpoint,
                    e=self.norms[0], f=self.norms[1], g=self.high.bounds[0], h=self.high.bounds[1])


class FehPrior(Prior):
    """Two-Gaussian local disk + halo metallicity distribution (reference: priors.py:345-381)."""
    kind = _cabi.PRIOR_FEH

    def __init__(self, halo_fraction=0.001, local=True, bounds=None):
        self.halo_fraction = float(halo_fraction)
        self.local = bool(local)
        self._norm = 1.0
        if bounds is not None:
            self.bounds = bounds

    _norm_cache = {}

    def _rebuild(self):
        lo, hi = self._bounds
        key = (lo, hi, self.halo_fraction, self.local)
        if key in FehPrior._norm_cache:
            self._norm = FehPrior._norm_cache[key]
            return
        if _quad is not None:
            self._norm = _quad(self._shape, lo, hi)[0]
            FehPrior._norm_cache[key] = self._norm
            return
        else:  # pragma: no cover
            self._norm = self.closed_form_norm()

    def closed_form_norm(self):
        lo, hi = self._bounds
        comps = ([(0.8, 0.016, 0.15), (0.2, -0.15, 0.22)] if self.local else [(1.0, -0.3, 0.3)])
        disk = sum(w * (_ncdf((hi - m) / s) - _ncdf((lo - m) / s)) for w, m, s in comps)
        if self.local:
            disk *= _ROOT_2PI / 2.5066282746310007
        halo = _ncdf((hi + 1.5) / 0.4) - _ncdf((lo + 1.5) / 0.4)
        return self.halo_fraction * halo + (1 - self.halo_fraction) * disk

    def _shape(self, feh):
        if self.local:
            disk = (1.0 / 2.5066282746310007
                    * (0.8 / 0.15 * np.exp(-0.5 * (feh - 0.016) ** 2.0 / 0.15 ** 2.0)
                       + 0.2 / 0.22 * np.exp(-0.5 * (feh + 0.15) ** 2.0 / 0.22 ** 2.0)))
        else:
            mu, sig = -0.3, 0.3
            disk = 1.0 / np.sqrt(2 * np.pi) / sig * np.exp(-0.5 * (feh - mu) ** 2 / sig ** 2)
        hmu, hsig = -1.5, 0.4
        halo = 1.0 / np.sqrt(2 * np.pi * hsig ** 2) * np.exp(-0.5 * (feh - hmu) ** 2 / hsig ** 2)
        return self.halo_fraction * halo + (1 - self.halo_fraction) * disk

    def _raw(self, x):
        return self._shape(x) / self._norm

    _raw_array = _raw

    def sample(self, n, rng=None):
        rng = rng or np.random.default_rng()
        comps = ([(0.8, 0.016, 0.15), (0.2, -0.15, 0.22)] if self.local else [(1.0, -0.3, 0.3)])
        w = np.array([c[0] for c in comps]) * (1 - self.halo_fraction)
        w = np.append(w, self.halo_fraction)
        mus = np.array([c[1] for c in comps] + [-1.5])
        sig = np.array([c[2] for c in comps] + [0.4])
        k = rng.choice(len(w), size=n, p=w / w.sum())
        out = rng.standard_normal(n) * sig[k] + mus[k]
        lo, hi = self.bounds
        bad = (out < lo) | (out > hi)
        while bad.any():
            kk = rng.choice(len(w), size=int(bad.sum()), p=w / w.sum())
            out[bad] = rng.standard_normal(kk.size) * sig[kk] + mus[kk]
            bad = (out < lo) | (out > hi)
        return out

    def _fields(self):
        return dict(a=self.halo_fraction, b=self._norm, c=1.0 if self.local else 0.0)


class AgePrior(FlatLogPrior):
    """Uniform in linear age; the parameter is log10(age/yr)."""

    def __init__(self, bounds=(5, 10.15)):
        super().__init__(bounds)


class DistancePrior(PowerLawPrior):
    def __init__(self, max_distance=10000):
        super().__init__(alpha=2.0, bounds=(0, max_distance))


class AVPrior(FlatPrior):
    def __init__(self, bounds=(0, 1.0)):
        super().__init__(bounds)


class QPrior(PowerLawPrior):
    """Mass-ratio prior (reference priors.py:502-505)."""

    def __init__(self, bounds=(0.1, 1)):
        super().__init__(alpha=0.3, bounds=bounds)


class SalpeterPrior(PowerLawPrior):
    """Salpeter IMF (reference priors.py:508-511)."""

    def __init__(self, bounds=(0.1, 10)):
        super().__init__(alpha=-2.35, bounds=bounds)


ONE_OVER_ROOT_2PI = 1.0 / _ROOT_2PI
LOG_ONE_OVER_ROOT_2PI = float(np.log(ONE_OVER_ROOT_2PI))


def powerlaw_pdf(x, alpha, lo, hi):
    """Normalised x^alpha on (lo, hi) (reference: priors.py:470-473)."""
    a1 = alpha + 1
    return a1 / (hi ** a1 - lo ** a1) * x ** alpha


def powerlaw_lnpdf(x, alpha, lo, hi):
    a1 = alpha + 1
    return np.log(a1 / (hi ** a1 - lo ** a1)) + alpha * np.log(x)


def BrokenPrior(components, breakpoints, bounds=None):
    """The reference's stitched-together prior (priors.py:143-232).  The device evaluates one such composition,
    the one the reference itself uses: a log-normal below a single breakpoint and a power law above it — with
    any parameters — so that is what this builds (a :class:`ChabrierPrior` with those parameters)."""
    comps = list(components)
    if (len(comps) == 2 and len(breakpoints) == 1 and isinstance(comps[0], LogNormalPrior)
            and isinstance(comps[1], PowerLawPrior)):
        lo_hi = bounds if bounds is not None else (0.0, float(comps[1].bounds[1]))
        return ChabrierPrior(bounds=lo_hi, mu=comps[0].mu, sigma=comps[0].sigma, alpha=comps[1].alpha,
                             breakpoint=float(breakpoints[0]), powerlaw_bounds=tuple(comps[1].bounds))
    raise NotImplementedError("the device evaluates broken priors of the form [LogNormalPrior, PowerLawPrior] with one "
                              "breakpoint (the Chabrier composition); got %s" % [type(c).__name__ for c in comps])


BoundedPrior = Prior        # the reference splits Prior / BoundedPrior (priors.py:107-141); one class covers both here


def __getattr__(name):      # priors.EEP_prior (reference priors.py:384-463) lives with the models that own it
    if name in ("EEP_prior", "EEPPrior"):
        from .starmodel import EEPPrior
        return EEPPrior
    raise AttributeError("module {!r} has no attribute {!r}".format(__name__, name))


DEVICE_PRIOR_TYPES = (FlatPrior, FlatLogPrior, PowerLawPrior, GaussianPrior, LogNormalPrior,
                      ChabrierPrior, FehPrior)


def is_host_prior(p) -> bool:
    """A prior object the device has no family for (a user's ``Prior`` subclass, a foreign object with ``lnpdf`` and
    ``bounds``): evaluated on the host, parameter by parameter."""
    return not isinstance(p, DEVICE_PRIOR_TYPES)


def check_host_prior(p, prop):
    if not callable(getattr(p, "lnpdf", None)) or not hasattr(p, "bounds"):
        raise TypeError("prior for %r must offer lnpdf(x) and bounds (got %r)" % (prop, p))


def flat_stand_in(p) -> "FlatPrior":
    """What the device evaluates in a host prior's place: flat over the prior's bounds (an unbounded side becomes
    +-1e300), so the bounds test stays on the device and the host adds ``lnpdf(x) + log(width)``."""
    lo, hi = p.bounds
    lo = float(lo) if np.isfinite(lo) else -1e300
    hi = float(hi) if np.isfinite(hi) else 1e300
    return FlatPrior((lo, hi))


def lnpdf_array(p, x):
    """``p.lnpdf`` of every element of ``x`` (a prior's own ``lnpdf_array`` if it has one)."""
    x = np.asarray(x, dtype=float)
    own = getattr(p, "lnpdf_array", None)
    if own is not None:
        return np.asarray(own(x), dtype=float)
    with np.errstate(all="ignore"):
        return np.array([p.lnpdf(float(v)) for v in x.ravel()], dtype=float).reshape(x.shape)

# ---- plain-data form of a prior (saved models) ----------------------------------------------------
#: the only classes a saved model may name
_SPEC_CLASSES = {c.__name__: c for c in (FlatPrior, FlatLogPrior, PowerLawPrior, GaussianPrior, LogNormalPrior,
                                         ChabrierPrior, FehPrior, AgePrior, DistancePrior, AVPrior, QPrior,
                                         SalpeterPrior)}


def prior_to_spec(p):
    """``{"cls": name, ...constructor parameters...}`` - numbers, booleans and lists only (JSON)."""
    name = type(p).__name__
    if name not in _SPEC_CLASSES:
        raise TypeError("cannot serialise a {}".format(name))
    lo, hi = (float(b) for b in p.bounds)
    if isinstance(p, ChabrierPrior):
        return dict(cls=name, bounds=[lo, hi], mu=p.low.mu, sigma=p.low.sigma, alpha=p.high.alpha,
                    breakpoint=float(p.breakpoints[0]), powerlaw_bounds=[float(b) for b in p.high.bounds])
    if isinstance(p, PowerLawPrior):
        return dict(cls=name, alpha=p.alpha, bounds=[lo, hi])
    if isinstance(p, (FlatPrior, FlatLogPrior)):
        return dict(cls=name, bounds=[lo, hi])
    if isinstance(p, GaussianPrior):
        # the normalisation is a record of its own: bounds assigned after construction do not change it
        return dict(cls=name, mean=p.mean, sigma=p.sigma, bounds=[lo, hi] if p.bounded else None, norm=p.norm)
    if isinstance(p, LogNormalPrior):
        return dict(cls=name, mu=p.mu, sigma=p.sigma)
    if isinstance(p, FehPrior):
        return dict(cls=name, halo_fraction=p.halo_fraction, local=p.local,
                    bounds=[lo, hi] if np.isfinite(lo) and np.isfinite(hi) else None)
    raise TypeError("cannot serialise a {}".format(name))


def prior_from_spec(spec):
    """Inverse of :func:`prior_to_spec`; the class is looked up in an explicit table, every parameter is coerced to
    float / bool, nothing else in the record is interpreted."""
    if not isinstance(spec, dict) or spec.get("cls") not in _SPEC_CLASSES:
        raise ValueError("not a prior record: {!r}".format(spec))
    cls = _SPEC_CLASSES[spec["cls"]]
    num = lambda k: float(spec[k])
    pair = lambda k: None if spec.get(k) is None else (float(spec[k][0]), float(spec[k][1]))
    obj = cls.__new__(cls)                 # sub-classes (AgePrior, DistancePrior, ...) only differ in their defaults
    if issubclass(cls, ChabrierPrior):
        ChabrierPrior.__init__(obj, bounds=pair("bounds"), mu=num("mu"), sigma=num("sigma"), alpha=num("alpha"),
                               breakpoint=num("breakpoint"), powerlaw_bounds=pair("powerlaw_bounds"))
    elif issubclass(cls, PowerLawPrior):
        PowerLawPrior.__init__(obj, num("alpha"), pair("bounds"))
    elif issubclass(cls, FlatPrior):
        FlatPrior.__init__(obj, pair("bounds"))
    elif issubclass(cls, FlatLogPrior):
        FlatLogPrior.__init__(obj, pair("bounds"))
    elif issubclass(cls, GaussianPrior):
        GaussianPrior.__init__(obj, num("mean"), num("sigma"))
        if pair("bounds") is not None:
            obj.bounded, obj._bounds = 1, pair("bounds")
        obj.norm = num("norm")
        obj.lognorm = math.log(obj.norm) if obj.norm > 0 else -math.inf
    elif issubclass(cls, LogNormalPrior):
        LogNormalPrior.__init__(obj, num("mu"), num("sigma"))
    else:
        FehPrior.__init__(obj, halo_fraction=num("halo_fraction"), local=bool(spec["local"]), bounds=pair("bounds"))
    return obj
