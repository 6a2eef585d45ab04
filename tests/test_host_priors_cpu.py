"""Host side of priors the device has no family for (isochrones_amd/priors.py): a user's subclass in the reference's style
(isochrones/priors.py:31-73: ``_pdf`` normalised over the bounds by the ``bounds`` setter, ``_lnpdf`` taken as it is,
``distribution.rvs`` for draws) behaves as the reference's base class would make it behave."""
import math

import numpy as np
import pytest

from isochrones_amd import priors as P


class Ramp(P.Prior):
    def __init__(self, bounds):
        self._norm = 1.0
        self.bounds = bounds

    def _pdf(self, x):
        return x


class WithOwnLog(P.Prior):
    def __init__(self):
        self._norm = 1.0

    def _pdf(self, x):
        return math.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)

    def _lnpdf(self, x):
        return -0.5 * x * x - 0.5 * math.log(2 * math.pi)


def test_subclass_with_pdf_only():
    r = Ramp((0.0, 2.0))
    assert r._norm == pytest.approx(2.0) and r.pdf(1.0) == pytest.approx(0.5) and r(1.0) == r.pdf(1.0)
    assert r.lnpdf(1.0) == pytest.approx(math.log(0.5)) and r.lnpdf(3.0) == -math.inf and r.pdf(-1.0) == 0.0
    r.test_integral()
    r.bounds = (1.0, 2.0)                                 # renormalised, as the reference's setter does
    assert r._norm == pytest.approx(1.5)
    r.test_integral()
    x = r.sample(20000, np.random.default_rng(0))
    assert x.min() >= 1.0 and x.max() <= 2.0 and abs(x.mean() - 14.0 / 9.0) < 0.01      # E[x] of x / 1.5 on [1, 2]
    r.test_sampling(rng=np.random.default_rng(1))
    assert P.is_host_prior(r) and not P.is_host_prior(P.FlatPrior((0, 1)))
    assert np.array_equal(P.lnpdf_array(r, [1.5, 5.0]), [r.lnpdf(1.5), -np.inf])
    assert P.flat_stand_in(r).bounds == (1.0, 2.0)


def test_subclass_with_its_own_lnpdf_and_no_bounds():
    g = WithOwnLog()
    assert g.bounds == (-np.inf, np.inf)
    assert g.lnpdf(1.0) == -0.5 - 0.5 * math.log(2 * math.pi)          # _lnpdf as it is (no log of a pdf)
    assert P.flat_stand_in(g).bounds == (-1e300, 1e300)
    with pytest.raises(NotImplementedError):
        g.sample(3)                                                     # no distribution, no finite bounds: as the reference
    g.distribution = type("D", (), {"rvs": staticmethod(lambda n: np.zeros(n))})()
    assert g.sample(4).shape == (4,)


def test_device_families_are_untouched():
    f = P.FlatPrior((0.0, 4.0))
    assert f.lnpdf(1.0) == math.log(0.25) and f.lnpdf(5.0) == -math.inf
    p = P.GaussianPrior(0.0, 1.0, bounds=(-1, 1))
    assert p.kind != 0 and not P.is_host_prior(p)
    P.check_host_prior(f, "AV")
    with pytest.raises(TypeError):
        P.check_host_prior(object(), "AV")
