"""The batched nested sampler behind fit_multinest, on analytic likelihoods (no GPU needed)."""
import numpy as np
import pytest

from isochrones_amd.nested import nested_sample, nested_sample_batched


def _gauss(mu, sig):
    mu, sig = np.asarray(mu, float), np.asarray(sig, float)

    def f(x):
        return -0.5 * np.sum(((x - mu) / sig) ** 2, axis=1)
    return f


def test_gaussian_evidence_and_posterior():
    d = 5
    mu = np.array([0.3, 0.5, 0.6, 0.45, 0.7]) * 10 - 2          # box [-2, 8]^5
    sig = np.array([0.3, 0.5, 0.2, 0.4, 0.6])
    res = nested_sample(_gauss(mu, sig), [-2.0] * d, [8.0] * d, nlive=600, seed=3)
    want = np.sum(np.log(np.sqrt(2 * np.pi) * sig)) - d * np.log(10.0)
    assert abs(res.logz - want) < 4 * res.logz_err + 0.05, (res.logz, want, res.logz_err)
    assert 0.05 < res.logz_err < 0.3 and res.efficiency > 0.05
    m = res.weights @ res.samples
    s = np.sqrt(res.weights @ (res.samples - m) ** 2)
    assert np.all(np.abs(m - mu) < 0.15 * sig) and np.all(np.abs(s / sig - 1) < 0.15)
    x, ll = res.equal_weight_samples(4000, np.random.default_rng(1))
    assert x.shape == (4000, d) and np.all(np.abs(x.mean(axis=0) - mu) < 0.2 * sig)
    assert np.allclose(ll, _gauss(mu, sig)(x))


def test_zero_likelihood_region_and_nan_are_excluded():
    """Half of the box has L = 0 (-inf / NaN, e.g. samples outside the model table): the evidence is
    the integral over the whole box, so it drops by the excluded mass only."""
    mu, sig = np.array([0.5, 0.5, 0.5]), np.array([0.05, 0.08, 0.04])
    g = _gauss(mu, sig)

    def f(x):
        ll = g(x)
        ll[x[:, 0] < 0.5] = -np.inf
        ll[(x[:, 1] > 0.9)] = np.nan
        return ll
    res = nested_sample(f, [0, 0, 0], [1, 1, 1], nlive=500, seed=5)
    want = np.sum(np.log(np.sqrt(2 * np.pi) * sig)) + np.log(0.5)
    assert abs(res.logz - want) < 4 * res.logz_err + 0.05, (res.logz, want)
    assert 0.3 < res.prior_fraction < 0.6
    assert np.all(res.samples[:, 0] >= 0.5) and np.all(res.samples[:, 1] <= 0.9)


def test_two_modes_both_recovered():
    sig = 0.03
    a, b = _gauss([0.25, 0.3], [sig, sig]), _gauss([0.75, 0.7], [sig, sig])

    def f(x):
        return np.logaddexp(a(x), b(x) + np.log(3.0))             # second mode carries 3x the mass
    res = nested_sample(f, [0, 0], [1, 1], nlive=800, seed=7)
    want = np.log(4.0 * 2 * np.pi * sig * sig)
    assert abs(res.logz - want) < 4 * res.logz_err + 0.05, (res.logz, want)
    right = res.weights[res.samples[:, 0] > 0.5].sum()
    assert abs(right - 0.75) < 0.06


def test_no_support_raises():
    with pytest.raises(RuntimeError):
        nested_sample(lambda x: np.full(x.shape[0], -np.inf), [0, 0], [1, 1], nlive=50, max_batch=4096)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_batched_variant_same_integral(seed):
    """K live points retired per macro-step (vectorised host loop): same evidence and posterior within the errors,
    on the Gaussian, the half-excluded box and the two-mode cases."""
    d = 5
    mu = np.array([0.3, 0.5, 0.6, 0.45, 0.7]) * 10 - 2
    sig = np.array([0.3, 0.5, 0.2, 0.4, 0.6])
    res = nested_sample_batched(_gauss(mu, sig), [-2.0] * d, [8.0] * d, nlive=600, seed=seed)
    want = np.sum(np.log(np.sqrt(2 * np.pi) * sig)) - d * np.log(10.0)
    assert abs(res.logz - want) < 4 * res.logz_err + 0.08, (res.logz, want, res.logz_err)
    m = res.weights @ res.samples
    s = np.sqrt(res.weights @ (res.samples - m) ** 2)
    assert np.all(np.abs(m - mu) < 0.15 * sig) and np.all(np.abs(s / sig - 1) < 0.15)
    sg = 0.03
    a, b = _gauss([0.25, 0.3], [sg, sg]), _gauss([0.75, 0.7], [sg, sg])
    res2 = nested_sample_batched(lambda x: np.logaddexp(a(x), b(x) + np.log(3.0)), [0, 0], [1, 1], nlive=800, seed=seed)
    assert abs(res2.logz - np.log(4.0 * 2 * np.pi * sg * sg)) < 4 * res2.logz_err + 0.08
    assert abs(res2.weights[res2.samples[:, 0] > 0.5].sum() - 0.75) < 0.07
    g = _gauss([0.5, 0.5, 0.5], [0.05, 0.08, 0.04])

    def f(x):
        ll = g(x)
        ll[x[:, 0] < 0.5] = -np.inf
        return ll
    res3 = nested_sample_batched(f, [0, 0, 0], [1, 1, 1], nlive=500, seed=seed)
    want3 = np.sum(np.log(np.sqrt(2 * np.pi) * np.array([0.05, 0.08, 0.04]))) + np.log(0.5)
    assert abs(res3.logz - want3) < 4 * res3.logz_err + 0.08


def test_proposal_hook_is_equivalent():
    """The device-proposal hook (here a numpy stand-in with the same contract: draw in the ellipsoid, evaluate,
    return only the points above the threshold) gives the same integral in both loops."""
    from isochrones_amd.nested import _draw_in_ellipsoid
    mu, sig = np.array([0.4, 0.6, 0.5]), np.array([0.05, 0.08, 0.04])
    g = _gauss(mu, sig)
    rng = np.random.default_rng(9)

    def propose(mean, A, want, threshold):
        u = rng.random((want, 3)) if mean is None else _draw_in_ellipsoid(rng, mean, A, want)
        ll = g(u)
        ok = ll > threshold
        return u[ok], ll[ok], want
    want = np.sum(np.log(np.sqrt(2 * np.pi) * sig))
    for fn in (nested_sample, nested_sample_batched):
        res = fn(g, [0, 0, 0], [1, 1, 1], nlive=500, seed=2, propose=propose)
        assert abs(res.logz - want) < 4 * res.logz_err + 0.08, (fn.__name__, res.logz, want)
        m = res.weights @ res.samples
        assert np.all(np.abs(m - mu) < 0.2 * sig)
