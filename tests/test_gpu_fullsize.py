"""GPU parity at BASELINE.json's full sizes: the full MIST-shaped tables (track [15,196,1710,18],
isochrone [107,15,1710,16], BC [70,26,18,13,nb]) and 10^6-sample batches, checked through a
seeded 50k sub-sample against the CPU oracle plus size-independent properties of the whole batch
(permutation equivariance, layout invariance, lnpost = lnprior + lnlike, prior short-circuit)."""
import numpy as np
import pytest

import isochrones_amd as ia
from tests import _fixtures as fx

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-10


def _properties(mod, pars):
    import torch
    n = pars.shape[0]
    pt = torch.as_tensor(pars, device="cuda")
    post = mod.lnpost(pt)
    soa = mod.lnpost(pt.T.contiguous(), soa=True)
    clean = lambda t: torch.nan_to_num(t, nan=5.0, neginf=-1e300)
    assert torch.equal(clean(post), clean(soa))                                   # row-major == SoA
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    assert torch.equal(clean(mod.lnpost(pt[perm])), clean(post[perm]))            # permutation equivariance
    _, prior, like = mod.evaluate_device(pt, parts=True)
    fin = torch.isfinite(prior)
    tot = prior + like
    ok = fin & torch.isfinite(tot)
    assert torch.allclose(post[ok], tot[ok], rtol=1e-12, atol=1e-12)              # lnpost = lnprior + lnlike
    assert bool((post[~fin] == -float("inf")).all())                              # prior short-circuit
    assert 0.02 < float(torch.isfinite(post).double().mean()) < 1.0
    return post


def test_cfg2_full_size_single_star():
    import bench
    ic, mod = bench.build_model()
    assert ic.model_grid.interp.grid.shape == (15, 196, 1710, 18) and ic.bc_grid.interp.grid.shape == (70, 26, 18, 13, 1)
    pars = bench.make_samples(np.random.default_rng(12345), 1_000_000, "prior")
    post = _properties(mod, pars)
    oic = fx.make_oracle_ic(ic)
    sub = np.random.default_rng(0).choice(pars.shape[0], 50_000, replace=False)
    want = oic.lnpost(mod.model_desc(), pars[sub].T.copy(), nthreads=16)
    fx.assert_close(post.cpu().numpy()[sub], want[0], RTOL, atol=ATOL, what="cfg2 lnpost")
    fx.assert_close(mod.lnprior(pars[sub]), want[1], RTOL, atol=ATOL, what="cfg2 lnprior")
    fx.assert_close(mod.lnlike(pars[sub]), want[2], RTOL, atol=ATOL, what="cfg2 lnlike")
    # interp_value / interp_mag of the same points
    T, g, f, m = ic.interp_mag([pars[sub, j] for j in range(5)], ["V"])
    wT, wg, wf, wm = oic.interp_mag(pars[sub].T.copy(), [0], nthreads=16)
    fx.assert_close(T, wT, 1e-12, what="Teff")
    fx.assert_close(m, wm, 1e-11, atol=1e-12, what="V mag")


def test_cfg3_full_size_binary_six_bands():
    bands = ("J", "H", "K", "BP", "RP", "G")
    ic = ia.synthetic_isochrone(bands=bands)
    assert ic.model_grid.interp.grid.shape == (107, 15, 1710, 16)
    mod = ia.BinaryStarModel(ic, J=(9.3, 0.02), H=(9.0, 0.02), K=(8.95, 0.02), BP=(10.7, 0.002), RP=(9.8, 0.002),
                             G=(10.3, 0.001), parallax=(2.0, 0.05))
    rng = np.random.default_rng(3)
    lo = np.array([1.0, 1.0, 5.0, -4.0, 1.0, 0.0])
    hi = np.array([1710.0, 1710.0, 10.3, 0.5, 1000.0, 1.0])
    pars = rng.uniform(lo, hi, size=(1_000_000, 6))
    pars[:, :2] = -np.sort(-pars[:, :2], axis=1)
    post = _properties(mod, pars)
    oic = fx.make_oracle_ic(ic)
    sub = np.random.default_rng(1).choice(pars.shape[0], 50_000, replace=False)
    want = oic.lnpost(mod.model_desc(), pars[sub].T.copy(), nthreads=16)
    # same tolerance as cfg 2.  (sigma_G = 0.001 mag turns a 1e-14 mag rounding difference into 2 r delta / (2 sigma^2)
    # = 1e-8 * |r| of lnlike, but |lnlike| itself is then r^2 / (2 sigma^2) = 5e5 r^2: the relative term covers it.)
    fx.assert_close(post.cpu().numpy()[sub], want[0], RTOL, atol=ATOL, what="cfg3 lnpost")
    fx.assert_close(mod.lnlike(pars[sub]), want[2], RTOL, atol=ATOL, what="cfg3 lnlike")
