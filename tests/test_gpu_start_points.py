"""iso_catalog_start_points (isochrones_amd/csrc/fast/start_points.h): the start points of a catalog fit, drawn, evaluated and
selected on the device.  What the reference does here: `sample_from_prior` per star until nwalkers rows have a finite
lnpost (isochrones/starmodel.py:903-949).  Checked: the kept lnpost values are the oracle's lnpost of the kept
positions; the kept rows are each star's best W of exactly the candidates the kernel's Philox stream defines (rebuilt on
the host and evaluated with the oracle); ordering, bounds, failure isolation, determinism; and the fit built on them."""
import numpy as np
import pytest

import isochrones_amd as ia
from isochrones_amd.catalog import CatalogPosterior, StarCatalog, initial_positions
from oracle.cpu_sampler import philox4x32_10
from tests import _fixtures as fx

pytestmark = pytest.mark.gpu
BLOCK = 256


def small_ic(kind, bands):
    fehs = np.array([-2.0, -1.0, -0.5, -0.25, 0.0, 0.25, 0.5])
    if kind == "track":
        masses = ia.grids.mist_masses()[20:150:3]
        eeps = np.arange(200.0, 700.0)
        return ia.synthetic_track(bands=bands, fehs=fehs, masses=masses, eeps=eeps, eep_bounds=(eeps[0], eeps[-1]),
                                  limits=dict(mass=(masses[0], masses[-1]), feh=(-2.0, 0.5), age=(5, 10.13)))
    ages = ia.grids.mist_log_ages()[40::3]
    eeps = np.arange(150.0, 700.0)
    return ia.synthetic_isochrone(bands=bands, ages=ages, fehs=fehs, eeps=eeps, eep_bounds=(eeps[0], eeps[-1]),
                                  limits=dict(age=(ages[0], ages[-1]), feh=(-2.0, 0.5)))


def table_cuts(ic, kind, ns, nb):
    """[lo, hi] per parameter the kernel cuts a star's bounds to: the tables' axes, 10^5 pc, the BC table's A_V axis."""
    mx, bx = ic.model_grid.interp.index_columns, ic.bc_grid.interp.index_columns
    r = lambda ax: (float(ax[0]), float(ax[-1]))
    av = r(bx[3]) if nb else (0.0, 10.0)
    if kind == "track":
        return [r(mx[1]), r(mx[2]), r(mx[0]), (0.0, 1.0e5), av]
    return [r(mx[2])] * ns + [r(mx[0]), r(mx[1]), (0.0, 1.0e5), av]


def host_candidates(desc, kind, ns, star, chunk, seed, cuts=None):
    """The BLOCK candidates lane 0..255 of the workgroup of `star` draw in `chunk` (start_points.h, same arithmetic)."""
    D = ns + 4
    lo = np.array([desc.bound_lo[j] for j in range(D)]); hi = np.array([desc.bound_hi[j] for j in range(D)])
    if cuts is not None:
        lo = np.maximum(lo, [c[0] for c in cuts]); hi = np.minimum(hi, [c[1] for c in cuts])
    a, b = lo.copy(), hi - lo
    if kind == "track":
        a[0], b[0] = np.log(lo[0]), np.log(hi[0] / lo[0])
    qd = ns + 2
    plx = desc.plx_val
    use_plx = bool(desc.has_parallax) and plx > 0 and np.isfinite(1000.0 / plx) if desc.has_parallax else False
    if use_plx:
        d0 = 1000.0 / plx
        rel = min(max(np.sqrt(desc.plx_unc * desc.plx_unc) / plx, 1e-3), 0.3)
        a[qd], b[qd] = d0 * (1.0 - 4.0 * rel), 8.0 * rel * d0
    else:
        dlo = max(lo[qd], 1.0)
        a[qd], b[qd] = np.log(dlo), np.log(hi[qd] / dlo)
    lane = np.arange(BLOCK, dtype=np.uint64)
    p = np.empty((BLOCK, D))
    for c in range((D + 1) // 2):
        r = philox4x32_10(np.full(BLOCK, 4 * chunk + c), np.full(BLOCK, star), lane, np.full(BLOCK, 0x57), seed & 0xFFFFFFFF, seed >> 32)
        u = [(r[0].astype(np.float64) * 2097152.0 + (r[1] & np.uint64(0x1FFFFF)).astype(np.float64)) / 9007199254740992.0,
             (r[2].astype(np.float64) * 2097152.0 + (r[3] & np.uint64(0x1FFFFF)).astype(np.float64)) / 9007199254740992.0]
        p[:, 2 * c] = b[2 * c] * u[0] + a[2 * c]
        if 2 * c + 1 < D:
            p[:, 2 * c + 1] = b[2 * c + 1] * u[1] + a[2 * c + 1]
    if kind == "track":
        p[:, 0] = np.exp(p[:, 0])
    if not use_plx:
        p[:, qd] = np.exp(p[:, qd])
    if ns > 1:
        p[:, :ns] = -np.sort(-p[:, :ns], axis=1)
    return p


@pytest.mark.parametrize("kind,ns,nb,W", [("track", 1, 3, 32), ("iso", 1, 1, 16), ("iso", 2, 6, 32), ("iso", 3, 9, 64), ("track", 1, 12, 8),
                                          # more walkers than a workgroup has lanes (the reference's default is 300, starmodel.py:889);
                                          # 600 x 7 parameters: records beyond the 64 KB of LDS a launch gets without asking
                                          ("iso", 1, 3, 300), ("track", 1, 3, 300), ("iso", 3, 2, 600)])
def test_start_points_are_the_best_of_the_kernels_candidate_stream(kind, ns, nb, W):
    bands = list(ia.grids.KNOWN_BANDS[:nb])
    ic = small_ic(kind, bands)
    S = 60
    cat, _ = ia.synthetic_catalog(ic, S, bands=bands, seed=3 + nb, mag_unc=0.02, with_parallax=True)
    df = cat.df.copy()
    df.loc[df.index[:5], "parallax"] = np.nan                  # no parallax: distance drawn log-uniform
    df.loc[df.index[5:8], "parallax"] = -0.3                   # negative parallax: the same
    cat = StarCatalog(df, bands=bands, props=list(cat.props))
    post = CatalogPosterior.from_catalog(cat, ic, N=ns)
    seed, over = 1234 + nb, 8
    best, lnp, failed = initial_positions(post, W, rng_seed=seed, oversample=over, max_tries=1, method="kernel")
    best2, lnp2, failed2 = initial_positions(post, W, rng_seed=seed, oversample=over, max_tries=1, method="kernel")
    assert bool((best == best2).logical_or(best.isnan() & best2.isnan()).all()) and bool((lnp == lnp2).all())    # deterministic
    best, lnp, failed = best.cpu().numpy(), lnp.cpu().numpy(), failed.cpu().numpy()
    oic = fx.make_oracle_ic(ic)
    chunks = max(1, (over * W + BLOCK - 1) // BLOCK)
    n_ok = 0
    for k in range(0, S, 3):
        desc = cat.model(k, ic, N=ns).model_desc()
        cand = np.concatenate([host_candidates(desc, kind, ns, k, c, seed, table_cuts(ic, kind, ns, nb)) for c in range(chunks)], axis=0)
        want = oic.lnpost(desc, np.ascontiguousarray(cand.T), nthreads=8, parts=False)
        want = np.where(np.isfinite(want), want, -np.inf)
        order = np.argsort(-want, kind="stable")[:W]
        if failed[k]:
            assert np.isfinite(want).sum() < W and np.isnan(best[k]).all()
            continue
        n_ok += 1
        # the kept values are the best W of the stream (ties / last-bit differences may swap neighbours: compare as sets of values)
        assert np.all(np.diff(lnp[k]) <= 0)
        np.testing.assert_allclose(lnp[k], want[order], rtol=1e-9, atol=1e-9)
        # ... at the candidates' positions (exp() of the device and of libm may differ in the last bits)
        np.testing.assert_allclose(best[k], cand[order], rtol=1e-12, atol=1e-12)
        # and the oracle's lnpost of the positions the kernel returned is what it reported
        got = oic.lnpost(desc, np.ascontiguousarray(best[k].T), nthreads=8, parts=False)
        np.testing.assert_allclose(got, lnp[k], rtol=1e-9, atol=1e-8)
        lo = np.array([desc.bound_lo[j] for j in range(ns + 4)]); hi = np.array([desc.bound_hi[j] for j in range(ns + 4)])
        assert np.all(best[k] >= lo - 1e-12) and np.all(best[k] <= hi * (1 + 1e-12))
        if ns > 1:
            assert np.all(np.diff(best[k][:, :ns], axis=1) <= 0)
    assert n_ok >= 10
    post.close()
    ic.release()


def test_more_chunks_are_drawn_until_every_star_has_its_walkers_and_hopeless_stars_are_flagged():
    bands = ["G", "BP", "RP"]
    ic = small_ic("track", bands)
    S, W = 40, 32
    cat, _ = ia.synthetic_catalog(ic, S, bands=bands, seed=11, mag_unc=0.002, with_parallax=True)
    df = cat.df.copy()
    df.loc[df.index[0], "G_mag"] = -40.0            # no model is this bright: every candidate of star 0 is rejected by ... nothing: finite but terrible
    df.loc[df.index[1], "parallax"] = 1e-9          # distance prior's upper bound 2000 / plx beyond the table's reach: still finite
    cat = StarCatalog(df, bands=bands, props=list(cat.props))
    post = CatalogPosterior.from_catalog(cat, ic, N=1)
    one, lnp1, f1 = initial_positions(post, W, rng_seed=5, oversample=1, max_tries=1, method="kernel")       # 256 candidates, one chunk
    many, lnp2, f2 = initial_positions(post, W, rng_seed=5, oversample=1, max_tries=8, method="kernel")
    assert int(f2.sum()) <= int(f1.sum())
    # the first chunk is the same stream: a star that had its W after one chunk keeps exactly those rows
    same = ~f1.cpu().numpy()
    assert np.array_equal(one.cpu().numpy()[same], many.cpu().numpy()[same])
    assert bool(torch_isfinite(lnp2[~f2]).all())
    assert bool(many[f2].isnan().all()) if bool(f2.any()) else True
    post.close()
    ic.release()


def torch_isfinite(x):
    import torch
    return torch.isfinite(x)


def test_fit_on_kernel_start_points_matches_the_framework_version_statistically():
    """Same catalog fitted from start points of the kernel and of the framework version (other random numbers, same
    candidate distribution): the summaries agree within their own scatter and both recover the truth."""
    from isochrones_amd.catalog import fit_stars_gpu
    import os
    bands = ["G", "BP", "RP"]
    ic = ia.synthetic_track(bands=bands)
    S = 400
    cat, truth = ia.synthetic_catalog(ic, S, bands=bands, seed=21, mag_unc=0.01)
    rows = {}
    for method in ("kernel", "torch"):
        os.environ["ISOCHRONES_AMD_START"] = method
        try:
            rows[method] = fit_stars_gpu(cat, ic, np.arange(S), nwalkers=32, nburn=150, niter=100, seed=4)
        finally:
            os.environ.pop("ISOCHRONES_AMD_START", None)
    k, t = rows["kernel"], rows["torch"]
    ok = (k[:, -1] == 1) & (t[:, -1] == 1)
    assert ok.mean() > 0.97
    i_d = 3 * 3            # distance median column (mass, eep, feh, distance, AV) x (median, p16, p84)
    d_true = np.asarray(truth["distance"])[ok]
    for r in (k, t):
        assert np.median(np.abs(r[ok, i_d] - d_true) / d_true) < 0.03
    # the two fits differ by less than the posterior widths
    width = 0.5 * (k[ok, i_d + 2] - k[ok, i_d + 1])
    assert np.median(np.abs(k[ok, i_d] - t[ok, i_d]) / width) < 0.5
    ic.release()
