"""TEST INFRASTRUCTURE - the stretch-move ensemble sampler on the CPU, driven by the oracle's lnpost.

What the reference does for a fit: ``StarModel.fit_mcmc`` hands ``self.lnpost`` to ``emcee.EnsembleSampler`` and
runs it (isochrones/starmodel.py:886-972; emcee's stretch move, Goodman & Weare 2010: z ~ g(z) on [1/a, a],
y = x_j + z (x_k - x_j), accept with probability min(1, z^(D-1) p(y) / p(x_k)); one Python call of lnpost per walker
and half-step, emcee's default ``vectorize=False``).  emcee is not in this image, and its numpy Mersenne-Twister
stream could not be reproduced on a GPU anyway; this module is the same move with the random numbers the device
sampler uses (Philox4x32-10, counter = (2*step + half, row_lo, row_hi, 0x51), key = seed:
isochrones_amd/csrc/fast/sampler.h), so a CPU fit and a GPU fit started from the same positions with the same seed
make the same moves and their chains can be compared step by step.

Used by: tests/_replay.py (the random numbers), bench.py's cpu_baseline legs (cfg 4: the 256-walker x 5000-step fit
timed on the host; cfg 5: a subsample of the catalog's stars fitted one after another, as scripts/batch_starfit does
per worker).  Never imported by the product.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011): uint32 counter arrays, scalar key -> 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK32 for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK32
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def philox_kat():
    """Known-answer vectors of Philox4x32-10 from the Random123 distribution (kat_vectors)."""
    out = philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(x[0]) for x in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xFFFFFFFF
    out = philox4x32_10([f], [f], [f], [f], f, f)
    assert [int(x[0]) for x in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    out = philox4x32_10([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0)
    assert [int(x[0]) for x in out] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def moves(step, half, rows, h, a, seed):
    """Random numbers of the moves of global rows `rows` at (step, half): partner index j in [0, h),
    stretch factor z, acceptance uniform u2 - the same arithmetic as stretch_move()."""
    rows = np.asarray(rows, dtype=np.uint64)
    step = np.asarray(step, dtype=np.uint64)
    r0, r1, r2, r3 = philox4x32_10(np.uint64(2) * step + np.uint64(half), rows & MASK32, rows >> np.uint64(32),
                                   np.full(rows.shape, 0x51, dtype=np.uint64), seed & 0xFFFFFFFF, seed >> 32)
    j = ((r0 * np.uint64(h)) >> np.uint64(32)).astype(np.int64)
    u1 = (r1.astype(np.float64) + (r2 & np.uint64(0xFFFF)).astype(np.float64) * (1.0 / 65536.0)) * (1.0 / 4294967296.0)
    u2 = (r3.astype(np.float64) + (r2 >> np.uint64(16)).astype(np.float64) * (1.0 / 65536.0) + 0.5 / 65536.0) * (
        1.0 / 4294967296.0)
    zr = (a - 1.0) * u1 + 1.0
    return j, zr * zr / a, u2


def stretch_fit(lnpost_rows, p0, lnp0, nsteps, a=2.0, seed=0, step0=0, row0=0, store=True, scalar_calls=True):
    """Run one ensemble of W = len(p0) walkers for `nsteps` iterations on the host.

    lnpost_rows(pars [n, D]) -> lnpost [n] (the oracle).  With ``scalar_calls`` the proposals of a half-step are
    evaluated one row per call (how emcee drives the reference's lnpost); otherwise one call per half-step
    (emcee's ``vectorize=True``).  ``row0`` = global row of walker 0 (star * W for star `star` of a catalog): it
    keys the random numbers exactly as the device sampler does.

    Returns (pos [W, D], lnp [W], chain [nsteps, W, D] | None, chain_lnp [nsteps, W] | None, n_accepted [W])."""
    pos = np.array(p0, dtype=np.float64, copy=True)
    lnp = np.array(lnp0, dtype=np.float64, copy=True)
    W, D = pos.shape
    h = W // 2
    assert W % 2 == 0 and lnp.shape == (W,)
    chain = np.empty((nsteps, W, D)) if store else None
    clnp = np.empty((nsteps, W)) if store else None
    nacc = np.zeros(W, dtype=np.int64)
    rows = row0 + np.arange(W, dtype=np.int64)
    new = np.empty(h)
    for t in range(nsteps):
        for half in (0, 1):
            lo, clo = half * h, (1 - half) * h
            j, z, u2 = moves(np.full(h, step0 + t, dtype=np.int64), half, rows[lo:lo + h], h, a, int(seed))
            x = pos[lo:lo + h]
            xj = pos[clo:clo + h][j]
            y = xj + z[:, None] * (x - xj)
            if scalar_calls:
                for k in range(h):
                    new[k] = lnpost_rows(y[k:k + 1])[0]
            else:
                new[:] = lnpost_rows(y)
            with np.errstate(invalid="ignore", divide="ignore"):
                acc = np.isfinite(new) & (np.log(u2) < (D - 1) * np.log(z) + new - lnp[lo:lo + h])
            pos[lo:lo + h][acc] = y[acc]
            lnp[lo:lo + h][acc] = new[acc]
            nacc[lo:lo + h] += acc
        if store:
            chain[t] = pos
            clnp[t] = lnp
    return pos, lnp, chain, clnp, nacc
