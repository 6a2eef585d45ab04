"""CPU checks of the test infrastructure in tests/_replay.py: the Philox4x32-10 generator against the
published known-answer vectors (the sampler kernels use the same generator with the same counter layout),
and the teacher-forced replay against a plain-numpy emulation of the kernel's half-step schedule (it must
accept a faithful chain and reject tampered ones)."""
import numpy as np
import pytest

from tests import _replay


def test_philox_known_answers():
    _replay.philox_kat()


def test_move_randoms_are_well_formed():
    j, z, u2 = _replay.moves(np.full(4096, 3), 1, np.arange(4096) + (1 << 33), 16, 2.0, 0x1234567890ABCDEF)
    assert j.min() >= 0 and j.max() < 16 and len(np.unique(j)) == 16
    assert z.min() >= 0.5 and z.max() <= 2.0
    assert u2.min() > 0.0 and u2.max() < 1.0
    # g(z) ~ 1/sqrt(z) on [1/a, a]: E[z] = (a^2 + a + 1)/(3a) = 7/6 for a = 2
    assert abs(z.mean() - 7.0 / 6.0) < 0.02


def _emulate(p0, lnp0, T, W, a, seed, step0, fn, stars):
    """The kernels' schedule in numpy: half 0 moves walkers [0, W/2) against [W/2, W), half 1 the reverse."""
    B, D, h = len(stars), p0.shape[1], W // 2
    pos, lnp = p0.reshape(B, W, D).copy(), lnp0.reshape(B, W).copy()
    chain, clnp = np.empty((T, B * W, D)), np.empty((T, B * W))
    for t in range(T):
        for half in (0, 1):
            lo, olo = half * h, (1 - half) * h
            rows = stars[:, None] * W + lo + np.arange(h)[None]
            j, z, u2 = _replay.moves(np.full(rows.shape, step0 + t), half, rows, h, a, seed)
            x = pos[:, lo:lo + h]
            xj = np.take_along_axis(pos[:, olo:olo + h], j[..., None], axis=1)
            y = xj + z[..., None] * (x - xj)
            blk = np.broadcast_to(np.arange(B)[:, None], (B, h)).reshape(-1)
            lnew = fn(blk, y.reshape(-1, D)).reshape(B, h)
            acc = np.isfinite(lnew) & (np.log(u2) < (D - 1) * np.log(z) + lnew - lnp[:, lo:lo + h])
            pos[:, lo:lo + h] = np.where(acc[..., None], y, x)
            lnp[:, lo:lo + h] = np.where(acc, lnew, lnp[:, lo:lo + h])
        chain[t], clnp[t] = pos.reshape(-1, D), lnp.reshape(-1)
    return chain, clnp


def test_replay_accepts_a_faithful_chain_and_rejects_tampering():
    rng = np.random.default_rng(0)
    W, D, T, a, seed, step0 = 16, 3, 50, 2.0, 99, 7
    stars = np.array([4, 11, 5000000000])           # global ensemble indices (row counter above 2^32 too)
    mu = rng.normal(size=(3, D))

    def fn(blk, p):
        lp = -0.5 * np.sum((p - mu[blk]) ** 2, axis=1)
        return np.where(p[:, 0] > mu[blk, 0] + 2.0, -np.inf, lp)       # a hard edge: non-finite proposals occur

    p0 = (mu[:, None, :] + 0.5 * rng.normal(size=(3, W, D))).reshape(-1, D)
    lnp0 = fn(np.repeat(np.arange(3), W), p0)
    assert np.isfinite(lnp0).all()
    chain, clnp = _emulate(p0, lnp0, T, W, a, seed, step0, fn, stars)
    st = _replay.replay(p0, lnp0, chain, clnp, W, a, seed, step0, fn, star_of_block=stars)
    assert st["moves"] == 3 * W * T and 0.2 < st["accepted"] / st["moves"] < 0.9 and st["near_ties"] == 0
    # wrong seed / wrong step offset / wrong ensemble index: the proposals no longer match
    for kw in (dict(seed=seed + 1), dict(step0=step0 + 1), dict(star_of_block=stars + 1)):
        args = dict(seed=seed, step0=step0, star_of_block=stars)
        args.update(kw)
        with pytest.raises(AssertionError):
            _replay.replay(p0, lnp0, chain, clnp, W, a, args["seed"], args["step0"], fn, star_of_block=args["star_of_block"])
    # a stored lnprob that is off by 1e-6 relative
    bad = clnp.copy()
    prev = np.concatenate([lnp0[None], clnp[:-1]])
    t, r = np.argwhere(clnp != prev)[10]
    bad[t, r] *= 1 + 1e-6
    with pytest.raises(AssertionError):
        _replay.replay(p0, lnp0, chain, bad, W, a, seed, step0, fn, star_of_block=stars)
    # an accepted move recorded as rejected (the oracle would have accepted it by a wide margin)
    bad_c, bad_l = chain.copy(), clnp.copy()
    prev_c = np.concatenate([p0[None], chain[:-1]])
    gain = clnp - prev
    t, r = np.unravel_index(np.argmax(gain), gain.shape)
    bad_c[t, r], bad_l[t, r] = prev_c[t, r], prev[t, r]
    with pytest.raises(AssertionError):
        _replay.replay(p0, lnp0, bad_c, bad_l, W, a, seed, step0, fn, star_of_block=stars)


def test_cpu_sampler_makes_the_moves_the_replay_expects():
    """oracle/cpu_sampler.stretch_fit (the host fit behind bench.py's measured cfg 4 / cfg 5 CPU baselines): its chain
    passes the teacher-forced replay with zero near ties, one-call-per-walker and one-call-per-half-ensemble give the
    same chain bit for bit, a second call continues the counter stream, and `row0` shifts the random numbers as the
    device sampler's global row does."""
    from oracle import cpu_sampler
    rng = np.random.default_rng(3)
    W, D, T, a, seed = 32, 5, 60, 2.0, 0xABCDEF
    mu = np.array([1.0, 355.0, 0.0, 100.0, 0.1])
    sig = np.array([0.05, 10.0, 0.1, 5.0, 0.05])

    def lnpost_rows(rows):
        r = (rows - mu) / sig
        out = -0.5 * np.sum(r * r, axis=1)
        out[rows[:, 4] < 0] = -np.inf                       # a hard bound, as the AV prior
        return out

    p0 = mu + sig * rng.standard_normal((W, D))
    p0[:, 4] = np.abs(p0[:, 4])
    lnp0 = lnpost_rows(p0)
    star = 7
    pos, lnp, chain, clnp, nacc = cpu_sampler.stretch_fit(lnpost_rows, p0, lnp0, T, a=a, seed=seed, row0=star * W, scalar_calls=True)
    pos2, lnp2, chain2, clnp2, nacc2 = cpu_sampler.stretch_fit(lnpost_rows, p0, lnp0, T, a=a, seed=seed, row0=star * W, scalar_calls=False)
    assert np.array_equal(chain, chain2) and np.array_equal(clnp, clnp2) and np.array_equal(nacc, nacc2)
    assert np.array_equal(pos, chain[-1]) and np.array_equal(lnp, clnp[-1])
    st = _replay.replay(p0, lnp0, chain, clnp, W, a, seed, 0, lambda blk, pars: lnpost_rows(pars), star_of_block=np.array([star]))
    assert st["moves"] == W * T and st["near_ties"] == 0 and 0.2 < st["accepted"] / st["moves"] < 0.9
    assert nacc.sum() == st["accepted"]
    # continuing: step0 = T picks the stream up where the first call left it
    _, _, c3, l3, _ = cpu_sampler.stretch_fit(lnpost_rows, pos, lnp, 10, a=a, seed=seed, step0=T, row0=star * W)
    _, _, c4, l4, _ = cpu_sampler.stretch_fit(lnpost_rows, p0, lnp0, T + 10, a=a, seed=seed, row0=star * W)
    assert np.array_equal(c3, c4[T:]) and np.array_equal(l3, l4[T:])
    # another star's rows draw other numbers
    _, _, c5, _, _ = cpu_sampler.stretch_fit(lnpost_rows, p0, lnp0, 5, a=a, seed=seed, row0=(star + 1) * W)
    assert not np.array_equal(c5, chain[:5])
    with pytest.raises(AssertionError):                      # and the replay notices the wrong star
        _replay.replay(p0, lnp0, chain, clnp, W, a, seed, 0, lambda blk, pars: lnpost_rows(pars), star_of_block=np.array([star + 1]))
