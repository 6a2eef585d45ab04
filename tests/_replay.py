"""TEST INFRASTRUCTURE - teacher-forced replay of the device-resident stretch-move sampler on the CPU.

The fused sampler kernels (isochrones_amd/csrc/fast/sampler.h) draw every move's random numbers from
Philox4x32-10 with counter (2*step + half, row_lo, row_hi, 0x51) and key = seed, so a stored chain can be
checked move by move without running a second sampler: for every stored step the proposal is rebuilt on the
host from the *stored* previous state (teacher forcing - an accept/reject flip cannot snowball), evaluated
with the CPU oracle, and the oracle's accept/reject decision and lnpost are compared with what the GPU
stored.  This replaces "the sampler's lnprob equals the HIP batch kernel's lnpost" (HIP vs HIP) by
"the sampler's every move equals the reference algorithm's" (HIP vs oracle).

What the reference does at this point: emcee's stretch move around `StarModel.lnpost`
(isochrones/starmodel.py:951-969; Goodman & Weare 2010: z ~ g(z) on [1/a, a], y = x_j + z (x_k - x_j),
accept with probability min(1, z^(D-1) p(y)/p(x_k))).
"""
import numpy as np

from oracle.cpu_sampler import moves, philox4x32_10, philox_kat  # noqa: F401  (the device sampler's random numbers)


def replay(p0, lnp0, chain, chain_lnp, W, a, seed, step0, lnpost_fn, star_of_block=None, margin=1e-9,
           lnp_rtol=1e-9, lnp_atol=1e-11):
    """Check a stored chain move by move.

    p0 [R, D], lnp0 [R]: state before the first stored step; chain [T, R, D], chain_lnp [T, R]: as
    iso_sampler_run stores them; R = B * W rows = B whole ensembles.  `star_of_block` [B] = the global
    ensemble index of each block (row = star * W + walker keys the random numbers; default 0..B-1).
    lnpost_fn(block_index [n], pars [n, D]) -> oracle lnpost [n].

    Returns a dict of counts; raises AssertionError on any disagreement beyond the tolerances:
      * rejected move: position and lnprob carried over bit for bit;
      * accepted move: stored position == rebuilt proposal (to a few ulp of its operands: FMA contraction), stored
        lnprob == oracle lnpost of it (lnp_rtol / lnp_atol), and that lnpost is finite;
      * the oracle's own decision (log u < (D-1) log z + lnpost(y) - lnpost(x)) equals the GPU's, except
        where |log u - lnq| < margin * (1 + |lnpost(y)| + |lnpost(x)|) (counted as `near_ties`).
    """
    p0, lnp0, chain, chain_lnp = (np.asarray(x, dtype=np.float64) for x in (p0, lnp0, chain, chain_lnp))
    T, R, D = chain.shape
    assert R % W == 0 and p0.shape == (R, D) and lnp0.shape == (R,) and chain_lnp.shape == (T, R)
    B, h = R // W, W // 2
    star_of_block = np.arange(B) if star_of_block is None else np.asarray(star_of_block)
    prev = np.concatenate([p0[None], chain[:-1]], axis=0).reshape(T, B, W, D)        # state before step t
    prev_lnp = np.concatenate([lnp0[None], chain_lnp[:-1]], axis=0).reshape(T, B, W)
    cur = chain.reshape(T, B, W, D)
    cur_lnp = chain_lnp.reshape(T, B, W)
    steps = (step0 + np.arange(T, dtype=np.int64))[:, None, None]
    grow = (star_of_block[:, None] * W + np.arange(W)[None, :])[None]                # global row [1, B, W]
    stats = dict(moves=0, accepted=0, near_ties=0, max_lnp_rel=0.0)
    for half in (0, 1):
        lo = half * h
        rows = np.broadcast_to(grow[:, :, lo:lo + h], (T, B, h))
        j, z, u2 = moves(np.broadcast_to(steps, (T, B, h)), half, rows, h, a, int(seed))
        x = prev[:, :, lo:lo + h, :]
        lold = prev_lnp[:, :, lo:lo + h]
        # half 0 reads the second half as it was before the step; half 1 reads the first half after its update
        other = prev[:, :, h:, :] if half == 0 else cur[:, :, :h, :]
        xj = np.take_along_axis(other, j[..., None], axis=2)
        y = xj + z[..., None] * (x - xj)
        got = cur[:, :, lo:lo + h, :]
        got_lnp = cur_lnp[:, :, lo:lo + h]
        moved = np.any(got != x, axis=-1) | (got_lnp != lold)
        blk = np.broadcast_to(np.arange(B)[None, :, None], (T, B, h))
        lnew = lnpost_fn(blk.reshape(-1), y.reshape(-1, D)).reshape(T, B, h)
        # --- GPU bookkeeping -------------------------------------------------------------------
        assert np.array_equal(got[~moved], x[~moved]) and np.array_equal(got_lnp[~moved], lold[~moved])
        # the device contracts (a - 1) u + 1 and xj + z (x - xj) into FMAs, numpy rounds twice: z may differ by a few
        # ulp and the product by one more, so the bound scales with the operands (a proposal of 1 pc built from walkers at
        # 200 and 1100 pc is only good to 1e-13 pc), not with the result
        ym, gm = y[moved], got[moved]
        scale = (np.abs(xj) + np.abs(z[..., None] * (x - xj)))[moved]
        assert np.all(np.abs(gm - ym) <= 2e-15 * scale + 1e-300), "accepted position is not the proposal"
        lm, sm = lnew[moved], got_lnp[moved]
        assert np.isfinite(sm).all(), "a non-finite proposal was accepted"
        assert np.isfinite(lm).all(), "the oracle rejects (non-finite lnpost) a proposal the GPU accepted"
        err = np.abs(sm - lm)
        tol = lnp_atol + lnp_rtol * np.abs(lm)
        assert np.all(err <= tol), "stored lnprob differs from the oracle: max %g" % float((err - tol).max())
        if lm.size:
            stats["max_lnp_rel"] = max(stats["max_lnp_rel"], float(np.max(err / np.maximum(1.0, np.abs(lm)))))
        # --- the oracle's decision -----------------------------------------------------------------
        with np.errstate(invalid="ignore", divide="ignore"):
            lnq = (D - 1) * np.log(z) + lnew - lold
            logu = np.log(u2)
            acc_o = np.isfinite(lnew) & (logu < lnq)
            near = np.isfinite(lnew) & (np.abs(logu - lnq) < margin * (1.0 + np.abs(lnew) + np.abs(lold)))
        flips = (acc_o != moved) & ~near
        assert not flips.any(), "%d accept/reject decisions differ from the oracle's (first at %s)" % (
            int(flips.sum()), np.argwhere(flips)[0])
        stats["moves"] += int(moved.size)
        stats["accepted"] += int(moved.sum())
        stats["near_ties"] += int(((acc_o != moved) & near).sum())
    return stats
