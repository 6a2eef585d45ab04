"""Batched nested sampling — the native counterpart of the reference's ``fit_multinest``
(isochrones/starmodel.py:717-802: ``pymultinest.run(self.mnest_loglike, self.mnest_prior, n_params,
n_live_points=...)`` with ``mnest_loglike = lnpost`` and ``mnest_prior`` = flat box transform).

The algorithm is Skilling's nested sampling with the live points bounded by one enlarged ellipsoid
in the unit cube (the single-mode form of MultiNest's constrained-prior sampling).  What is
GPU-shaped about it: replacement points are not drawn one at a time; every refill proposes a whole
batch inside the current ellipsoid and evaluates it with ONE batched ``loglike`` call (one fused
lnpost launch).  Draws are queued and consumed in order: a queued draw is uniform in its ellipsoid,
hence uniform in any later constrained region {L > L_min} that the ellipsoid still covers, so it
stays valid as L_min rises; draws below the current L_min are discarded.  With 10^10 lnpost/s on
the device the proposal efficiency hardly matters, which is why one generous ellipsoid suffices.

The sampler is host-side numpy around a batched callable, so it is testable without a GPU
(tests/test_nested_cpu.py uses analytic likelihoods); models plug in ``lnpost`` on [N, n_params]."""
from __future__ import annotations

import heapq

import numpy as np


class NestedResult:
    """samples [n, ndim] with ``logl`` and normalised posterior ``weights``; ``logz`` ±
    ``logz_err``; ``equal_weight_samples(n)`` resamples like MultiNest's post_equal_weights."""

    def __init__(self, samples, logl, logwt, logz, logz_err, information, ncall, niter, efficiency, prior_fraction):
        self.samples = samples
        self.logl = logl
        self.logwt = logwt
        self.logz = logz
        self.logz_err = logz_err
        self.information = information
        self.ncall = ncall
        self.niter = niter
        self.efficiency = efficiency
        self.prior_fraction = prior_fraction
        w = np.exp(logwt - logwt.max())
        self.weights = w / w.sum()

    def equal_weight_samples(self, n=None, rng=None):
        rng = rng or np.random.default_rng(0)
        w = self.weights
        if n is None:
            n = max(1, int(1.0 / np.sum(w * w)))          # Kish effective sample size
        # systematic resampling
        pos = (rng.random() + np.arange(n)) / n
        idx = np.minimum(np.searchsorted(np.cumsum(w), pos), w.size - 1)
        rng.shuffle(idx)
        return self.samples[idx], self.logl[idx]


def _bounding_ellipsoid(u, enlarge):
    """mean, A with {x: |A^-1 (x - mean)| <= 1} covering every live point, volume-enlarged."""
    d = u.shape[1]
    mean = u.mean(axis=0)
    cov = np.cov(u, rowvar=False).reshape(d, d) + 1e-14 * np.eye(d)
    try:
        chol = np.linalg.cholesky(cov)
    except np.linalg.LinAlgError:
        chol = np.diag(np.sqrt(np.diag(cov)))
    z = np.linalg.solve(chol, (u - mean).T)
    r2 = float(np.max(np.sum(z * z, axis=0)))
    return mean, chol * np.sqrt(r2) * enlarge ** (1.0 / d)


def _draw_in_ellipsoid(rng, mean, A, m):
    d = mean.size
    z = rng.standard_normal((m, d))
    z *= (rng.random(m) ** (1.0 / d) / np.linalg.norm(z, axis=1))[:, None]
    x = mean + z @ A.T
    return x[np.all((x >= 0.0) & (x <= 1.0), axis=1)]


def nested_sample_batched(loglike, lo, hi, nlive=1000, tol=0.5, enlarge=1.5, remove=None, max_batch=1 << 20,
                          max_calls=int(2e9), seed=0, max_iter=None, propose=None, transform=None):
    """The same integral with the K = ``remove`` lowest live points retired per macro-step (default nlive // 10) and
    all K replacements drawn above the highest of their thresholds — nested sampling with a live-point count that
    drops from nlive to nlive - K + 1 inside a macro-step (shrinkage exp(-1 / n_live) per retired point, as in
    dynamic nested sampling / the final live-point sweep).  Everything inside a macro-step is vectorised, so the
    host loop is ~nlive / K times shorter than in :func:`nested_sample`; the price is a slightly lower proposal
    efficiency (every replacement must beat the batch's highest threshold).  Same result object.

    ``propose(mean, A, want, threshold) -> (u [k, d], logl [k], n_evaluated)``, optional: draws ``want`` points
    uniformly in the ellipsoid ``{mean + A z, |z| <= 1}`` of the unit cube, evaluates them and returns the ones
    inside the cube with logl > threshold (``mean is None``: uniform in the cube).  Models pass a device-resident
    implementation (random numbers, transform, lnpost and the threshold test on the GPU; only the accepted points
    come back), which is what keeps hard posteriors - proposal efficiencies of 1e-3 and below - cheap.

    ``transform(u [n, d]) -> theta [n, d]``, optional: the unit cube -> parameter map when it is not the plain box
    ``lo + u (hi - lo)`` (the reference's generic ``mnest_prior`` also sorts each system's EEPs, starmodel.py:644-656);
    a ``propose`` hook must apply the same map."""
    lo = np.asarray(lo, dtype=float)
    hi = np.asarray(hi, dtype=float)
    d = lo.size
    if nlive < max(20, 4 * (d + 1)):                   # too few points for a macro-step and a sane ellipsoid
        return nested_sample(loglike, lo, hi, nlive=nlive, tol=tol, enlarge=enlarge, max_batch=max_batch,
                             max_calls=max_calls, seed=seed, max_iter=max_iter, transform=transform)
    rng = np.random.default_rng(seed)
    span = hi - lo
    K = max(1, int(nlive // 10 if remove is None else remove))
    K = min(K, nlive - 2 * (d + 1))                    # keep enough points for the bounding ellipsoid
    ncall = 0
    to_pars = transform if transform is not None else (lambda u: lo + u * span)

    def evaluate(u):
        nonlocal ncall
        ncall += u.shape[0]
        ll = np.asarray(loglike(to_pars(u)), dtype=float).reshape(-1)
        return np.where(np.isfinite(ll), ll, -np.inf)

    live_u = np.empty((0, d))
    live_l = np.empty(0)
    tried = 0
    m = max(4 * nlive, 4096)
    while live_l.size < nlive:
        if propose is not None:
            u_ok, l_ok, n_ev = propose(None, None, m, -np.inf)
            ncall += n_ev
        else:
            u = rng.random((m, d))
            ll = evaluate(u)
            ok = ll > -np.inf
            u_ok, l_ok = u[ok], ll[ok]
        live_u = np.vstack([live_u, u_ok])
        live_l = np.concatenate([live_l, l_ok])
        tried += m
        if tried > max_calls or (tried >= 64 * m and live_l.size == 0):
            raise RuntimeError("nested_sample: no point of the prior box has a finite log-likelihood")
        m = min(max_batch, 2 * m)
    frac = live_l.size / tried
    live_u, live_l = live_u[:nlive].copy(), live_l[:nlive].copy()

    dead_u, dead_l, dead_logw = [], [], []
    logz = -np.inf
    logx = 0.0
    it = 0
    eff = 0.2
    while True:
        order = np.argsort(live_l)
        idx = order[:K]
        thr = live_l[idx]                                           # ascending thresholds of this macro-step
        n_at = nlive - np.arange(K)                                 # live points present when each one is retired
        logx_seq = logx - np.cumsum(1.0 / n_at)
        prev = np.concatenate([[logx], logx_seq[:-1]])
        logw = prev + np.log1p(-np.exp(logx_seq - prev))            # X_{j-1} - X_j
        dead_u.append(live_u[idx].copy())
        dead_l.append(thr.copy())
        dead_logw.append(logw + thr)
        logz = np.logaddexp(logz, np.logaddexp.reduce(logw + thr))
        logx = float(logx_seq[-1])
        it += K
        keep = order[K:]
        if live_l.max() + logx < logz + np.log(tol) or (max_iter is not None and it >= max_iter):
            live_u, live_l = live_u[keep], live_l[keep]
            break
        # K replacements above thr[-1], uniform inside the enlarged ellipsoid of the surviving points
        mean, A = _bounding_ellipsoid(live_u[keep], enlarge)
        new_u, new_l = np.empty((0, d)), np.empty(0)
        while new_l.size < K:
            want = int(np.clip((K - new_l.size) / max(eff, 1e-7) * 1.3, 256, max_batch))
            if propose is not None:
                u_ok, l_ok, n_ev = propose(mean, A, want, float(thr[-1]))
                ncall += n_ev
            else:
                cand = _draw_in_ellipsoid(rng, mean, A, want)
                if cand.shape[0] == 0:
                    continue
                cl = evaluate(cand)
                okc = cl > thr[-1]
                u_ok, l_ok = cand[okc], cl[okc]
            eff = 0.5 * eff + 0.5 * max(l_ok.size, 1) / want
            new_u = np.vstack([new_u, u_ok])
            new_l = np.concatenate([new_l, l_ok])
            if ncall > max_calls:
                raise RuntimeError("nested_sample: max_calls exceeded (efficiency %.2e)" % eff)
        live_u[idx] = new_u[:K]
        live_l[idx] = new_l[:K]
    # the remaining live points, retired one by one without replacement
    n_left = live_l.size
    if n_left:
        order = np.argsort(live_l)
        n_at = n_left - np.arange(n_left)
        logx_seq = logx - np.cumsum(1.0 / n_at)
        logx_seq[-1] = -np.inf                                      # the last point takes all the remaining volume
        prev = np.concatenate([[logx], logx_seq[:-1]])
        with np.errstate(divide="ignore"):
            logw = prev + np.log1p(-np.exp(logx_seq - prev))
        dead_u.append(live_u[order])
        dead_l.append(live_l[order])
        dead_logw.append(logw + live_l[order])
        logz = np.logaddexp(logz, np.logaddexp.reduce(logw + live_l[order]))
    samples = to_pars(np.vstack(dead_u))
    logl = np.concatenate(dead_l)
    logwt = np.concatenate(dead_logw) - logz
    w = np.exp(logwt)
    info = max(float(np.sum(w * logl) - logz), 0.0)                 # H = sum w_i ln L_i / Z - ln Z
    return NestedResult(samples, logl, logwt, float(logz + np.log(frac)), float(np.sqrt(info / nlive)), info, ncall, it,
                        (it + nlive) / max(ncall, 1), frac)


def nested_sample(loglike, lo, hi, nlive=1000, tol=0.5, enlarge=1.5, batch=None, max_batch=1 << 20, max_calls=int(2e9),
                  seed=0, max_iter=None, propose=None, transform=None):
    """Nested sampling of ``exp(loglike(theta))`` under the flat prior on the box [lo, hi].

    ``propose``: optional device-side proposal hook, see :func:`nested_sample_batched` (only the draws above
    the current threshold are queued).

    loglike : callable, [n, ndim] float64 array -> [n] (NaN / -inf = zero likelihood)
    tol     : stop when the live points could add less than ``tol`` to logZ (MultiNest's
              evidence_tolerance, default 0.5)
    enlarge : volume enlargement of the bounding ellipsoid
    Returns a :class:`NestedResult`."""
    rng = np.random.default_rng(seed)
    lo = np.asarray(lo, dtype=float)
    hi = np.asarray(hi, dtype=float)
    d = lo.size
    span = hi - lo
    ncall = 0
    to_pars = transform if transform is not None else (lambda u: lo + u * span)

    def evaluate(u):
        nonlocal ncall
        ncall += u.shape[0]
        ll = np.asarray(loglike(to_pars(u)), dtype=float).reshape(-1)
        return np.where(np.isfinite(ll), ll, -np.inf)

    # live points: prior draws with non-zero likelihood; the zero-likelihood part of the box only rescales Z
    live_u = np.empty((0, d))
    live_l = np.empty(0)
    tried = 0
    m = max(4 * nlive, 4096)
    while live_l.size < nlive:
        u = rng.random((m, d))
        ll = evaluate(u)
        ok = ll > -np.inf
        # whole batches only: the finite fraction stays an unbiased estimate
        live_u = np.vstack([live_u, u[ok]])
        live_l = np.concatenate([live_l, ll[ok]])
        tried += m
        if tried > max_calls or (tried >= 64 * m and live_l.size == 0):
            raise RuntimeError("nested_sample: no point of the prior box has a finite log-likelihood")
        m = min(max_batch, 2 * m)
    frac = live_l.size / tried
    live_u, live_l = live_u[:nlive].copy(), live_l[:nlive].copy()

    heap = [(live_l[i], i) for i in range(nlive)]
    heapq.heapify(heap)
    dead_u, dead_l, dead_logw = [], [], []
    logz = -np.inf
    h_acc = 0.0                                   # sum of w_i L_i logL_i / Z bookkeeping via running update
    logx_prev = 0.0
    queue_u, queue_l = np.empty((0, d)), np.empty(0)
    qpos = 0
    accepted_since_refill = nlive
    drawn_since_refill = nlive
    it = 0
    lmax = float(live_l.max())
    while True:
        lmin, i = heap[0]
        logx = -(it + 1) / nlive
        logw = logx_prev + np.log1p(-np.exp(logx - logx_prev))      # X_{i-1} - X_i
        # logZ and information (Skilling 2006, eq. 16-17)
        logz_new = np.logaddexp(logz, logw + lmin)
        term = np.exp(logw + lmin - logz_new) * lmin
        h_acc = term + np.exp(logz - logz_new) * (h_acc + logz) - logz_new if np.isfinite(logz) else term - logz_new
        logz = logz_new
        dead_u.append(live_u[i].copy())
        dead_l.append(lmin)
        dead_logw.append(logw + lmin)
        logx_prev = logx
        it += 1
        if lmax + logx < logz + np.log(tol) or (max_iter is not None and it >= max_iter):
            break
        # replacement with L > lmin
        while True:
            while qpos < queue_l.size and not (queue_l[qpos] > lmin):
                qpos += 1
            if qpos < queue_l.size:
                break
            # refill: bounding ellipsoid of the live points, one batched evaluation
            eff = max(accepted_since_refill, 1) / max(drawn_since_refill, 1)
            want = int(np.clip((nlive // 4) / max(eff, 1e-6), 1024, max_batch)) if batch is None else int(batch)
            mean, A = _bounding_ellipsoid(live_u, enlarge)
            accepted_since_refill, drawn_since_refill = 0, want
            if propose is not None:
                queue_u, queue_l, n_ev = propose(mean, A, want, float(lmin))
                ncall += n_ev
                qpos = 0
                if queue_l.size == 0:
                    continue
            else:
                cand = _draw_in_ellipsoid(rng, mean, A, want)
                if cand.shape[0] == 0:
                    continue
                queue_u, queue_l, qpos = cand, evaluate(cand), 0
            if ncall > max_calls:
                raise RuntimeError("nested_sample: max_calls exceeded (efficiency %.2e)" % eff)
        live_u[i] = queue_u[qpos]
        live_l[i] = queue_l[qpos]
        lmax = max(lmax, float(queue_l[qpos]))
        heapq.heapreplace(heap, (queue_l[qpos], i))
        qpos += 1
        accepted_since_refill += 1

    # remaining live points share the last shell equally
    logw_live = logx_prev - np.log(nlive)
    order = np.argsort(live_l)
    for j in order:
        logz_new = np.logaddexp(logz, logw_live + live_l[j])
        term = np.exp(logw_live + live_l[j] - logz_new) * live_l[j]
        h_acc = term + np.exp(logz - logz_new) * (h_acc + logz) - logz_new
        logz = logz_new
        dead_u.append(live_u[j].copy())
        dead_l.append(live_l[j])
        dead_logw.append(logw_live + live_l[j])
    samples = to_pars(np.array(dead_u))
    logl = np.array(dead_l)
    logwt = np.array(dead_logw) - logz
    info = max(float(h_acc), 0.0)
    return NestedResult(samples, logl, logwt, float(logz + np.log(frac)), float(np.sqrt(info / nlive)), info, ncall, it,
                        (it + nlive) / max(ncall, 1), frac)
