"""Randomised soak of the device-resident sampler: random model configurations (tests/soak/soak.py's generator:
track / isochrone tables, 1-3 stars, 0-11 bands, spectroscopy, parallax, asteroseismic terms, random priors), random
ensemble sizes, stretch scales, seeds and both kernel forms; every stored move is replayed on the host with the same
Philox counters and evaluated with the CPU oracle (tests/_replay.py).
Usage on the GPU box: python tests/soak/soak_sampler.py [seconds] [seed].  Exit code 1 on any disagreement."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (device memory for the sampler)
import isochrones_amd as ia  # noqa: F401
from isochrones_amd._cabi import IsoError
from isochrones_amd.sampler import FusedEnsembleSampler
from tests import _fixtures as fx
from tests import _replay
from tests.soak import soak


def start_points(rng, mod, lo, hi, W, ball):
    """W rows with a finite lnpost: spread over the bounds, or a ball around the best of them."""
    x = rng.uniform(lo, hi, size=(40_000, lo.size))
    ns = lo.size - 4
    if ns > 1:
        x[:, :ns] = -np.sort(-x[:, :ns], axis=1)
    lp = mod.lnpost(x)
    ok = np.flatnonzero(np.isfinite(lp))
    if ok.size < W:
        return None
    if not ball:
        return x[rng.choice(ok, W, replace=False)]
    best = x[ok[np.argmax(lp[ok])]]
    for shrink in (1e-3, 1e-4, 1e-5, 1e-6):
        p = best + shrink * (hi - lo) * rng.standard_normal((4 * W, lo.size))
        if ns > 1:
            p[:, :ns] = -np.sort(-p[:, :ns], axis=1)
        good = np.flatnonzero(np.isfinite(mod.lnpost(p)))
        if good.size >= W:
            return p[good[:W]]
    return None


def diagnose(p0, chain, clnp, lnp0, W, a, seed):
    """Where a stored accepted position is not the rebuilt proposal (first few cases)."""
    T, R, D = chain.shape
    h = W // 2
    prev = np.concatenate([p0[None], chain[:-1]], axis=0)
    prev_l = np.concatenate([lnp0[None], clnp[:-1]], axis=0)
    steps = np.arange(T, dtype=np.int64)[:, None]
    shown = 0
    for half in (0, 1):
        lo = half * h
        rows = np.broadcast_to(np.arange(lo, lo + h)[None, :], (T, h))
        j, z, u2 = _replay.moves(np.broadcast_to(steps, (T, h)), half, rows, h, a, int(seed))
        x = prev[:, lo:lo + h]
        other = prev[:, h:] if half == 0 else chain[:, :h]
        xj = np.take_along_axis(other, j[..., None], axis=1)
        y = xj + z[..., None] * (x - xj)
        got = chain[:, lo:lo + h]
        moved = np.any(got != x, axis=-1) | (clnp[:, lo:lo + h] != prev_l[:, lo:lo + h])
        bad = moved & np.any(np.abs(got - y) > 2e-15 * (np.abs(xj) + np.abs(z[..., None] * (x - xj))) + 1e-300, axis=-1)
        for t, k in np.argwhere(bad)[:3]:
            print("   step %d half %d walker %d partner %d z %.17g\n     x   %s\n     xj  %s\n     y   %s\n     got %s\n     got - y %s"
                  % (t, half, lo + k, j[t, k] + (h if half == 0 else 0), z[t, k], x[t, k].tolist(), xj[t, k].tolist(),
                     y[t, k].tolist(), got[t, k].tolist(), (got[t, k] - y[t, k]).tolist()), flush=True)
            shown += 1
    return shown


def catalog_run(rng):
    """One random catalog through the catalog (MULTI) instantiation of the sampler kernels: random band lists, bands a
    star was not observed in (NaN -> masked term), spectroscopic columns, with / without parallax, 1-2 stars per system,
    random ensemble size / scale / kernel form; a subset of the stars replayed with each star's own oracle model.
    Returns (stats or None if skipped, cfg)."""
    import pandas as pd
    import torch
    from isochrones_amd.catalog import CatalogPosterior, StarCatalog, initial_positions
    kind = str(rng.choice(["track", "iso"]))
    N = 1 if kind == "track" else int(rng.choice([1, 1, 2]))
    nb = int(rng.integers(1, 9))
    bands = list(ia.grids.DEFAULT_BANDS[:nb])
    S = int(rng.integers(20, 400))
    # (260 / 300 walkers: one ensemble per workgroup - the reference's default shape, whose register-capped form reads a star's
    # block through scalar loads; fewer stars and steps there, the replay is 10 x the moves)
    W = int(rng.choice([8, 16, 32, 64, 64, 260, 300]))
    a = float(rng.choice([1.3, 2.0, 3.0]))
    mode = str(rng.choice(["auto", "stepwise", "persistent-dense"]))
    T = int(rng.integers(8, 40)) if W < 256 else int(rng.integers(4, 10))
    if W >= 256:
        S = min(S, 60)
    sseed = int(rng.integers(0, 2 ** 40))
    plx = bool(rng.random() < 0.7)
    cfg = dict(catalog=True, kind=kind, N=N, nb=nb, S=S, W=W, a=a, mode=mode, T=T, seed=sseed, parallax=plx)
    ic = ia.synthetic_track(bands=bands) if kind == "track" else ia.synthetic_isochrone(bands=bands)
    cat, _ = ia.synthetic_catalog(ic, S, bands=bands, seed=int(rng.integers(0, 2 ** 31)), mag_unc=float(rng.choice([0.005, 0.02, 0.1])),
                                  with_parallax=plx)
    df = cat.df.copy()
    props = list(cat.props)
    if nb > 1:                       # bands some stars were not observed in
        for b in bands[1:]:
            if rng.random() < 0.5:
                df.loc[df.index[rng.random(S) < 0.3], b + "_mag"] = np.nan
    if rng.random() < 0.4:
        df["logg"] = 4.4 + 0.1 * rng.standard_normal(S); df["logg_unc"] = 0.2
        props.append("logg")
    cat = StarCatalog(df, bands=bands, props=props)
    os.environ["ISOCHRONES_AMD_SAMPLER"] = mode
    try:
        post = CatalogPosterior.from_catalog(cat, ic, N=N)
    except IsoError:
        ic.release()
        return None, cfg
    pos, lnp, failed = initial_positions(post, W, rng_seed=int(rng.integers(0, 2 ** 31)))
    good = np.flatnonzero(~failed.cpu().numpy())
    if good.size < 5:
        post.close(); ic.release()
        return None, cfg
    if bool(failed.any()):
        pos[failed] = pos[int(good[0])]
        lnp[failed] = 0.0
    D = post.n_params
    pick = np.sort(rng.choice(good, min(40, good.size), replace=False))
    descs = [cat.model(int(k), ic, N=N).model_desc() for k in pick]
    oic = fx.make_oracle_ic(ic)

    def fn(blk, pars):
        out = np.empty(pars.shape[0])
        for b in range(len(descs)):
            sel = np.flatnonzero(blk == b)
            if sel.size:
                out[sel] = oic.lnpost(descs[b], np.ascontiguousarray(pars[sel].T), nthreads=4, parts=False)
        return out
    sel = torch.as_tensor(pick, device=pos.device)
    p_sel = pos[sel].reshape(-1, D).cpu().numpy()
    l_sel = lnp[sel].reshape(-1).cpu().numpy()
    want0 = fn(np.repeat(np.arange(len(pick)), W), p_sel)
    if not np.allclose(l_sel, want0, rtol=1e-9, atol=1e-9):
        raise AssertionError("catalog start lnpost differs from the oracle: %g" % float(np.max(np.abs(l_sel - want0))))
    fs = FusedEnsembleSampler(post, W, a=a, seed=sseed)
    fs.run_mcmc(pos, T, lnprob0=lnp, store=True)
    ch = fs.chain_steps.reshape(T, S, W, D)[:, sel].reshape(T, -1, D).cpu().numpy()
    cl = fs._lnprob.view(T, S, W)[:, sel].reshape(T, -1).cpu().numpy()
    try:
        st = _replay.replay(p_sel, l_sel, ch, cl, W, a, sseed, 0, fn, star_of_block=pick, lnp_atol=1e-7, margin=1e-8)
    finally:
        fs.close(); post.close(); ic.release()
    return st, cfg


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    os.environ["ISOCHRONES_AMD_PATH"] = "auto"
    t0 = time.time()
    runs = cat_runs = skipped = fails = moves = accepted = ties = 0
    worst = 0.0
    while time.time() - t0 < budget:
        if rng.random() < float(os.environ.get("SOAK_CATALOG_FRACTION", 0.15)):
            cfg = {}
            try:
                st, cfg = catalog_run(rng)
                if st is None:
                    skipped += 1
                    continue
                moves += st["moves"]; accepted += st["accepted"]; ties += st["near_ties"]
                worst = max(worst, st["max_lnp_rel"])
                cat_runs += 1
            except AssertionError as e:
                print("MISMATCH", e, json.dumps(cfg), flush=True)
                fails += 1
            continue
        cfg, ic, mod, axes, lo, hi = soak.build(rng)
        W = int(rng.choice([4, 8, 16, 30, 64, 100, 256]))
        a = float(rng.choice([1.3, 2.0, 3.0]))
        sseed = int(rng.integers(0, 2 ** 40))
        mode = str(rng.choice(["auto", "stepwise"]))
        T = int(rng.integers(10, 60))
        ball = bool(rng.random() < 0.6)
        # pin any of the choices from the environment (to chase a failure): SOAK_W, SOAK_A, SOAK_MODE, SOAK_BALL
        W = int(os.environ.get("SOAK_W", W))
        a = float(os.environ.get("SOAK_A", a))
        mode = os.environ.get("SOAK_MODE", mode)
        ball = bool(int(os.environ.get("SOAK_BALL", int(ball))))
        cfg.update(W=W, a=a, seed=sseed, mode=mode, T=T, ball=ball)
        os.environ["ISOCHRONES_AMD_SAMPLER"] = mode
        try:
            fs = FusedEnsembleSampler(mod, W, a=a, seed=sseed)
        except IsoError:                # a model the fused kernels do not take (no band, non-uniform EEP axis, ...)
            skipped += 1
            ic.release()
            continue
        p0 = start_points(rng, mod, lo, hi, W, ball)
        if p0 is None:
            skipped += 1
            ic.release()
            continue
        oic = fx.make_oracle_ic(ic)
        desc = mod.model_desc()

        def fn(blk, pars, oic=oic, desc=desc):
            return oic.lnpost(desc, np.ascontiguousarray(pars.T), nthreads=16, parts=False)
        lnp0 = fn(None, p0)
        if not np.isfinite(lnp0).all():         # GPU finite, oracle not: the batch soak's business, but count it
            print("MISMATCH start points", json.dumps(cfg), flush=True)
            fails += 1
            ic.release()
            continue
        try:
            fs.run_mcmc(p0, T, lnprob0=lnp0, store=True)
            st = _replay.replay(p0, lnp0, fs.chain_steps.cpu().numpy(), fs._lnprob.cpu().numpy(), W, a, sseed, 0, fn,
                                lnp_atol=1e-7, margin=1e-8)
            moves += st["moves"]; accepted += st["accepted"]; ties += st["near_ties"]
            worst = max(worst, st["max_lnp_rel"])
            if st["near_ties"] > 3:
                print("MISMATCH near ties", st, json.dumps(cfg), flush=True)
                fails += 1
        except AssertionError as e:
            print("MISMATCH", e, json.dumps(cfg), flush=True)
            if "not the proposal" in str(e):
                diagnose(p0, fs.chain_steps.cpu().numpy(), fs._lnprob.cpu().numpy(), lnp0, W, a, sseed)
            fails += 1
        runs += 1
        del fs
        ic.release()
    print("sampler soak: %d single-model runs + %d catalog runs (%d configurations skipped), %.3g moves replayed against the oracle, %.3g accepted, "
          "%d near ties, %d disagreements, largest stored-lnprob difference %.2e (relative), %.0f s"
          % (runs, cat_runs, skipped, moves, accepted, ties, fails, worst, time.time() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
