"""Randomised differential soak: many model configurations x sample mixes, GPU (every kernel path) vs the
CPU oracle.  Usage on the GPU box: python tests/soak/soak.py [seconds] [seed].  Exit code 1 on any mismatch."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import isochrones_amd as ia
from isochrones_amd import priors as P
from tests import _fixtures as fx

RTOL, ATOL = 1e-9, 1e-10


def random_prior(rng, name, lo, hi):
    fam = rng.choice(["default", "default", "Flat", "Gaussian", "GaussianB", "LogNormal", "PowerLaw", "FlatLog"])
    mid, span = 0.5 * (lo + hi), hi - lo
    if fam == "Flat":
        a = rng.uniform(lo, mid); return P.FlatPrior((a, rng.uniform(a + 0.05 * span, hi)))
    if fam == "Gaussian" and name != "feh":
        return P.GaussianPrior(rng.uniform(lo, hi), span * rng.uniform(0.05, 0.5))
    if fam in ("GaussianB", "Gaussian"):
        a = rng.uniform(lo, mid); return P.GaussianPrior(rng.uniform(lo, hi), span * rng.uniform(0.05, 0.5), bounds=(a, rng.uniform(a + 0.05 * span, hi)))
    if fam == "LogNormal" and lo >= 0:
        return P.LogNormalPrior(float(np.log(max(mid, 1e-3))), rng.uniform(0.2, 1.0))
    if fam == "PowerLaw" and lo >= 0:
        return P.PowerLawPrior(rng.choice([-2.35, -1.0 + 1e-3, 0.3, 2.0]), (max(lo, 1e-3) if rng.random() < 0.5 else lo + 1e-3, hi))
    if fam == "FlatLog" and name == "age":
        a = rng.uniform(lo, mid); return P.FlatLogPrior((a, rng.uniform(a + 0.05 * span, hi)))
    return None


def build(rng):
    kind = rng.choice(["track", "iso"])
    n_stars = 1 if kind == "track" else int(rng.choice([1, 1, 2, 3]))
    nb = int(rng.choice([0, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]))
    # pin the shape from the environment (to chase a failure): SOAK_KIND, SOAK_NSTARS, SOAK_NB
    kind = os.environ.get("SOAK_KIND", kind)
    n_stars = 1 if kind == "track" else int(os.environ.get("SOAK_NSTARS", n_stars))
    nb = int(os.environ.get("SOAK_NB", nb))
    bands = list(ia.grids.DEFAULT_BANDS[:max(nb, 1)])
    all_fehs = np.array([-2.0, -1.5, -1.0, -0.75, -0.5, -0.25, 0.0, 0.25, 0.5])
    # the [Fe/H] axis keeps its end points (the bounds below) and a random subset of the inner nodes
    inner = np.sort(rng.choice(all_fehs[1:-1], int(rng.integers(2, 8)), replace=False))
    def thin(eeps):
        """now and then an EEP axis with missing nodes: not uniform, so no O(1) index and no fused kernels -
        the generic kernel's bisection has to carry the whole model"""
        if rng.random() < 0.12:
            keep = np.ones(eeps.size, bool)
            keep[rng.choice(np.arange(1, eeps.size - 1), int(rng.integers(1, 30)), replace=False)] = False
            return eeps[keep]
        return eeps
    if kind == "track":
        fehs = np.concatenate([[-2.0], inner, [0.5]]); masses = ia.grids.mist_masses()[20:150:int(rng.integers(2, 5))]
        eeps = thin(np.arange(200.0, 200.0 + int(rng.integers(300, 700))))
        ic = ia.synthetic_track(bands=bands, fehs=fehs, masses=masses, eeps=eeps, eep_bounds=(eeps[0], eeps[-1]),
                                limits=dict(mass=(masses[0], masses[-1]), feh=(-2.0, 0.5), age=(5, 10.13)))
        axes = [masses, eeps, fehs]; lo = np.array([masses[0], eeps[0], -2.0, 5.0, 0.0]); hi = np.array([masses[-1], eeps[-1], 0.5, 2000.0, 1.0])
    else:
        ages = ia.grids.mist_log_ages()[40::int(rng.integers(2, 5))]; fehs = np.concatenate([[-2.0], inner, [0.5]])
        eeps = thin(np.arange(150.0, 150.0 + int(rng.integers(300, 750))))
        ic = ia.synthetic_isochrone(bands=bands, ages=ages, fehs=fehs, eeps=eeps, eep_bounds=(eeps[0], eeps[-1]),
                                    limits=dict(age=(ages[0], ages[-1]), feh=(-2.0, 0.5)))
        axes = [eeps] * n_stars + [ages, fehs]
        lo = np.array([eeps[0]] * n_stars + [ages[0], -2.0, 5.0, 0.0]); hi = np.array([eeps[-1]] * n_stars + [ages[-1], 0.5, 2000.0, 1.0])
    obs = {}
    if rng.random() < 0.6: obs["Teff"] = (rng.uniform(4500, 7000), 100.0)
    if rng.random() < 0.5: obs["logg"] = (rng.uniform(3.5, 4.8), 0.1)
    if rng.random() < 0.5: obs["feh"] = (rng.uniform(-0.5, 0.3), 0.15)
    if rng.random() < 0.6: obs["parallax"] = (rng.choice([2.0, 10.0, -0.5]), 0.05)
    if rng.random() < 0.25:
        obs["nu_max"] = (rng.uniform(500, 3500), 100.0)
        if rng.random() < 0.6: obs["delta_nu"] = (rng.uniform(40, 160), 3.0)
    for j in range(nb): obs[bands[j]] = (10.0 + 0.3 * j + rng.normal(0, 0.5), rng.choice([0.002, 0.02, 0.1]))
    kw = {}
    if rng.random() < 0.3: kw["maxAV"] = float(rng.uniform(0.2, 1.5))
    if rng.random() < 0.3: kw["max_distance"] = float(rng.uniform(200, 3000))
    if rng.random() < 0.2: kw["halo_fraction"] = float(rng.uniform(0.0, 0.3))
    mod = ia.BasicStarModel(ic, N=n_stars, **obs, **kw)
    desc_pri = {}
    for name, (a, b) in dict(mass=(0.1, 10.0), age=(6.0, 10.1), feh=(-2.0, 0.5), distance=(0.0, 3000.0), AV=(0.0, 1.0)).items():
        if rng.random() < 0.35:
            pr = random_prior(rng, name, a, b)
            if pr is not None:
                if name == ic.eep_replaces and rng.random() < 0.7:
                    mod._priors["eep"].orig_prior = pr
                else:
                    mod.set_prior(**{name: pr})
                desc_pri[name] = type(pr).__name__
    return dict(kind=kind, n_stars=n_stars, nb=nb, obs=sorted(obs), kw=kw, priors=desc_pri, n_feh=int(fehs.size),
                uniform_eep=bool(np.all(np.diff(eeps) == 1.0))), ic, mod, axes, lo, hi


def samples(rng, axes, lo, hi, n):
    span = hi - lo
    x = rng.uniform(lo - 0.03 * span, hi + 0.03 * span, size=(n, lo.size))
    k = n // 8
    for j in range(lo.size):                       # exact nodes / bounds / specials in every slot
        col = x[:, j]
        if j < len(axes):
            col[j * k:j * k + k // 2] = rng.choice(axes[j], k // 2)
        col[j * k + k // 2:j * k + k // 2 + 8] = [lo[j], hi[j], np.nan, np.inf, -np.inf, 0.0, -0.0, np.nextafter(hi[j], np.inf)]
    ns = lo.size - 4
    if ns > 1:
        x[: n // 2, :ns] = -np.sort(-x[: n // 2, :ns], axis=1)
    return x


def same(got, want, what, cfg):
    np.seterr(all="ignore")
    got, want = np.asarray(got), np.asarray(want)
    bad = np.isnan(got) != np.isnan(want)
    bad |= np.isinf(want) & (got != want)
    bad |= np.isinf(got) & (got != want)
    fin = np.isfinite(want) & np.isfinite(got)
    bad |= fin & (np.abs(got - want) > ATOL + RTOL * np.abs(want))
    if bad.any():
        i = int(np.flatnonzero(bad)[0])
        print("MISMATCH", what, json.dumps(cfg), "row", i, "got", got[i], "want", want[i], "n_bad", int(bad.sum()), flush=True)
        return i
    return -1


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0, configs, evals, fails = time.time(), 0, 0, 0
    n_fin = n_inf = n_nan = 0
    tightest = 0.0
    while time.time() - t0 < budget:
        cfg, ic, mod, axes, lo, hi = build(rng)
        n = 60_000
        x = samples(rng, axes, lo, hi, n)
        oic = fx.make_oracle_ic(ic)
        try:
            w_post, w_prior, w_like = oic.lnpost(mod.model_desc(), x.T.copy(), nthreads=16)
        except Exception as e:
            print("oracle failed", cfg, e); fails += 1; continue
        n_fin += int(np.isfinite(w_post).sum()); n_inf += int(np.isinf(w_post).sum()); n_nan += int(np.isnan(w_post).sum())
        for path in ("auto", "compact", "generic"):
            os.environ["ISOCHRONES_AMD_PATH"] = path
            ic.release(); mod._dirty()
            cfg["path"] = path
            g_post = mod.lnpost(x)
            f = np.isfinite(w_post) & np.isfinite(g_post)
            if f.any():
                tightest = max(tightest, float(np.max(np.abs(g_post[f] - w_post[f]) / np.maximum(1.0, np.abs(w_post[f])))))
            r = same(g_post, w_post, "lnpost", cfg)
            r2 = same(mod.lnprior(x), w_prior, "lnprior", cfg)
            r3 = same(mod.lnlike(x), w_like, "lnlike", cfg)
            for rr in (r, r2, r3):
                if rr >= 0:
                    fails += 1
                    print("   pars", x[rr].tolist(), flush=True)
            evals += 3 * n
            if path != "compact":
                # the batch primitives of the same interpolator: interp_mag (packed BC tables on `auto`) and
                # interp_value of a random column subset (wide pack on `auto`)
                ns = cfg["n_stars"]
                prim = np.ascontiguousarray(np.column_stack([x[:, 0]] + [x[:, ns + j] for j in range(4)]).T)
                prim[3] = np.abs(prim[3]) + 1e-3
                bsel = list(rng.choice(ic.bands, size=int(rng.integers(1, len(ic.bands) + 1)), replace=False))
                wT, wg, wf, wm = oic.interp_mag(prim, [ic.bc_grid.interp.column_index[b] for b in bsel], nthreads=16)
                gT, gg, gf, gm = ic.interp_mag(list(prim), bsel)
                csel = list(rng.choice(len(ic.model_grid.interp.columns), size=int(rng.integers(1, 9)), replace=False))
                order = ic.param_index_order
                xs = [prim[order[0]], prim[order[1]], prim[order[2]]]
                wv = oic.model.interp(xs, csel)
                gv = ic.model_grid.interp(xs, [ic.model_grid.interp.columns[c] for c in csel])
                for got, want, what in ((gT, wT, "Teff"), (gm, wm, "mags"), (gv, wv, "interp_value")):
                    if same(np.ravel(got), np.ravel(want), what, cfg) >= 0:
                        fails += 1
                evals += 2 * n
        configs += 1
        ic.release()
    print("soak: %d configurations, %.3g GPU evaluations, %d mismatching checks, %.0f s; oracle lnpost: %d finite, %d -inf, %d NaN; "
          "largest |gpu - oracle| / max(1, |oracle|) = %.2e" % (configs, evals, fails, time.time() - t0, n_fin, n_inf, n_nan, tightest))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
