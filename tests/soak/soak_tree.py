"""Differential soak of the observation-tree kernels (fast and generic) against the oracle: the reference tree
configurations of tests/golden (docs/multiple.ipynb shapes) and, every round, freshly drawn trees (random observation
sets - unresolved bands, a seeing-limited image, an AO image, relative or absolute photometry - with random N / index
assignments: tests/test_tree_cpu.py's generator) x wide, special-value-laden samples."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import oracle as orc          # tests/ = the only place the oracle is used as a checker
from tests import _fixtures as fx
import isochrones_amd as ia
import isochrones_amd.observation as obs_api
from tests.test_tree_cpu import TREE_CASES, make_tree_model, _random_tree_spec, _build_tree
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, fails, evals, rounds, fresh = time.time(), 0, 0, 0, 0
worst = 0.0


def check(ic, mod, cfg):
    global fails, evals, worst
    names = list(mod.param_names)
    lo = np.array([mod.bounds("eep" if nm.startswith("eep") else nm.split("_")[0])[0] for nm in names], float)
    hi = np.array([mod.bounds("eep" if nm.startswith("eep") else nm.split("_")[0])[1] for nm in names], float)
    hi = np.where(np.isfinite(hi), hi, 3000.0)
    n = 40_000
    span = hi - lo
    x = rng.uniform(lo - 0.03 * span, hi + 0.03 * span, size=(n, lo.size))
    half = n // 2
    i = 0
    for s in mod.obs.systems:                       # half of the rows with ordered EEPs per system
        k = mod.obs.Nstars[s]
        x[:half, i:i + k] = -np.sort(-x[:half, i:i + k], axis=1)
        i += 4 + k
    for j in range(lo.size):
        x[j * 8:j * 8 + 6, j] = [lo[j], hi[j], np.nan, np.inf, -np.inf, 0.0]
    # a cluster near a plausible solution so that many rows have a finite posterior
    c = np.array([350.0 if nm.startswith("eep") else {"age": 9.6, "feh": 0.0, "distance": 400.0, "AV": 0.1}[nm.split("_")[0]] for nm in names])
    w = np.array([30.0 if nm.startswith("eep") else {"age": 0.2, "feh": 0.1, "distance": 50.0, "AV": 0.05}[nm.split("_")[0]] for nm in names])
    x[half:half + n // 4] = c + w * rng.standard_normal((n // 4, lo.size))
    oic = fx.make_oracle_ic(ic)
    w_post, w_prior, w_like = orc.tree_lnpost(oic, mod.tree_desc(), x.T.copy(), nthreads=16)
    g_post = mod.lnpost(x)
    f = np.isfinite(w_post) & np.isfinite(g_post)
    if f.any():
        worst = max(worst, float(np.max(np.abs(g_post[f] - w_post[f]) / np.maximum(1, np.abs(w_post[f])))))
    for got, want, what in ((g_post, w_post, "lnpost"), (mod.lnprior(x), w_prior, "lnprior"), (mod.lnlike(x), w_like, "lnlike")):
        if soak.same(got, want, what, cfg) >= 0:
            fails += 1
    evals += 3 * n


def fresh_tree():
    spec = _random_tree_spec(rng)
    n_fine = max(len(srcs) for _, _, _, srcs in spec)
    N = [int(rng.choice([1, 1, 2])) for _ in range(n_fine)]
    index = [0] * n_fine if rng.random() < 0.5 else [int(v) for v in rng.permutation(n_fine)]
    if sum(N[j] for j in range(n_fine) if index[j] == index[0]) > 3:
        N = [1] * n_fine
    kw = dict(N=N if n_fine > 1 else N[0], index=index if n_fine > 1 else index[0])
    if rng.random() < 0.5: kw["parallax"] = (float(rng.choice([2.0, 5.0])), 0.05)
    if rng.random() < 0.4: kw["Teff"] = (float(rng.uniform(5000, 6500)), 100)
    if rng.random() < 0.3: kw["AV"] = (0.2, 0.1)
    return spec, kw


iso_meta = fx.load(TREE_CASES[0])["meta"]
while time.time() - t0 < budget:
    for case in TREE_CASES:
        meta = fx.load(case)["meta"]
        for path in ("auto", "generic"):
            os.environ["ISOCHRONES_AMD_PATH"] = path
            ic, mod = make_tree_model(meta)
            check(ic, mod, dict(case=case, path=path))
            ic.release()
    for _ in range(6):
        spec, kw = fresh_tree()
        for path in ("auto", "generic"):
            os.environ["ISOCHRONES_AMD_PATH"] = path
            ic = fx.make_ic(dict(kind="iso", limits=iso_meta["limits"], eep_bounds=iso_meta["eep_bounds"]))
            try:
                mod = ia.TreeStarModel(ic, obs=_build_tree(obs_api, spec, "fresh"), **kw)
            except Exception:             # a layout the tree builder refuses (as the reference does)
                ic.release()
                break
            check(ic, mod, dict(case="fresh", path=path, spec=str(spec), kw=str(kw)))
            ic.release()
        else:
            fresh += 1
    rounds += 1
print("tree soak: %d rounds x (%d reference configurations + 6 fresh trees, %d built) x 2 kernels, %.3g GPU evaluations, "
      "%d mismatching checks, largest relative difference %.2e" % (rounds, len(TREE_CASES), fresh, evals, fails, worst))
sys.exit(1 if fails else 0)
