"""``starfit``: fit the star described by a folder's ``star.ini`` and keep the result next to it
(reference: isochrones/starfit.py:14-175, the function behind ``scripts/starfit``).

For every requested multiplicity the fitted model is stored as ``<models>_starmodel_<mult>.npz`` in the
folder; an existing file is loaded instead of refitting unless ``overwrite`` is set.  Plots and the
command-line wrapper of the reference are not provided.
"""
from __future__ import annotations

import logging
import os
import time

from . import ini
from .models import get_ichrone
from .priors import FlatPrior
from .starmodel import BasicStarModel

NSTARS = {"single": 1, "binary": 2, "triple": 3}


def starfit(folder, multiplicities=("single",), models="mist", use_emcee=False, overwrite=False, verbose=False,
            starmodel_type=None, ini_file="star.ini", bands=None, feh_prior=None, ichrone=None, logger=None,
            **fit_kwargs):
    """-> the model of the last multiplicity fitted (or loaded).  ``fit_kwargs`` go to ``mod.fit``
    (``fit_multinest`` by default, ``fit_mcmc`` with ``use_emcee=True``)."""
    Mod = BasicStarModel if starmodel_type is None else starmodel_type
    logger = logger or logging.getLogger("isochrones_amd.starfit")
    folder = os.path.abspath(folder)
    name = os.path.basename(folder)
    ini_path = ini_file if os.path.isabs(ini_file) else os.path.join(folder, ini_file)
    mod = None
    for mult in multiplicities:
        if mult not in NSTARS:
            raise ValueError("multiplicity must be one of {}, got {!r}".format(sorted(NSTARS), mult))
        filename = os.path.join(folder, "{}_starmodel_{}.npz".format(models, mult))
        start = time.time()
        if os.path.exists(filename) and not overwrite:
            mod = Mod.load_hdf(filename, name=name, ic=ichrone)
            logger.info("%s exists.  Use overwrite=True to refit.", os.path.basename(filename))
            continue
        if ichrone is None:
            found = ini.get_bands(ini_path)
            ichrone = get_ichrone(models, bands=list(dict.fromkeys(list(bands or []) + found)))
        scalars, _ = ini.read_ini(ini_path)
        extra = {} if "N" in scalars else dict(N=NSTARS[mult])
        mod = Mod.from_ini(ichrone, folder, ini_file=ini_file, use_emcee=use_emcee, name=name, **extra)
        if feh_prior == "flat":
            mod.set_prior(feh=FlatPrior(tuple(mod.ic.model_grid.get_limits("feh"))))
        mod.fit(verbose=verbose, **fit_kwargs)
        mod.save_hdf(filename, overwrite=True)
        logger.info("%s starfit successful for %s in %.1f s.", mult, name, time.time() - start)
    return mod


def batch_starfit(folders, rank=None, world=None, **kwargs):
    """``starfit`` over many folders, split over the ranks of the job the way the reference's
    ``scripts/batch_starfit`` splits a list file over SLURM tasks (line NR, 1-based, goes to task NR % P,
    scripts/batch_starfit:60-62).  ``folders``: a list of folders or the path of a file with one folder per
    line.  ``rank`` / ``world`` default to the default ``torch.distributed`` group (one process per GPU), or
    to a single process.  Returns ``{folder: model or the exception that stopped its fit}`` for this rank's share;
    a failing folder is logged and does not stop the rest (the reference logs and carries on, starfit.py:165-169).

    ``batched=True``: the MI355X form of the same job — all folders of this rank whose ``star.ini`` describes an
    unresolved 1-3 star system are fitted *together* by the device-resident ensemble sampler (S stars x W walkers
    per launch, as ``fit_catalog``), and every folder still gets its own stored model with its own posterior
    samples.  Extra keywords: ``nwalkers``, ``nburn``, ``niter``, ``seed``, ``save`` (default True)."""
    from .catalog import shard_of
    if isinstance(folders, (str, os.PathLike)):
        base = os.path.dirname(os.path.abspath(folders))
        with open(folders) as f:
            folders = [ln.strip() for ln in f if ln.strip() and not ln.lstrip().startswith("#")]
        folders = [p if os.path.isabs(p) else os.path.join(base, p) for p in folders]
    if rank is None or world is None:
        try:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
        except ImportError:          # pragma: no cover
            on = False
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    logger = kwargs.get("logger") or logging.getLogger("isochrones_amd.starfit")
    out = {}
    if kwargs.pop("batched", False):
        mine = [f for i, f in enumerate(folders) if shard_of(i, world) == rank]
        return _batch_starfit_device(mine, logger=logger, **{k: v for k, v in kwargs.items() if k != "logger"})
    for i, folder in enumerate(folders):
        if shard_of(i, world) != rank:
            continue
        try:
            out[folder] = starfit(folder, **kwargs)
        except KeyboardInterrupt:
            raise
        except Exception as e:       # noqa: BLE001 - one bad star must not end the batch
            logger.error("starfit calculation failed for %s: %s", folder, e)
            out[folder] = e
    return out


_BATCHABLE = ("Teff", "logg", "feh", "parallax")


def _batch_starfit_device(folders, multiplicities=("single",), models="mist", ini_file="star.ini", ichrone=None,
                          overwrite=False, bands=None, nwalkers=32, nburn=150, niter=100, seed=0, save=True,
                          chunk_stars=512, logger=None, **unused):
    import numpy as np
    import pandas as pd
    from .catalog import StarCatalog, fit_stars_gpu
    from .starmodel import BasicStarModel
    logger = logger or logging.getLogger("isochrones_amd.starfit")
    if unused:
        logger.warning("batch_starfit(batched=True) samples with the ensemble sampler; ignored keywords: %s",
                       ", ".join(sorted(unused)))
    out = {}
    # ---- read every ini once; anything the batch cannot express is fitted on its own afterwards ----
    recs, alone = [], []
    for folder in folders:
        folder = os.path.abspath(folder)
        path = ini_file if os.path.isabs(ini_file) else os.path.join(folder, ini_file)
        try:
            kw = BasicStarModel.ini_keywords(path)
        except Exception as e:       # noqa: BLE001
            logger.error("cannot read %s: %s", path, e)
            out[folder] = e
            continue
        recs.append((folder, kw))
    if ichrone is None and recs:
        found = list(dict.fromkeys(list(bands or []) + [k for _, kw in recs for k in kw if ini.parse_band(k)]))
        ichrone = get_ichrone(models, bands=found)
    ic = ichrone
    for mult in multiplicities:
        if mult not in NSTARS:
            raise ValueError("multiplicity must be one of {}, got {!r}".format(sorted(NSTARS), mult))
        N = NSTARS[mult]
        todo = []
        for folder, kw in recs:
            filename = os.path.join(folder, "{}_starmodel_{}.npz".format(models, mult))
            if os.path.exists(filename) and not overwrite:
                out[folder] = BasicStarModel.load_hdf(filename, name=os.path.basename(folder), ic=ic)
                continue
            odd = [k for k in kw if k not in _BATCHABLE and k not in ic.bands and k not in ("ra", "dec")]
            bad_shape = [k for k in kw if (k in _BATCHABLE or k in ic.bands) and np.size(kw[k]) != 2]
            if odd or bad_shape or not any(k in ic.bands for k in kw):
                alone.append((folder, mult))          # maxAV, N, nu_max, ... : the per-folder route knows them all
                continue
            todo.append((folder, kw, filename))
        if not todo:
            continue
        bset = [b for b in ic.bands if any(b in kw for _, kw, _ in todo)]
        props = [p for p in _BATCHABLE if any(p in kw for _, kw, _ in todo)]
        cols = {}
        for b in bset:
            cols[b + "_mag"] = [kw.get(b, (np.nan, np.nan))[0] for _, kw, _ in todo]
            cols[b + "_mag_unc"] = [kw.get(b, (np.nan, np.nan))[1] for _, kw, _ in todo]
        for p_ in props:
            cols[p_] = [kw.get(p_, (np.nan, np.nan))[0] for _, kw, _ in todo]
            cols[p_ + "_unc"] = [kw.get(p_, (np.nan, np.nan))[1] for _, kw, _ in todo]
        df = pd.DataFrame(cols, index=[os.path.basename(f) for f, _, _ in todo])
        cat = StarCatalog(df, bands=bset, props=props)
        start = time.time()
        rows, chain, lnps = fit_stars_gpu(cat, ic, np.arange(len(todo)), N=N, nwalkers=nwalkers, nburn=nburn,
                                          niter=niter, seed=seed, return_chains=True)
        logger.info("%d %s fits sampled together in %.2f s", len(todo), mult, time.time() - start)
        ok = rows[:, -1] == 1
        S, W, T, D = chain.shape
        for c0 in range(0, S, chunk_stars):                       # derived columns for a slab of stars per launch
            c1 = min(S, c0 + chunk_stars)
            flat = chain[c0:c1].reshape(-1, D)
            host = flat.cpu().numpy()
            lnp_host = lnps[c0:c1].reshape(c1 - c0, -1).cpu().numpy()
            if N == 1:                                            # every model column + magnitude, one call per slab
                derived = ic(*[host[:, j] for j in range(5)])
                dcols = list(derived.columns)
                dvals = derived.values
                dvals_plus = np.column_stack([dvals, 1000.0 / host[:, 3], host[:, 3], host[:, 4]])
            for k in range(c0, c1):
                folder, kw, filename = todo[k]
                if not ok[k]:
                    out[folder] = RuntimeError("no walker with a finite lnpost could be drawn for {}".format(folder))
                    logger.error("starfit calculation failed for %s: no valid starting point", folder)
                    continue
                mod = BasicStarModel.from_ini(ic, folder, ini_file=ini_file, N=N, use_emcee=True)
                lo, hi = (k - c0) * W * T, (k - c0 + 1) * W * T
                names = list(mod.param_names)
                if N == 1:
                    extra = [j for j, c in enumerate(dcols) if c not in names]
                    sdf = pd.DataFrame(np.column_stack([host[lo:hi], lnp_host[k - c0], dvals[lo:hi][:, extra]]),
                                       columns=names + ["lnprob"] + [dcols[j] for j in extra])
                    # derived_samples of a single star = the same table + parallax, distance, AV (starmodel.py:1653-1707)
                    keep = [j for j, c in enumerate(dcols) if c not in ("distance", "AV")]
                    ddf = pd.DataFrame(dvals_plus[lo:hi][:, keep + [len(dcols), len(dcols) + 1, len(dcols) + 2]],
                                       columns=[dcols[j] for j in keep] + ["parallax", "distance", "AV"])
                    mod._derived_samples, mod._derived_for = ddf, sdf
                else:
                    sdf = pd.DataFrame(np.column_stack([host[lo:hi], lnp_host[k - c0]]), columns=names + ["lnprob"])
                mod._samples, mod._fit_kind = sdf, "mcmc-batched"
                if save:
                    mod.save_hdf(filename, overwrite=True)
                out[folder] = mod
    for folder, mult in alone:
        try:
            out[folder] = starfit(folder, multiplicities=[mult], models=models, ini_file=ini_file, ichrone=ic,
                                  overwrite=overwrite, use_emcee=True, nwalkers=max(nwalkers, 2 * 8), nburn=nburn,
                                  niter=niter, seed=seed)
        except KeyboardInterrupt:
            raise
        except Exception as e:       # noqa: BLE001
            logger.error("starfit calculation failed for %s: %s", folder, e)
            out[folder] = e
    return out


def write_catalog_ini(catalog, ic=None, root=".", N=1, nest_directories=True, clobber=True):
    """Lay a catalog out as the folder tree ``starfit`` / ``batch_starfit`` walk: ``<root>/[<prefix>/]<star>/star.ini``
    (reference: StarCatalog.write_ini, catalog.py:141-158).  With ``nest_directories`` the stars are grouped under the
    leading characters of their names, one character per factor of 100 stars, so no directory grows beyond a few
    hundred entries.  ``clobber`` removes a star's existing folder (and whatever was fitted in it) first.  Returns the
    absolute star folders in catalog order."""
    import shutil
    prefix_len = 0
    size = len(catalog)
    while nest_directories and size >= 100:
        size //= 100
        prefix_len += 1
    folders = []
    for mod in catalog.iter_models(ic, N=N):
        name = str(mod.name)
        parent = os.path.join(root, name[:prefix_len]) if nest_directories else root
        target = os.path.abspath(os.path.join(parent, name))
        if clobber and os.path.isdir(target):
            shutil.rmtree(target)
        mod.write_ini(root=parent)
        folders.append(target)
    return folders

