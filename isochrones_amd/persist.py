"""Saving a fitted model and loading it back (reference: ``StarModel.save_hdf`` / ``load_hdf``,
isochrones/starmodel.py:1205-1317 and 1843-1959).

The reference stores the posterior samples (and derived samples) as pandas HDF5 tables and pickles the
model's keyword measurements, bounds and priors into the table's attributes.  The same content is written
here into one ``.npz`` container (numpy only: pytables / h5py are optional and absent from the ROCm image):

* ``samples`` / ``samples_columns`` and ``derived`` / ``derived_columns``  — the two DataFrames,
* ``obs_*``       — the observation tree's photometry rows (generic model only),
* ``meta``        — JSON: class, name, N / index, measurements, bounds, grid type and bands, evidence,
* ``priors``      — JSON records of the prior objects: class name + constructor parameters (the reference pickles
  them into HDF5 attributes; nothing executable is stored or read here).

``save_hdf`` / ``load_hdf`` keep the reference's names: they write / read this container when the file
name ends in ``.npz`` and otherwise need pytables (pandas ``HDFStore``) for the reference's own layout.
"""
from __future__ import annotations

import json
import os

import numpy as np

FORMAT_VERSION = 2


def _dump_priors(priors):
    """Prior objects as JSON records (class name + constructor parameters).  Nothing executable is stored: earlier
    containers pickled them, and an unpickler restricted by module name can still be steered to arbitrary callables
    through dotted attribute paths, so that format is refused on load."""
    from .priors import prior_to_spec
    return json.dumps({k: (None if v is None else prior_to_spec(v)) for k, v in priors.items()})


def _load_priors(text):
    from .priors import prior_from_spec
    return {k: (None if v is None else prior_from_spec(v)) for k, v in json.loads(text).items()}


def _frame_arrays(df):
    if df is None or len(df.columns) == 0:
        return np.empty((0, 0)), np.array([], dtype=str)
    return np.ascontiguousarray(df.values, dtype=np.float64), np.array([str(c) for c in df.columns])


def _frame(values, columns):
    import pandas as pd
    if len(columns) == 0:
        return None
    return pd.DataFrame(values, columns=[str(c) for c in columns])


def _plain(v):
    if isinstance(v, (tuple, list, np.ndarray)):
        return [_plain(x) for x in v]
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    return v


def _ic_meta(ic):
    return dict(ic_kind="track" if ic.eep_replaces == "age" else "iso", ic_bands=list(ic.bands),
                ic_name=getattr(ic, "name", "mist"))


def _rebuild_ic(meta):
    from .models import get_ichrone
    return get_ichrone("mist", bands=meta["ic_bands"], tracks=meta["ic_kind"] == "track")


def save_model(mod, filename, overwrite=False):
    from .starmodel import BasicStarModel, TreeStarModel
    if os.path.exists(filename) and not overwrite:
        raise IOError("{} exists.  Set the overwrite option.".format(filename))
    has_fit = getattr(mod, "_samples", None) is not None or getattr(mod, "_sampler", None) is not None \
        or getattr(mod, "_nested", None) is not None
    samples = mod.samples if has_fit else None
    meta = dict(format=FORMAT_VERSION, cls=type(mod).__name__, name=mod.name, **_ic_meta(mod.ic))
    meta["bounds"] = {k: (_plain(v) if v is not None else None) for k, v in mod._bounds.items()}
    if getattr(mod, "_nested", None) is not None:
        meta["evidence"] = [float(mod._nested.logz), float(mod._nested.logz_err)]
    arrays = {}
    priors = {k: v for k, v in mod._priors.items() if k != "eep"}
    priors["__eep_orig_prior__"] = mod._priors["eep"].orig_prior
    derived = None
    if isinstance(mod, BasicStarModel):
        meta.update(kind="basic", N=int(mod.N), kwargs={k: _plain(v) for k, v in mod.kwargs.items()},
                    eep_bounds=_plain(mod.eep_bounds), ra=mod.ra, dec=mod.dec,
                    directory=getattr(mod, "_directory", "."))
        derived = mod.derived_samples if has_fit else None
    elif isinstance(mod, TreeStarModel):
        obs = mod.obs
        df = obs.to_df()
        spec = obs._model_spec
        meta.update(kind="tree", N=_plain(spec[2]), index=_plain(spec[3]), leaves=spec[1],
                    spectroscopy={k: {p: _plain(v) for p, v in d.items()} for k, d in obs.spectroscopy.items()},
                    limits={k: {p: _plain(v) for p, v in d.items()} for k, d in obs.limits.items()},
                    parallax={str(k): _plain(v) for k, v in obs.parallax.items()},
                    AV={str(k): _plain(v) for k, v in obs.AV.items()}, obs_name=obs.name,
                    eep_bounds=_plain(mod._priors["eep"].bounds))
        arrays["obs_numeric"] = np.ascontiguousarray(
            df[["resolution", "mag", "e_mag", "separation", "pa"]].values, dtype=np.float64)
        arrays["obs_relative"] = df["relative"].values.astype(bool)
        arrays["obs_name"] = np.array(df["name"].values, dtype=str)
        arrays["obs_band"] = np.array(df["band"].values, dtype=str)
    else:
        raise TypeError("cannot save a {}".format(type(mod).__name__))
    arrays["samples"], arrays["samples_columns"] = _frame_arrays(samples)
    arrays["derived"], arrays["derived_columns"] = _frame_arrays(derived)
    arrays["priors"] = np.array(_dump_priors(priors))
    arrays["meta"] = np.array(json.dumps(meta))
    tmp = filename + ".tmp.npz"
    np.savez(tmp, **arrays)          # uncompressed: zlib on ~1.5 MB of float64 noise costs 20-60 ms per model for nothing
    os.replace(tmp, filename)
    return filename


def load_model(cls, filename, ic=None, name=None):
    import pandas as pd
    from . import starmodel as sm
    from .observation import ObservationTree
    if not os.path.exists(filename):
        raise IOError("{} does not exist.".format(filename))
    with np.load(filename, allow_pickle=False) as z:
        meta = json.loads(str(z["meta"]))
        if meta.get("format") != FORMAT_VERSION:
            raise ValueError("{}: unknown container version {!r}".format(filename, meta.get("format")))
        samples = _frame(z["samples"], z["samples_columns"])
        derived = _frame(z["derived"], z["derived_columns"])
        if z["priors"].dtype.kind != "U":
            raise ValueError("{}: priors are stored in the pickled form of an older container, which is no longer "
                             "read; save the model again".format(filename))
        priors = _load_priors(str(z["priors"]))
        obs_arrays = {k: z[k] for k in z.files if k.startswith("obs_")}
    if ic is None:
        ic = _rebuild_ic(meta)
    if list(ic.bands) != meta["ic_bands"] and not set(meta["ic_bands"]) <= set(ic.bands):
        raise ValueError("the grid holds bands {} but the model was saved with {}".format(list(ic.bands), meta["ic_bands"]))
    target = getattr(sm, meta["cls"], None)
    if cls is not None and target is not None and not issubclass(target, cls) and not issubclass(cls, target):
        raise TypeError("{} holds a {}, not a {}".format(filename, meta["cls"], cls.__name__))
    target = target or cls
    name = meta["name"] if name is None else name
    as_pair = lambda d: {k: tuple(v) for k, v in d.items()}
    if meta["kind"] == "basic":
        kw = as_pair(meta["kwargs"])
        extra = {} if target in (sm.SingleStarModel, sm.BinaryStarModel, sm.TripleStarModel) else dict(N=meta["N"])
        mod = target(ic, name=name, directory=meta.get("directory", "."), eep_bounds=meta["eep_bounds"],
                     ra=meta.get("ra"), dec=meta.get("dec"), **extra, **kw)
    else:
        num = obs_arrays["obs_numeric"]
        df = pd.DataFrame(dict(name=obs_arrays["obs_name"], band=obs_arrays["obs_band"], resolution=num[:, 0],
                               mag=num[:, 1], e_mag=num[:, 2], separation=num[:, 3], pa=num[:, 4],
                               relative=obs_arrays["obs_relative"]))
        if len(df):
            obs = ObservationTree.from_df(df, name=meta.get("obs_name"))
        else:
            obs = ObservationTree(name=meta.get("obs_name"))
        obs.define_models(ic, leaves=meta.get("leaves"), N=meta["N"], index=meta["index"])
        for label, props in meta["spectroscopy"].items():
            obs.add_spectroscopy(label=label, **as_pair(props))
        for label, props in meta.get("limits", {}).items():
            obs.add_limit(label=label, **as_pair(props))
        for system, v in meta["parallax"].items():
            obs.add_parallax(tuple(v), system=int(system))
        for system, v in meta["AV"].items():
            obs.add_AV(tuple(v), system=int(system))
        mod = target(ic, obs=obs, name=name, eep_bounds=meta["eep_bounds"])
    eep_orig = priors.pop("__eep_orig_prior__", None)
    mod.set_prior(**priors)
    if eep_orig is not None:
        mod._priors["eep"].orig_prior = eep_orig
    mod.set_bounds(**{k: tuple(v) for k, v in meta["bounds"].items() if v is not None})
    for k, v in meta["bounds"].items():               # bounds never asked for stay unset, as they were
        if v is None and k in mod._bounds:
            mod._bounds[k] = None
    mod._samples = samples
    if samples is not None:
        mod._fit_kind = "loaded"
    if meta["kind"] == "basic" and derived is not None:
        mod._derived_samples, mod._derived_for = derived, samples
    if "evidence" in meta:
        mod._loaded_evidence = tuple(meta["evidence"])
    return mod


def save_hdf(mod, filename, path="", overwrite=False, append=False):
    """The reference's entry point.  ``*.npz`` -> the container above; anything else is written with pandas'
    HDFStore in the reference's layout (``<path>/samples``, ``<path>/derived_samples``), which needs pytables."""
    if str(filename).endswith(".npz"):
        return save_model(mod, filename, overwrite=overwrite or append)
    try:
        import tables  # noqa: F401
    except ImportError as e:
        raise ImportError("writing HDF5 needs pytables, which is not installed; use a file name ending in .npz "
                          "(or mod.save(...)) for the numpy container") from e
    import pandas as pd
    if os.path.exists(filename):
        with pd.HDFStore(filename) as store:
            present = path in store
        if present:
            if overwrite:
                os.remove(filename)
            elif not append:
                raise IOError("{} in {} exists.  Set either overwrite or append option.".format(path, filename))
    samples = mod.samples
    samples.to_hdf(filename, key=path + "/samples")
    if hasattr(mod, "derived_samples"):
        mod.derived_samples.to_hdf(filename, key=path + "/derived_samples")
    with pd.HDFStore(filename) as store:
        attrs = store.get_storer("{}/samples".format(path)).attrs
        attrs.ic_bands = list(mod.ic.bands)
        attrs.kwargs = getattr(mod, "kwargs", {})
        attrs._bounds = mod._bounds
        attrs._priors = {k: v for k, v in mod._priors.items() if k != "eep"}
        attrs.eep_bounds = tuple(mod._priors["eep"].bounds)
        attrs.name = mod.name
    return filename
