"""TEST INFRASTRUCTURE — container-only harness that imports the *reference* package in place.

Purpose: generate golden vectors (tests/golden/*.npz) and pin oracle/iso_oracle.c against the
real reference implementation.  ``/root/reference`` does not exist on the GPU box, so nothing
in the product, the ``-m gpu`` tests, ``smoke()`` or ``bench.py`` may import this module.

How the reference is made importable without installing anything (no source is copied):

* ``numba`` is not installed -> an identity shim (``jit`` returns the function unchanged).  The
  numba functions on the path are plain float64 Python, so the shimmed functions compute the
  same IEEE-754 results (none of them uses ``fastmath``).
* ``isochrones/__init__.py`` imports the whole world -> a bare package object with
  ``__path__`` pointing at the reference is registered instead, and sub-modules are imported
  one by one.
* import-only stubs for emcee / corner / configobj / astropy / asciitree (never called on the
  numeric path).
* MIST tables are not available offline -> the grid classes are subclassed and their ``df`` /
  ``interp`` supplied from isochrones_amd.grids' synthetic recipe.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("ISO_REFERENCE_ROOT", "/root/reference")

_installed = False


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "isochrones"))


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims():
    """Register the numba shim, the import-only stubs and the bare ``isochrones`` package."""
    global _installed
    if _installed:
        return
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only mount

    def jit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    class NumbaPendingDeprecationWarning(Warning):
        pass

    class TypingError(Exception):
        pass

    nb = _module("numba", jit=jit, njit=jit, prange=range, uint32=np.uint32, int64=np.int64,
                 float64=np.float64, NumbaPendingDeprecationWarning=NumbaPendingDeprecationWarning,
                 TypingError=TypingError)
    _module("numba.errors", NumbaPendingDeprecationWarning=NumbaPendingDeprecationWarning,
            TypingError=TypingError, NumbaDeprecationWarning=NumbaPendingDeprecationWarning)
    nb.errors = sys.modules["numba.errors"]
    nb.core = _module("numba.core")
    nb.core.errors = _module("numba.core.errors", TypingError=TypingError,
                             NumbaPendingDeprecationWarning=NumbaPendingDeprecationWarning,
                             NumbaDeprecationWarning=NumbaPendingDeprecationWarning)

    for name in ("emcee", "corner"):
        if name not in sys.modules:
            _module(name)

    class _Dummy(dict):
        def __init__(self, *a, **k):
            super().__init__()

    _module("configobj", ConfigObj=_Dummy, Section=_Dummy)

    class _Const:
        def __init__(self, v):
            self.cgs = types.SimpleNamespace(value=v)
            self.value = v

    astropy = _module("astropy")
    astropy.coordinates = _module("astropy.coordinates", SkyCoord=_Dummy)
    astropy.constants = _module("astropy.constants", G=_Const(6.6743e-8), M_sun=_Const(1.98840987e33),
                                R_sun=_Const(6.957e10))
    astropy.units = _module("astropy.units")
    asciitree = _module("asciitree", LeftAligned=_Dummy, Traversal=object)
    asciitree.drawing = _module("asciitree.drawing", BoxStyle=_Dummy, BOX_DOUBLE=None, BOX_BLANK=None)

    pkg = types.ModuleType("isochrones")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "isochrones")]
    sys.modules["isochrones"] = pkg
    _installed = True


def ref(modname: str):
    """Import ``isochrones.<modname>`` from the reference tree."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install_shims()
    return importlib.import_module("isochrones." + modname)


# --------------------------------------------------------------------------------------
# reference objects over synthetic tables
# --------------------------------------------------------------------------------------

def make_ref_dfinterp(grid, axes, columns, index_names=None):
    """A reference DFInterpolator wrapping an already-dense table, without the DataFrame
    round trip (the numeric path only reads these attributes:
    reference isochrones/interp.py:576-588, isochrones/models.py:416-428)."""
    interp_mod = ref("interp")
    obj = interp_mod.DFInterpolator.__new__(interp_mod.DFInterpolator)
    obj.filename = None
    obj.is_full = True
    obj.columns = list(columns)
    obj.n_columns = len(columns)
    obj.grid = np.ascontiguousarray(grid, dtype=float)
    obj.index_columns = tuple(np.asarray(a, dtype=float) for a in axes)
    obj.index_names = index_names or ["ax%d" % i for i in range(len(axes))]
    obj.ndim = len(axes)
    obj.column_index = {c: i for i, c in enumerate(columns)}
    return obj


class _FakeGrid:
    """Duck-typed stand-in for the reference's Grid objects: just what ModelGridInterpolator
    and BasicStarModel read (``interp``, ``get_limits``, ``eep_replaces``, ``bands``, ``df``)."""

    def __init__(self, interp, limits=None, eep_replaces=None, bands=None):
        self.interp = interp
        self._limits = dict(limits or {})
        self.eep_replaces = eep_replaces
        self.bands = list(bands) if bands is not None else None
        self.fehs = None

    def get_limits(self, prop):
        return self._limits[prop]

    @property
    def df(self):
        import pandas as pd
        return pd.DataFrame(columns=self.interp.columns)


def make_ref_ic(kind, model_table, bc_table, limits, eep_bounds):
    """Reference ModelGridInterpolator subclass instance (``kind`` = 'track' | 'iso') bound to
    synthetic tables.  ``model_table`` / ``bc_table`` = (grid, axes, columns)."""
    models = ref("models")
    base = models.EvolutionTrackInterpolator if kind == "track" else models.IsochroneInterpolator
    g, ax, cols = model_table
    bg, bax, bands = bc_table
    mg = _FakeGrid(make_ref_dfinterp(g, ax, cols), limits=limits,
                   eep_replaces="age" if kind == "track" else "mass")
    bcg = _FakeGrid(make_ref_dfinterp(bg, bax, bands), bands=bands)

    class _IC(base):
        grid_type = None
        bc_type = types.SimpleNamespace(default_bands=tuple(bands))

    ic = _IC(bands=list(bands))
    ic._model_grid = mg
    ic._bc_grid = bcg
    ic.eep_bounds = tuple(eep_bounds)
    ic.grid = mg  # ``eep_replaces`` property of the base class reads self.grid
    return ic


# --------------------------------------------------------------------------------------
# in-memory stand-in for pandas' HDF5 I/O
# --------------------------------------------------------------------------------------

class memory_hdf:
    """Context manager: ``DataFrame.to_hdf`` / ``Series.to_hdf`` / ``pd.read_hdf`` keep their objects in a dict
    instead of an HDF5 file.  pytables is not installed, and the reference's grid classes cache every
    intermediate frame through HDF5 (grid.py:103-118, models.py:126-153, mist/models.py:403-435, bc.py:99-118).
    This replaces the *storage*, not any of the reference's arithmetic: what is written is what is read back.
    ``preload(path, key, obj)`` plants a frame as if an earlier run had cached it (and touches the file, because
    the reference tests ``os.path.exists`` before reading)."""

    def __init__(self):
        self.store = {}

    def preload(self, path, key, obj):
        path = os.path.abspath(path)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        open(path, "a").close()
        self.store[(path, key)] = obj.copy()

    def __enter__(self):
        import pandas as pd
        self._saved = (pd.DataFrame.to_hdf, pd.Series.to_hdf, pd.read_hdf)
        store = self.store

        def to_hdf(obj, path, key=None, *args, **kwargs):
            path = os.path.abspath(path)
            open(path, "a").close()
            store[(path, key)] = obj.copy()

        def read_hdf(path, key=None, *args, **kwargs):
            path = os.path.abspath(path)
            if (path, key) in store:
                return store[(path, key)].copy()
            same_file = [v for (p, _), v in store.items() if p == path]
            if key is None and len(same_file) == 1:
                return same_file[0].copy()
            if not same_file:
                raise FileNotFoundError(path)
            raise KeyError(key)

        pd.DataFrame.to_hdf = to_hdf
        pd.Series.to_hdf = to_hdf
        pd.read_hdf = read_hdf
        return self

    def __exit__(self, *exc):
        import pandas as pd
        pd.DataFrame.to_hdf, pd.Series.to_hdf, pd.read_hdf = self._saved
        return False
