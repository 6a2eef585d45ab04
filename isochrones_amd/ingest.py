"""Table ingest helpers ("next" row f1 of SURVEY 8): turn arrays / DataFrames / the reference's
``full_grid*.npz`` cache into device-resident interpolators, and compute the derived columns the
posterior needs.

What the reference does (and where):
* ``DFInterpolator._make_grid`` caches the dense grid as ``np.savez(filename, grid=, columns=)``
  (isochrones/interp.py:590-614; file names ``full_grid{tag}.npz``, models.py:163-165).  The axis
  vectors are *not* in that file (they come from ``df.index.levels``), so they are passed in here.
* ``dt_deep`` = d log10(star_age) / d EEP along each (feh, mass) track, ``np.gradient`` over the
  track's populated points (isochrones/mist/models.py:403-435); ``dm_deep`` = d initial_mass /
  d EEP along each (age, feh) isochrone (isochrones/models.py:126-153).
* ragged age arrays for ``get_eep``: ``age_grid[n_feh*n_mass, n_eep]`` + ``lengths``
  (isochrones/models.py:171-203).
The HDF5 stores of the reference need pytables, which this image lacks; a user who has it can
pass the DataFrame straight to ``DFInterpolator(df)``.
"""
from __future__ import annotations

import numpy as np

from .interp import DFInterpolator


def load_full_grid_npz(filename, index_columns, index_names=None):
    """DFInterpolator from a reference ``full_grid*.npz`` (keys ``grid``, ``columns``) + its axes."""
    d = np.load(filename, allow_pickle=False)
    grid = np.ascontiguousarray(d["grid"], dtype=float)
    columns = [str(c) for c in d["columns"]]
    return DFInterpolator.from_arrays(grid, index_columns, columns, index_names)


def populated_prefix(values):
    """Number of leading non-NaN entries along the last axis (tracks/isochrones are populated
    contiguously from their first point)."""
    ok = ~np.isnan(values)
    return np.where(ok.all(axis=-1), values.shape[-1], np.argmin(ok, axis=-1))


def deep_derivative(values, eeps):
    """np.gradient(values, eep) over the populated (non-NaN) points of every last-axis row;
    NaN elsewhere.  Rows with < 2 populated points stay NaN."""
    values = np.asarray(values, float)
    eeps = np.asarray(eeps, float)
    out = np.full(values.shape, np.nan)
    flat_v = values.reshape(-1, values.shape[-1])
    flat_o = out.reshape(-1, values.shape[-1])
    for r in range(flat_v.shape[0]):
        ok = ~np.isnan(flat_v[r])
        if ok.sum() >= 2:
            flat_o[r, ok] = np.gradient(flat_v[r, ok], eeps[ok])
    return out


def add_dt_deep(dfi: DFInterpolator, age_column="star_age", log=True):
    """Append ``dt_deep`` to a track table (index (feh, mass, EEP))."""
    v = dfi.grid[..., dfi.column_index[age_column]]
    with np.errstate(invalid="ignore", divide="ignore"):
        v = np.log10(v) if log else v
    dfi.add_column(deep_derivative(v, dfi.index_columns[2]), "dt_deep")
    return dfi


def add_dm_deep(dfi: DFInterpolator, mass_column="initial_mass"):
    """Append ``dm_deep`` to an isochrone table (index (age, feh, EEP))."""
    v = dfi.grid[..., dfi.column_index[mass_column]]
    dfi.add_column(deep_derivative(v, dfi.index_columns[2]), "dm_deep")
    return dfi


def ragged_age_arrays(dfi: DFInterpolator, column="age"):
    """(age_grid [n0*n1, n_eep], lengths [n0*n1]) of a track table, as the reference's
    ``get_array_grids``: row i = the populated ages of track i, NaN beyond ``lengths[i]``."""
    v = dfi.grid[..., dfi.column_index[column]]
    n0, n1, ne = v.shape
    ages = np.ascontiguousarray(v.reshape(n0 * n1, ne))
    lengths = populated_prefix(ages).astype(np.int64)
    out = np.full_like(ages, np.nan)
    for r in range(ages.shape[0]):
        out[r, : lengths[r]] = ages[r, : lengths[r]]
    return out, lengths


# ---- self-contained table files ---------------------------------------------------------------
# The reference keeps the BC tables only as HDF5 (bc.py:89-118) and its ``full_grid*.npz`` cache
# without the axis vectors.  A machine that can read those (pandas + pytables) exports each table once
# with ``save_table_npz(DFInterpolator(df), "mist_tracks.npz")``; the GPU box then needs numpy only.

def save_table_npz(dfi: DFInterpolator, filename):
    """grid + axes + column / index names of any DFInterpolator in one ``.npz``."""
    arrays = {"grid": dfi.grid, "columns": np.array(list(dfi.columns)), "index_names": np.array(list(dfi.index_names))}
    for d, ax in enumerate(dfi.index_columns):
        arrays["axis%d" % d] = np.asarray(ax, dtype=float)
    np.savez(filename, **arrays)


def load_table_npz(filename):
    d = np.load(filename, allow_pickle=False)
    ndim = d["grid"].ndim - 1
    return DFInterpolator.from_arrays(np.ascontiguousarray(d["grid"], dtype=float), [d["axis%d" % k] for k in range(ndim)],
                                      [str(c) for c in d["columns"]], [str(c) for c in d["index_names"]])


def interpolator_from_tables(model, bc, tracks, bands=None, limits=None, eep_bounds=None):
    """ModelGridInterpolator over two tables (DFInterpolator objects or ``save_table_npz`` files): the model
    table indexed (feh, mass, EEP) for ``tracks=True`` or (log age, feh, EEP) otherwise, with the reference's
    column names (Teff, logg, feh, Mbol, age|mass, dt_deep|dm_deep, ...), and the BC table indexed
    (Teff, logg, [Fe/H], Av) with one column per band."""
    from .models import (BolometricCorrectionGrid, EvolutionTrackGrid, EvolutionTrackInterpolator, IsochroneGrid,
                         IsochroneInterpolator)
    model = load_table_npz(model) if isinstance(model, (str, bytes)) or hasattr(model, "__fspath__") else model
    bc = load_table_npz(bc) if isinstance(bc, (str, bytes)) or hasattr(bc, "__fspath__") else bc
    bands = list(bands) if bands is not None else list(bc.columns)
    grid_cls, ic_cls = (EvolutionTrackGrid, EvolutionTrackInterpolator) if tracks else (IsochroneGrid, IsochroneInterpolator)
    return ic_cls(grid_cls(model, limits=limits), BolometricCorrectionGrid(bc, bands=bands), bands=bands,
                  eep_bounds=eep_bounds)
