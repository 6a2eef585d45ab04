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
* column standardisation of the raw MIST frames: rename to the short names, ``Teff``, ``Mbol``, ``radius``,
  ``density`` (isochrones/models.py:102-109), surface ``feh`` (isochrones/mist/models.py:81-85), track
  ``age = log10(star_age)`` (isochrones/mist/models.py:219-223) -> :func:`standardize_mist_frame`.
* bolometric corrections: per-photometric-system frames indexed (Teff, logg, [Fe/H], Av, Rv), joined, the
  requested bands renamed to their short names (``get_band``, isochrones/mist/bc.py:165-233), the ``Rv = 3.1``
  slice (mist/bc.py:161-163), dense ``is_full`` table (isochrones/bc.py:99-118) -> :func:`bc_table_from_frames`.
All of these are pinned to the reference's own classes run on synthetic raw frames
(tests/golden/ingest.npz, oracle/make_golden.py:run_ingest_cases; tests/test_ingest_golden.py).
The HDF5 stores of the reference need pytables, which this image lacks; a user who has it can
pass the DataFrame straight to ``DFInterpolator(df)``, or export the frames once with
:func:`export_frame_npz` and load them here with numpy alone.
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


def ragged_age_arrays(dfi: DFInterpolator, column="age", n_eep=None, with_dt_deep=False):
    """(age_grid [n0*n1, n_eep], lengths [n0*n1]) of a track table, as the reference's ``get_array_grids``
    (isochrones/models.py:171-203): row i = the rows track i has, left-justified, NaN beyond ``lengths[i]``.
    ``n_eep`` widens the arrays (the reference allocates MIST's 1710 columns whatever the table holds);
    ``with_dt_deep=True`` also returns the ``dt_deep`` column laid out the same way."""
    # which (track, EEP) rows a track has is decided once, from the age column (a row the ragged frame lacked is NaN
    # there in the padded dense table), and both columns are laid out with that one mask: the reference copies
    # ``subdf[age].values`` and ``subdf.dt_deep.values`` row-aligned (models.py:189-194), so a NaN ``dt_deep`` at a
    # populated age - a single-point track, whose derivative is undefined - stays next to its age instead of shifting
    # the rest of the row
    n0, n1, ne = dfi.grid.shape[:3]
    populated = ~np.isnan(dfi.grid[..., dfi.column_index[column]]).reshape(n0 * n1, ne)
    lengths = populated.sum(axis=1).astype(np.int64)

    def lay_out(col):
        rows = dfi.grid[..., dfi.column_index[col]].reshape(n0 * n1, ne)
        out = np.full((n0 * n1, max(ne, n_eep or 0)), np.nan)
        for r in range(rows.shape[0]):
            out[r, : lengths[r]] = rows[r][populated[r]]
        return out

    ages = lay_out(column)
    if with_dt_deep:
        return ages, lay_out("dt_deep"), lengths
    return ages, lengths


# ---- raw MIST frames -> the standard columns ---------------------------------------------------
_MSUN_G = 1.98840987e33      # astropy.constants.M_sun.cgs / R_sun.cgs as the reference reads them (models.py:19-21)
_RSUN_CM = 6.957e10

#: raw MIST column -> short name (reference prop_map: models.py:41-52, mist/models.py:24-32,195-204)
MIST_COLUMN_MAP = {"EEP": "eep", "star_mass": "mass", "initial_mass": "initial_mass", "log_Teff": "logTeff",
                   "log_g": "logg", "log_L": "logL"}
STANDARD_TRACK_COLUMNS = ("eep", "feh", "mass", "initial_mass", "radius", "density", "logTeff", "Teff", "logg", "logL",
                          "Mbol", "delta_nu", "nu_max", "phase", "interpolated", "star_age", "age")
STANDARD_ISO_COLUMNS = ("eep", "age", "feh", "mass", "initial_mass", "radius", "density", "logTeff", "Teff", "logg",
                        "logL", "Mbol", "delta_nu", "nu_max", "phase")


def standardize_mist_frame(raw, tracks):
    """The reference's ``get_df`` for a raw MIST frame (one row per model point, the column names of the
    ``.track.eep`` / ``.iso`` files plus the nominal ``initial_feh`` / ``feh`` of the file): MultiIndex
    (initial_feh, initial_mass, EEP) for tracks or (log10_isochrone_age_yr, feh, EEP) for isochrones, sorted, and
    the standard columns.  The derivative column (``dt_deep`` / ``dm_deep``) is added on the dense table by
    :func:`add_dt_deep` / :func:`add_dm_deep`."""
    import pandas as pd
    index_cols = ("initial_feh", "initial_mass", "EEP") if tracks else ("log10_isochrone_age_yr", "feh", "EEP")
    df = raw.sort_values(by=list(index_cols))
    index = pd.MultiIndex.from_arrays([df[c].to_numpy(float) for c in index_cols], names=index_cols)
    col = {MIST_COLUMN_MAP.get(c, c): df[c].to_numpy(float) for c in df.columns}
    if not tracks:
        col["age"] = col["log10_isochrone_age_yr"]
    col["Teff"] = 10 ** col["logTeff"]
    col["Mbol"] = 4.74 - 2.5 * col["logL"]
    col["radius"] = 10 ** col["log_R"]
    col["density"] = col["mass"] * _MSUN_G / (4.0 / 3 * np.pi * (col["radius"] * _RSUN_CM) ** 3)
    col["feh"] = col["log_surf_z"] - np.log10(col["surface_h1"]) - np.log10(0.0181)     # surface [Fe/H]
    if tracks:
        col["age"] = np.log10(col["star_age"])
    names = STANDARD_TRACK_COLUMNS if tracks else STANDARD_ISO_COLUMNS
    return pd.DataFrame({c: col[c] for c in names}, index=index)


def model_table_from_raw(raw, tracks):
    """Raw MIST frame -> dense NaN-padded table with its derivative column: what the reference's
    ``MISTEvolutionTrackGrid().interp`` / ``MISTIsochroneGrid().interp`` hold."""
    dfi = DFInterpolator(standardize_mist_frame(raw, tracks), is_full=False)
    return add_dt_deep(dfi) if tracks else add_dm_deep(dfi)


# ---- bolometric corrections --------------------------------------------------------------------
#: photometric systems of the MIST BC tables (one ``<system>.h5`` frame each in the reference's store)
MIST_PHOT_SYSTEMS = ("UBVRIplus", "WISE", "CFHT", "DECam", "GALEX", "JWST", "LSST", "PanSTARRS", "SkyMapper", "SPITZER",
                     "UKIDSS", "SDSSugriz", "HST_ACSWF", "HST_ACSHR", "HST_WFC3", "HST_WFPC2")
_SHORT = {"K": "2MASS_Ks", "kep": "Kepler_Kp", "Kepler": "Kepler_Kp", "Kp": "Kepler_Kp", "TESS": "TESS",
          "Bp": "Gaia_BP_DR2Rev", "Rp": "Gaia_RP_DR2Rev"}
_SHORT.update({b: "SDSS_" + b for b in "ugriz"})
_SHORT.update({b: "Bessell_" + b for b in "UBVRI"})
_SHORT.update({b: "2MASS_" + b for b in ("J", "H", "Ks")})
_SHORT.update({b: "WISE_" + b for b in ("W1", "W2", "W3", "W4")})
_SHORT.update({b: "Gaia_%s_DR2Rev" % b for b in ("G", "BP", "RP")})


def mist_band(b, table_columns=None):
    """(photometric system, table column) of a band name, resolved as the reference's
    ``MISTBolometricCorrectionGrid.get_band`` does (isochrones/mist/bc.py:165-233): the short names (J, H, K, G, BP,
    RP, V, g, W1, Kepler, TESS, ...) map to their catalogue columns; ``<System>_<band>`` with an all-letter system
    name goes to that system (PanSTARRS columns are spelled ``PS_<band>``, ``UK_`` / ``UKIRT_`` mean UKIDSS); any
    other name must be a column of one of the tables - ``table_columns`` = {system: column names} of the frames at
    hand (the reference keeps a static list of every MIST filter for this last step; here the tables themselves
    are asked)."""
    import re
    if b in _SHORT:
        col = _SHORT[b]
        phot = "SDSSugriz" if col.startswith("SDSS_") else "WISE" if col.startswith("WISE_") else "UBVRIplus"
        return phot, col
    m = re.match("([a-zA-Z]+)_([a-zA-Z_]+)", b)
    if m and m.group(1) in MIST_PHOT_SYSTEMS:
        return m.group(1), ("PS_" + m.group(2)) if m.group(1) == "PanSTARRS" else m.group(0)
    if m and m.group(1) in ("UK", "UKIRT"):
        return "UKIDSS", "UKIDSS_" + m.group(2)
    for system, cols in (table_columns or {}).items():
        if b in cols:
            return system, b
    raise ValueError("MIST grids cannot resolve band {}!".format(b))


def export_frame_npz(df, filename):
    """Write a pandas frame with a MultiIndex (a BC frame of the reference's ``<phot>.h5``, or a model grid) as
    plain arrays: ``index`` [n_rows, n_levels], ``index_names``, ``values`` [n_rows, n_cols], ``columns``.
    Run it where pandas can read the HDF5 store; everything downstream needs numpy only."""
    np.savez(filename, index=np.array([list(t) for t in df.index.values], dtype=float),
             index_names=np.array([str(n) for n in df.index.names]), values=np.asarray(df.values, dtype=float),
             columns=np.array([str(c) for c in df.columns]))


def bc_table_from_frames(frames, bands, rv=3.1):
    """Dense BC table of the requested ``bands`` from per-photometric-system frames.

    ``frames``: list of ``(index [n, 5] or [n, 4], values [n, k], columns [k])`` tuples or of
    :func:`export_frame_npz` file names; index levels (Teff, logg, [Fe/H], Av[, Rv]).  Frames are joined on the
    index, the bands' columns renamed to the short names, rows with ``Rv == rv`` kept, and the result laid out as
    the full product grid [nT, ng, nf, nA, n_bands] (the reference's ``is_full = True`` reshape, bc.py:27,
    interp.py:598-600; an incomplete product raises).  Column order: frame order, as the reference's
    ``pd.concat(axis=1)``; look columns up by name."""
    loaded = []
    for fr in frames:
        if isinstance(fr, (str, bytes)) or hasattr(fr, "__fspath__"):
            d = np.load(fr, allow_pickle=False)
            fr = (d["index"], d["values"], [str(c) for c in d["columns"]])
        loaded.append((np.asarray(fr[0], float), np.asarray(fr[1], float), [str(c) for c in fr[2]]))
    present = {"frame%d" % k: fr[2] for k, fr in enumerate(loaded)}
    want = {}
    for b in bands:
        want.setdefault(mist_band(b, present)[1], []).append(b)
    key_rows = None
    cols, names = [], []
    for index, values, columns in loaded:
        if index.shape[1] == 5:
            keep = index[:, 4] == rv
            index, values = index[keep, :4], values[keep]
        elif index.shape[1] != 4:
            raise ValueError("BC frames are indexed (Teff, logg, [Fe/H], Av[, Rv])")
        order = np.lexsort(index.T[::-1])
        index, values = index[order], values[order]
        if key_rows is None:
            key_rows = index
        elif index.shape != key_rows.shape or not np.array_equal(index, key_rows):
            raise ValueError("BC frames do not share one index")
        for j, c in enumerate(columns):
            for short in want.get(c, ()):
                cols.append(values[:, j])
                names.append(short)
    missing = [b for b in bands if b not in names]
    if missing:
        raise ValueError("bands %s not found in the given frames" % missing)
    axes = [np.unique(key_rows[:, k]) for k in range(4)]
    shape = tuple(a.size for a in axes)
    if key_rows.shape[0] != int(np.prod(shape)):
        raise ValueError("the BC frame is not a full product grid (%d rows, levels %s)" % (key_rows.shape[0], shape))
    grid = np.stack(cols, axis=1).reshape(shape + (len(names),))
    return DFInterpolator.from_arrays(np.ascontiguousarray(grid), axes, names, ["Teff", "logg", "[Fe/H]", "Av"])


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
