"""Table ingest ("next" row f1) against the reference's own grid classes.

tests/golden/ingest.npz was produced by oracle/make_golden.py:run_ingest_cases, which runs the reference's
``MISTEvolutionTrackGrid`` / ``MISTIsochroneGrid`` / ``MISTBolometricCorrectionGrid`` on small synthetic raw
frames (ragged tracks, two photometric systems, three Rv values) and stores inputs and outputs as plain arrays:
the standardised frames (isochrones/models.py:102-124, mist/models.py:81-85,219-223), ``dt_deep``
(mist/models.py:403-435), ``dm_deep`` (models.py:126-153), the ragged age arrays (models.py:171-203), the
NaN-padded dense grids (interp.py:590-614) and the BC table (bc.py:99-118, mist/bc.py:161-233).
Everything here is host-side numpy/pandas: bit-for-bit equality is required."""
import numpy as np
import pandas as pd
import pytest

from isochrones_amd import ingest
from tests import _fixtures as fx


@pytest.fixture(scope="module")
def g():
    return fx.load("ingest")


def _raw(g, pre):
    return pd.DataFrame(g[pre + "_raw"], columns=[str(c) for c in g[pre + "_raw_columns"]])


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("pre,tracks,deriv", [("track", True, "dt_deep"), ("iso", False, "dm_deep")])
def test_standard_columns_dense_grid_and_derivative(g, pre, tracks, deriv):
    raw = _raw(g, pre)
    raw = raw.sample(frac=1.0, random_state=3)                     # row order of the input must not matter
    ref_cols = [str(c) for c in g[pre + "_columns"]]
    df = ingest.standardize_mist_frame(raw, tracks)
    assert set(df.columns) | {deriv} == set(ref_cols)              # the reference orders them through a set()
    assert _same(np.array([list(t) for t in df.index.values]), g[pre + "_index"])
    for c in df.columns:
        assert _same(df[c].to_numpy(), g[pre + "_values"][:, ref_cols.index(c)]), c
    dfi = ingest.model_table_from_raw(raw, tracks)
    assert dfi.grid.shape == g[pre + "_grid"].shape
    for k in range(3):
        assert np.array_equal(dfi.index_columns[k], g[pre + "_axes%d" % k])
    for c in dfi.columns:
        assert _same(dfi.grid[..., dfi.column_index[c]], g[pre + "_grid"][..., ref_cols.index(c)]), c
    d = dfi.grid[..., dfi.column_index[deriv]]
    assert np.isfinite(d).sum() == len(raw) and np.isnan(d).sum() == d.size - len(raw)


def test_ragged_age_arrays(g):
    dfi = ingest.model_table_from_raw(_raw(g, "track"), True)
    want_age, want_dt, want_len = g["track_age_arrays"], g["track_dt_deep_arrays"], g["track_lengths"]
    age, dt, lengths = ingest.ragged_age_arrays(dfi, "age", n_eep=want_age.shape[1], with_dt_deep=True)
    assert np.array_equal(lengths, want_len) and len(set(want_len.tolist())) > 3        # really ragged
    assert _same(age, want_age) and _same(dt, want_dt)
    age2, len2 = ingest.ragged_age_arrays(dfi, "age")
    assert np.array_equal(len2, want_len) and _same(age2, want_age[:, : age2.shape[1]])


def test_bc_frames_to_dense_table(g, tmp_path):
    bands = [str(b) for b in g["bc_bands"]]
    frames = [(g["bc_index"], g["bc_%s_values" % p], [str(c) for c in g["bc_%s_columns" % p]]) for p in ("UBVRIplus", "WISE")]
    bc = ingest.bc_table_from_frames(frames, bands)
    ref_cols = [str(c) for c in g["bc_columns"]]
    assert sorted(bc.columns) == sorted(ref_cols) == sorted(bands)
    assert bc.grid.shape == g["bc_grid"].shape and bc.index_names == ["Teff", "logg", "[Fe/H]", "Av"]
    for k in range(4):
        assert np.array_equal(bc.index_columns[k], g["bc_axes%d" % k])
    for c in bands:
        assert np.array_equal(bc.grid[..., bc.column_index[c]], g["bc_grid"][..., ref_cols.index(c)]), c
    # shuffled rows, the export-file route, and a frame already sliced at Rv = 3.1
    perm = np.random.default_rng(0).permutation(len(g["bc_index"]))
    idx = pd.MultiIndex.from_arrays(g["bc_index"][perm].T, names=["Teff", "logg", "[Fe/H]", "Av", "Rv"])
    files = []
    for p in ("UBVRIplus", "WISE"):
        fr = pd.DataFrame(g["bc_%s_values" % p][perm], index=idx, columns=[str(c) for c in g["bc_%s_columns" % p]])
        files.append(str(tmp_path / (p + ".npz")))
        ingest.export_frame_npz(fr, files[-1])
    bc2 = ingest.bc_table_from_frames(files, bands)
    assert bc2.columns == bc.columns and np.array_equal(bc2.grid, bc.grid)
    keep = g["bc_index"][:, 4] == 3.1
    bc3 = ingest.bc_table_from_frames([(g["bc_index"][keep, :4], f[1][keep], f[2]) for f in frames], bands)
    assert np.array_equal(bc3.grid, bc.grid)
    other = ingest.bc_table_from_frames(frames, bands, rv=2.5)
    assert not np.array_equal(other.grid, bc.grid)
    with pytest.raises(ValueError):
        ingest.bc_table_from_frames(frames, ["J", "z"])            # SDSS frame not given
    with pytest.raises(ValueError):
        drop = np.arange(len(g["bc_index"])) != int(np.flatnonzero(keep)[5])
        ingest.bc_table_from_frames([(f[0][drop], f[1][drop], f[2]) for f in frames], bands)   # not a full product


def test_band_names(g):
    """(system, column) of 35 band names as the reference's get_band resolves them (isochrones/mist/bc.py:165-233)."""
    assert len(g["band_names"]) >= 35
    for b, phot, col in zip(g["band_names"], g["band_phot"], g["band_column"]):
        if str(phot) == "!unresolved":                           # the reference raises ValueError for these
            with pytest.raises(ValueError):
                ingest.mist_band(str(b), {"UBVRIplus": ["Tycho_B"]})
        else:
            # full column names are looked up in the tables at hand (the reference: in its static filter list)
            known = {"UBVRIplus": ["Tycho_B", "Hipparcos_Hp", "Gaia_G_MAW"], "HST_WFPC2": ["WFPC2_F555W"],
                     "SDSSugriz": ["SDSS_g"]}
            assert ingest.mist_band(str(b), known) == (str(phot), str(col)), b
    assert (g["band_phot"] == "!unresolved").sum() >= 2


def test_unknown_band_is_an_error():
    with pytest.raises(ValueError):
        ingest.mist_band("nonsense")


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_other_seeds_and_table_shapes_against_the_reference_itself(seed, tmp_path, monkeypatch):
    """Container-only (needs /root/reference): the reference's grid classes are run on freshly drawn raw frames -
    other random values, other numbers of [Fe/H] / mass / age nodes, other ragged track lengths - and every check of
    this file is repeated on them."""
    import os
    import sys
    if not os.path.isdir("/root/reference/isochrones"):
        pytest.skip("the reference tree is only present in the build container")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(os.path.join(root, "oracle"))
    monkeypatch.setenv("ISO_GOLDEN_OUT", str(tmp_path))
    import make_golden
    fresh = make_golden.run_ingest_cases(seed=seed, save=False)
    fresh = {k: np.asarray(v) for k, v in fresh.items()}
    for pre, tracks, deriv in (("track", True, "dt_deep"), ("iso", False, "dm_deep")):
        test_standard_columns_dense_grid_and_derivative(fresh, pre, tracks, deriv)
    test_ragged_age_arrays(fresh)
    test_bc_frames_to_dense_table(fresh, tmp_path)


def test_ragged_arrays_keep_dt_deep_aligned_with_its_ages():
    """The reference lays ``age`` and ``dt_deep`` out row-aligned (models.py:189-194: both are ``.values`` of the same
    sub-frame): a NaN ``dt_deep`` at a populated age (a single-point track: no derivative) stays next to its age and the
    values after it do not shift; ``lengths`` describes both arrays."""
    from isochrones_amd.interp import DFInterpolator
    fehs, masses, eeps = np.array([-0.5, 0.0]), np.array([0.8, 1.0]), np.arange(1.0, 7.0)
    grid = np.full((2, 2, 6, 3), np.nan)                      # columns: age, dt_deep, Teff
    rng = np.random.default_rng(0)
    lengths = {(0, 0): 6, (0, 1): 4, (1, 0): 1, (1, 1): 5}
    for (i, j), n in lengths.items():
        grid[i, j, :n, 0] = np.sort(rng.uniform(8, 10, n))
        grid[i, j, :n, 1] = rng.uniform(0.1, 1.0, n)
        grid[i, j, :n, 2] = rng.uniform(4000, 7000, n)
    grid[1, 0, 0, 1] = np.nan                                 # single-point track: its derivative is NaN
    grid[0, 1, 1, 1] = np.nan                                 # a NaN derivative in the middle of a track
    dfi = DFInterpolator.from_arrays(grid, [fehs, masses, eeps], ["age", "dt_deep", "Teff"], ["feh", "mass", "eep"])
    age, dt, ln = ingest.ragged_age_arrays(dfi, "age", with_dt_deep=True)
    assert list(ln) == [6, 4, 1, 5]
    assert np.isnan(dt[2, 0]) and np.array_equal(age[2, :1], grid[1, 0, :1, 0])
    assert np.array_equal(age[1, :4], grid[0, 1, :4, 0])
    assert np.array_equal(dt[1, :4], grid[0, 1, :4, 1], equal_nan=True) and np.isnan(dt[1, 1]) and not np.isnan(dt[1, 2])
    for r, n in enumerate(ln):
        assert np.isnan(age[r, n:]).all() and np.isnan(dt[r, n:]).all()
