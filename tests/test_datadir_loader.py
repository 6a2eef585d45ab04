"""`get_ichrone('mist')` over the reference's data directory.  tests/golden/isochrones_tree/ is a small $ISOCHRONES
tree whose caches the REFERENCE's own classes wrote (oracle/make_golden.py::run_datadir_case: MISTEvolutionTrackGrid /
MISTIsochroneGrid `.interp` -> full_grid<tag>.npz, get_array_grids -> array_grid<tag>.npz; isochrones/grid.py:132-137,
interp.py:590-614, models.py:163-203) plus the one-off exports a user without pytables needs (BC frames, axis vectors);
tests/golden/datadir.npz holds what the reference's MIST interpolators and star models return on those grids."""
import json
import os
import warnings

import numpy as np
import pytest

import isochrones_amd as ia
from isochrones_amd import mist
from tests import _fixtures as fx

TREE = os.path.join(fx.GOLDEN, "isochrones_tree")
RTOL, ATOL = 1e-11, 1e-12


@pytest.fixture()
def tree_env(monkeypatch):
    monkeypatch.setenv("ISOCHRONES", TREE)
    return fx.load("datadir")


@pytest.mark.parametrize("tracks", [True, False])
def test_get_ichrone_loads_the_reference_caches(tree_env, tracks):
    g = tree_env
    pre = "track" if tracks else "iso"
    bands = g["meta"]["bands"]
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # real tables: no "synthetic" warning
        ic = ia.get_ichrone("mist", bands=bands, tracks=tracks)
    assert ic.data_source.startswith(TREE) and type(ic).__name__ == ("EvolutionTrackInterpolator" if tracks else "IsochroneInterpolator")
    m = ic.model_grid.interp
    ref = np.load(mist.mist_paths(TREE, tracks)["full_grid"])
    assert np.array_equal(m.grid, ref["grid"], equal_nan=True) and list(m.columns) == [str(c) for c in ref["columns"]]
    assert tuple(m.grid.shape) == tuple(g[pre + "_grid_shape"])
    for k in range(3):
        assert np.array_equal(m.index_columns[k], g[pre + "_axis%d" % k])
    # the BC table: the reference's Rv = 3.1 slice of the joined frames, its band columns by the short names
    assert set(ic.bc_grid.interp.columns) == set(bands) and ic.bc_grid.interp.grid.shape[:4] == (4, 4, 3, 3)
    # MIST's bounds come with the grid classes (mist/models.py:37), the EEP bounds of the interpolator too (0, 1710)
    assert ic.model_grid.get_limits("feh") == (-4, 0.5) and tuple(ic.eep_bounds) == (0, 1710)


def test_axes_are_also_recovered_from_the_table_itself(tree_env):
    """The reference's cache has no axis vectors; the mass, EEP and age axes are repeated by table columns and read off
    them (what the real MIST grids rely on, with MIST's 15 metallicities for the third): same numbers as the index
    levels of the reference's frame."""
    g = tree_env
    for tracks, pre in ((True, "track"), (False, "iso")):
        d = np.load(mist.mist_paths(TREE, tracks)["full_grid"])
        grid, col = d["grid"], {str(c): j for j, c in enumerate(d["columns"])}
        assert np.array_equal(mist._axis_from_column(grid[..., col["eep"]], 2, "EEP"), g[pre + "_axis2"])
        if tracks:
            assert np.array_equal(mist._axis_from_column(grid[..., col["initial_mass"]], 1, "mass"), g[pre + "_axis1"])
        else:
            assert np.array_equal(mist._axis_from_column(grid[..., col["age"]], 0, "age"), g[pre + "_axis0"])
        with pytest.raises(mist.MistDataNotFound, match="export_axes"):       # 3 metallicities are not MIST's 15
            mist.model_axes(grid, list(col), tracks, axes_file=None)
    # a 15-metallicity table needs no axes file
    fehs = ia.grids.MIST_FEHS
    grid = np.full((15, 2, 4, 3), np.nan)
    grid[..., 0] = np.arange(1.0, 5.0)                       # eep
    grid[:, 0, :, 1], grid[:, 1, :, 1] = 0.8, 1.1            # initial_mass
    grid[..., 2] = 5000.0
    axes, names = mist.model_axes(grid, ["eep", "initial_mass", "Teff"], True)
    assert np.array_equal(axes[0], fehs) and list(axes[1]) == [0.8, 1.1] and list(axes[2]) == [1, 2, 3, 4]


def test_ragged_age_arrays_match_the_references_array_grid_file(tree_env):
    ic = ia.get_ichrone("mist", bands=["G"], tracks=True)
    ref = np.load(mist.mist_paths(TREE, True)["array_grid"])
    age, dt, lengths = ia.ingest.ragged_age_arrays(ic.model_grid.interp, "age", n_eep=ref["age"].shape[1], with_dt_deep=True)
    assert np.array_equal(lengths, ref["lengths"])
    assert np.array_equal(age, ref["age"], equal_nan=True) and np.array_equal(dt, ref["dt_deep"], equal_nan=True)


@pytest.mark.parametrize("tracks", [True, False])
def test_oracle_on_the_loaded_tables_reproduces_the_reference(tree_env, tracks):
    """The CPU checker over the tables this build loaded gives the reference's numbers (interp_value, interp_mag,
    lnprior / lnlike / lnpost of a SingleStarModel) - the loaded tables are the reference's tables in every respect
    the path reads."""
    g = tree_env
    pre = "track" if tracks else "iso"
    meta = g["meta"]
    ic = ia.get_ichrone("mist", bands=meta["bands"], tracks=tracks)
    ic.eep_bounds = tuple(meta[pre + "_eep_bounds"])
    mod = ia.SingleStarModel(ic, **{k: tuple(v) for k, v in meta["obs"].items()})
    oic = fx.make_oracle_ic(ic)
    pars = g[pre + "_pars"]
    post, prior, like = oic.lnpost(mod.model_desc(), np.ascontiguousarray(pars.T))
    fx.assert_close(prior, g[pre + "_lnprior"], RTOL, ATOL, what="lnprior on the loaded tree")
    fx.assert_close(post, g[pre + "_lnpost"], RTOL, ATOL, what="lnpost on the loaded tree")
    ok = np.isfinite(g[pre + "_lnprior"])                     # lnlike is compared where the reference evaluates it
    fx.assert_close(like[ok], g[pre + "_lnlike"][ok], RTOL, ATOL, what="lnlike on the loaded tree")
    assert np.isfinite(g[pre + "_lnpost"]).sum() > 100
    m = ic.model_grid.interp
    cols = [m.column_index[str(c)] for c in g[pre + "_interp_value_cols"]]
    order = (2, 0, 1) if tracks else (1, 2, 0)                # parameter order -> grid axes (models.py:669,696)
    vals = fx_oracle_interp(m, [pars[:, order[k]] for k in range(3)], cols)
    fx.assert_close(vals, g[pre + "_interp_value"], 1e-12, what="interp_value on the loaded tree")
    T, lg, f, mags = oic.interp_mag(np.ascontiguousarray(pars.T), [ic.bc_grid.interp.column_index[b] for b in meta["bands"]])
    fx.assert_close(mags, g[pre + "_mags"], 1e-12, what="interp_mag on the loaded tree")
    fx.assert_close(T, g[pre + "_Teff"], 1e-12, what="Teff")


def fx_oracle_interp(dfi, xs, cols):
    from oracle import oracle as orc
    return orc.OracleTable(dfi.grid, dfi.index_columns).interp(xs, cols)


def test_no_data_directory_means_a_warning_not_silence(monkeypatch, tmp_path):
    monkeypatch.setenv("ISOCHRONES", str(tmp_path))
    with pytest.warns(UserWarning, match="synthetic MIST-shaped tables"):
        ic = ia.get_ichrone("mist", bands=["G"], tracks=True)
    assert ic.model_grid.interp.grid.shape[:3] == (15, 196, 1710) and ic.data_source == "synthetic"
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ia.get_ichrone("synthetic", bands=["G"], tracks=True)            # asked for by name: nothing to warn about
    with pytest.raises(ValueError, match="Unknown stellar models"):
        ia.get_ichrone("parsec")


def test_missing_bc_frames_are_an_error_that_says_what_to_do(monkeypatch, tmp_path):
    import shutil
    root = tmp_path / "iso"
    shutil.copytree(os.path.join(TREE, "mist"), root / "mist")
    with pytest.raises(mist.MistDataNotFound, match="bolometric-correction frame"):
        mist.load_mist(["J"], tracks=True, root=str(root))
    os.makedirs(root / "BC" / "mist")
    (root / "BC" / "mist" / "UBVRIplus.h5").write_bytes(b"")           # the reference's HDF5 store, no pytables here
    try:
        import tables  # noqa: F401
        pytest.skip("pytables is installed")
    except ImportError:
        pass
    with pytest.raises(mist.MistDataNotFound, match="export_frame_npz"):
        mist.load_mist(["J"], tracks=True, root=str(root))
    # get_ichrone does NOT fall back to the synthetic tables here: the caches exist, this build just cannot read one of
    # them - a fit on invented physics that "succeeded" would be worse than the error
    monkeypatch.setenv("ISOCHRONES", str(root))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with pytest.raises(mist.MistDataNotFound, match="export_frame_npz"):
            ia.get_ichrone("mist", bands=["J"], tracks=True)


def test_a_cache_that_cannot_be_used_is_an_error_not_a_synthetic_fit(monkeypatch, tmp_path):
    """A model cache whose [Fe/H] axis is not MIST's (no axes file), or whose axis column is not an axis: the loader's
    error reaches the caller of get_ichrone; only an EMPTY data directory gives the synthetic tables."""
    import shutil
    root = tmp_path / "iso"
    shutil.copytree(TREE, root)
    d = dict(np.load(root / "mist" / "tracks" / "full_grid_v1.2_vvcrit0.4.npz", allow_pickle=False))
    os.remove(root / "mist" / "tracks" / "full_grid_v1.2_vvcrit0.4_axes.npz") if os.path.exists(
        root / "mist" / "tracks" / "full_grid_v1.2_vvcrit0.4_axes.npz") else None
    cols = [str(c) for c in d["columns"]]
    g = d["grid"].copy()
    g[..., cols.index("eep")] += np.linspace(0.0, 0.5, g.shape[1])[None, :, None]      # no longer constant across a node
    np.savez(root / "mist" / "tracks" / "full_grid_v1.2_vvcrit0.4.npz", grid=g, columns=d["columns"])
    monkeypatch.setenv("ISOCHRONES", str(root))
    assert not mist.nothing_there(tracks=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with pytest.raises(mist.MistDataNotFound):
            ia.get_ichrone("mist", bands=["G"], tracks=True)
    empty = tmp_path / "empty"
    empty.mkdir()
    monkeypatch.setenv("ISOCHRONES", str(empty))
    assert mist.nothing_there(tracks=True)


def test_mass_axis_recovered_to_an_ulp_is_snapped_to_the_nominal_list():
    """The reference's interpolated tracks carry lo (1 - d) + hi d in `initial_mass`: a node whose cells differ by an
    ulp is still that node, and nodes that are MIST's mass list to rounding are that list."""
    masses = ia.grids.mist_masses()
    vals = np.repeat(masses[None, :, None], 3, axis=0).repeat(5, axis=2).astype(float)
    vals[1, 7, 2] = np.nextafter(vals[1, 7, 2], np.inf)
    vals[2, 100, 0] = np.nextafter(vals[2, 100, 0], -np.inf)
    got = mist._snap(mist._axis_from_column(vals, 1, "initial mass"), masses)
    assert np.array_equal(got, masses)
    vals[0, 3, 1] *= 1.0 + 1e-9                       # not rounding any more
    with pytest.raises(mist.MistDataNotFound, match="not constant"):
        mist._axis_from_column(vals, 1, "initial mass")


def test_companion_grid_comes_from_the_same_directory(tree_env):
    ic = ia.get_ichrone("mist", bands=["G"], tracks=False)
    trk = ic.track
    assert trk.data_source.endswith(os.path.join("tracks", "full_grid_v1.2_vvcrit0.4.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("tracks", [True, False])
def test_gpu_lnpost_on_the_loaded_tree_matches_the_reference(tree_env, tracks):
    """HIP path on the tables `get_ichrone('mist')` loaded from the reference-written tree vs the reference's own
    lnprior / lnlike / lnpost, interp_value and interp_mag (tests/golden/datadir.npz)."""
    g = tree_env
    pre = "track" if tracks else "iso"
    meta = g["meta"]
    ic = ia.get_ichrone("mist", bands=meta["bands"], tracks=tracks)
    ic.eep_bounds = tuple(meta[pre + "_eep_bounds"])
    mod = ia.SingleStarModel(ic, **{k: tuple(v) for k, v in meta["obs"].items()})
    pars = g[pre + "_pars"]
    fx.assert_close(mod.lnpost(pars), g[pre + "_lnpost"], 1e-9, 1e-10, what="GPU lnpost on the loaded tree")
    fx.assert_close(mod.lnprior(pars), g[pre + "_lnprior"], 1e-9, 1e-10, what="GPU lnprior on the loaded tree")
    vals = ic.interp_value([pars[:, 0], pars[:, 1], pars[:, 2]], [str(c) for c in g[pre + "_interp_value_cols"]])
    fx.assert_close(vals, g[pre + "_interp_value"], 1e-12, what="GPU interp_value on the loaded tree")
    T, lg, f, mags = ic.interp_mag([pars[:, j] for j in range(5)], meta["bands"])
    fx.assert_close(mags, g[pre + "_mags"], 1e-12, what="GPU interp_mag on the loaded tree")
    for p in pars[:5]:                                         # the scalar call forms of the reference's API
        want = g[pre + "_lnpost"][np.where((pars == p).all(axis=1))[0][0]]
        got = mod.lnpost(list(p))
        assert (np.isnan(got) and np.isnan(want)) or got == pytest.approx(want, rel=1e-9, abs=1e-10) or (got == want)
