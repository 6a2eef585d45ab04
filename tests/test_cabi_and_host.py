"""CPU tests: the C-ABI library loads and exports every symbol the header declares (no compute
calls without a GPU), and the host-side logic (priors, descriptor packing, table building)."""
import ctypes
import os
import re

import numpy as np
import pytest

import isochrones_amd as ia
from isochrones_amd import _cabi, priors
from tests import _fixtures as fx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from isochrones_amd.csrc import build as hip_build
    return hip_build.build()


def test_header_symbols_exported(built_lib):
    header = open(os.path.join(ROOT, "include", "isochrones_amd.h")).read()
    declared = set(re.findall(r"\b(iso_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_cabi.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(built_lib)
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.iso_version.restype = ctypes.c_char_p
    assert b"isochrones_amd" in lib.iso_version()


def test_library_was_built_from_the_sources_in_the_tree(built_lib):
    """The digest stored next to libiso_hip.so is the digest of the sources, headers and flags in the tree: a stale
    prebuilt binary cannot pass for a fresh build (file times are not consulted)."""
    from isochrones_amd.csrc import build as hip_build
    assert hip_build.built_digest() == hip_build.source_digest()
    assert hip_build.built_library_sha256() == hip_build.file_sha256(built_lib)      # the binary itself, not only the stamp
    assert os.path.getsize(built_lib) > 1 << 20


def test_a_swapped_library_file_is_not_taken_for_the_build(built_lib, tmp_path, monkeypatch):
    """up_to_date() hashes the .so: a library file that is not the one the stamped build wrote fails the check even
    though the source digest in the stamp still matches."""
    import shutil
    from isochrones_amd.csrc import build as hip_build
    assert hip_build.up_to_date()
    fake = tmp_path / "libiso_hip.so"
    shutil.copy(built_lib, fake)
    with open(fake, "ab") as f:
        f.write(b"\0")
    monkeypatch.setattr(hip_build, "OUT", str(fake))
    assert not hip_build.up_to_date()


def test_struct_layout_matches_header(built_lib):
    """ctypes mirror vs the C compiler's layout of iso_prior / iso_model_desc."""
    import subprocess, tempfile
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "isochrones_amd.h"
    int main(void){
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(iso_prior), sizeof(iso_model_desc),
             offsetof(iso_model_desc, mag_val), offsetof(iso_model_desc, has_parallax),
             offsetof(iso_model_desc, prior_mass), offsetof(iso_model_desc, eep_lo),
             offsetof(iso_model_desc, bound_lo), sizeof(iso_tree_desc), sizeof(iso_tree_term),
             offsetof(iso_tree_desc, terms), offsetof(iso_tree_desc, prior_mass), offsetof(iso_tree_desc, bound_lo));
      return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = list(map(int, subprocess.check_output([exe]).split()))
    M = _cabi.IsoModelDesc
    T = _cabi.IsoTreeDesc
    want = [ctypes.sizeof(_cabi.IsoPrior), ctypes.sizeof(M), M.mag_val.offset, M.has_parallax.offset,
            M.prior_mass.offset, M.eep_lo.offset, M.bound_lo.offset, ctypes.sizeof(T),
            ctypes.sizeof(_cabi.IsoTreeTerm), T.terms.offset, T.prior_mass.offset, T.bound_lo.offset]
    assert got == want


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ic = ia.synthetic_track(bands=("V",), fehs=[-1, 0], masses=[0.8, 1.0, 1.2], eeps=np.arange(300., 340.))
    mod = ia.SingleStarModel(ic, V=(10, 0.1))
    with pytest.raises(_cabi.IsoError):
        mod.lnpost([1.0, 320.0, -0.5, 100.0, 0.1])
    with pytest.raises(_cabi.IsoError):
        ic.interp_value([1.0, 320.0, -0.5], ["Teff"])


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setenv("ISOCHRONES_AMD_LIB", "/nonexistent/libiso_hip.so")
    monkeypatch.setattr(_cabi, "_LIB", None)
    with pytest.raises(_cabi.IsoError):
        _cabi.lib()


def test_prior_norms_match_reference_and_closed_forms():
    for case in fx.MODEL_CASES:
        meta = fx.load(case)["meta"]
        mod = fx.make_model(meta)
        assert tuple(mod.bounds("distance")) == tuple(meta["distance_bounds"])
        assert tuple(mod.bounds("AV")) == tuple(meta["AV_bounds"])
        if meta.get("priors") or meta.get("eep_orig_prior"):
            continue                                   # default-prior constants below
        assert np.allclose(mod._priors["mass"].norms, meta["mass_norms"], rtol=1e-13)
        assert np.isclose(mod._priors["feh"]._norm, meta["feh_norm"], rtol=1e-13)
        assert tuple(mod.bounds("distance")) == tuple(meta["distance_bounds"])
        assert tuple(mod.bounds("AV")) == tuple(meta["AV_bounds"])
        assert list(mod.param_names) == meta["param_names"]
        m = mod._priors["mass"]
        assert np.isclose(m.closed_form_total(), m.norms[0], rtol=1e-7)
        f = mod._priors["feh"]
        if np.isfinite(f.bounds[0]):
            assert np.isclose(f.closed_form_norm(), f._norm, rtol=1e-9)


def test_default_mist_norms():
    """with MIST limits the reference finds norms [0.43777194, 8.45012957] and a feh
    norm of 0.99918672."""
    ic = ia.synthetic_track(bands=("V",), fehs=[-4, 0.5], masses=[0.1, 300.0], eeps=np.arange(1., 5.))
    mod = ia.SingleStarModel(ic, V=(10, 0.1))
    assert np.allclose(mod._priors["mass"].norms, [0.43777194, 8.45012957], rtol=1e-8)
    assert np.isclose(mod._priors["feh"]._norm, 0.99918672, rtol=1e-8)
    assert mod.bounds("age") == (5, 10.13) and mod.bounds("eep") == (0, 1710)


def test_priors_integrate_to_one():
    """reference tests/test_priors.py: every prior integrates to 1 over its bounds."""
    from scipy.integrate import quad
    ps = [priors.ChabrierPrior(bounds=(0.1, 100)), priors.FehPrior(bounds=(-4, 0.5)), priors.AgePrior(),
          priors.DistancePrior(3000), priors.AVPrior(), priors.GaussianPrior(1.0, 0.3, bounds=(0, 2)),
          priors.FlatLogPrior((-3, 1)), priors.PowerLawPrior(-2.35, (0.1, 10))]
    for p in ps:
        lo, hi = p.bounds
        pts = [1.0] if isinstance(p, priors.ChabrierPrior) else None
        tot = quad(p.pdf, lo, hi, points=pts, limit=200)[0]
        assert np.isclose(tot, 1.0, rtol=1e-6), (type(p).__name__, tot)
    f = priors.FehPrior(bounds=(-2, 0.5))
    assert f(-2.5) == 0 and f(0.7) == 0 and f(0.0) > 0          # tests/test_priors.py:47-51
    for p in ps:
        x = p.sample(2000, np.random.default_rng(3))
        assert np.all((x >= p.bounds[0]) & (x <= p.bounds[1]))


@pytest.mark.parametrize("name", ["AgePrior", "DistancePrior", "AVPrior", "QPrior", "SalpeterPrior", "FehPrior",
                                  "ChabrierPrior"])
def test_reference_prior_self_checks(name):
    """reference tests/test_priors.py:1-58, test for test: integral == 1 and sample() follows pdf."""
    p = getattr(priors, name)()
    if name == "FehPrior":
        assert p.bounds == (-np.inf, np.inf)
        p.test_sampling(rng=np.random.default_rng(5))
        p.bounds = (-3, 0.25)
        assert p(-3.5) == 0 and p(0.4) == 0
    p.test_integral()
    p.test_sampling(rng=np.random.default_rng(6))


def test_descriptor_packing():
    meta = fx.load("iso_binary_phot6")["meta"]
    mod = fx.make_model(meta)
    d = mod.model_desc()
    assert d.n_stars == 2 and d.n_bands == 6 and d.has_parallax == 1 and d.has_numax == 0
    assert [mod.ic.bc_grid.interp.columns[d.bc_cols[j]] for j in range(6)] == ["J", "H", "K", "BP", "RP", "G"]
    assert np.isnan(d.spec_val[0]) and d.prior_mass.kind == _cabi.PRIOR_CHABRIER
    assert d.prior_distance.hi == 800.0     # 2000 / parallax
    assert mod.param_names == ("eep_0", "eep_1", "age", "feh", "distance", "AV")
    meta = fx.load("track_single_astero")["meta"]
    d = fx.make_model(meta).model_desc()
    assert d.has_numax == 1 and d.has_dnu == 1 and d.dnu_unc == d.dnu_val


def test_synthetic_tables_shapes():
    g, ax, cols = ia.grids.synthetic_track_grid(eeps=np.arange(1.0, 41.0))
    assert g.shape == (15, 196, 40, 18) and len(cols) == 18
    assert ia.grids.mist_masses().size == 196 and ia.grids.mist_log_ages().size == 107
    assert [a.size for a in ia.grids.bc_axes()] == [70, 26, 18, 13]
    g, ax, cols = ia.grids.synthetic_iso_grid(eeps=np.arange(100.0, 300.0))
    assert g.shape == (107, 15, 200, 16) and np.isnan(g).any() and np.isfinite(g).any()


def test_reference_style_entry_points_exist():
    """Names a user of the reference imports for this path (isochrones/__init__.py, mist/__init__.py)."""
    from isochrones_amd.starmodel import EEPPrior
    assert priors.EEP_prior is EEPPrior                      # isochrones/priors.py:394
    from isochrones_amd.mist import MIST_EvolutionTrack, MIST_Isochrone   # noqa: F401
    from isochrones_amd.interp import DFInterpolator                      # noqa: F401
    from isochrones_amd.starmodel import BasicStarModel, StarModel        # noqa: F401
    for name in ("get_ichrone", "StarModel", "SingleStarModel", "BinaryStarModel", "TripleStarModel", "StarCatalog",
                 "ObservationTree", "Observation", "Source", "priors"):
        assert hasattr(ia, name), name
    for name in ("interp_value", "interp_mag", "get_eep", "generate", "isochrone", "model_value", "model_mag", "mass", "radius"):
        assert hasattr(ia.ModelGridInterpolator, name), name
    for name in ("lnpost", "lnlike", "lnprior", "lnpost_polychord", "mnest_prior", "mnest_loglike", "fit_mcmc", "fit_multinest", "evidence",
                 "samples", "derived_samples", "sample_from_prior", "emcee_p0", "set_prior", "set_bounds", "bounds"):
        assert hasattr(ia.BasicStarModel, name), name


def test_table_files_round_trip(tmp_path):
    """ingest.save_table_npz / load_table_npz / interpolator_from_tables: the real-data route that needs no
    HDF5 support on the GPU box (tables exported once where pandas + pytables exist)."""
    from isochrones_amd import ingest
    ic = ia.synthetic_track(bands=("V", "G"), fehs=[-1, 0], masses=[0.8, 1.0, 1.2], eeps=np.arange(300., 340.))
    ingest.save_table_npz(ic.model_grid.interp, tmp_path / "trk.npz")
    ingest.save_table_npz(ic.bc_grid.interp, tmp_path / "bc.npz")
    ic2 = ingest.interpolator_from_tables(tmp_path / "trk.npz", str(tmp_path / "bc.npz"), tracks=True, bands=["G"],
                                          limits=dict(mass=(0.8, 1.2), feh=(-1, 0), age=(5, 10.13)), eep_bounds=(300, 339))
    m1, m2 = ic.model_grid.interp, ic2.model_grid.interp
    assert np.array_equal(m1.grid, m2.grid, equal_nan=True) and list(m1.columns) == list(m2.columns)
    assert all(np.array_equal(a, b) for a, b in zip(m1.index_columns, m2.index_columns))
    assert list(m2.index_names) == list(m1.index_names) and ic2.bands == ["G"] and ic2.eep_bounds == (300, 339)
    assert np.array_equal(ic.bc_grid.interp.grid, ic2.bc_grid.interp.grid)
    d1 = ia.SingleStarModel(ic, G=(10.0, 0.02), Teff=(5700, 100)).model_desc()
    d2 = ia.SingleStarModel(ic2, G=(10.0, 0.02), Teff=(5700, 100)).model_desc()
    assert d1.n_bands == d2.n_bands == 1 and d1.bc_cols[0] == d2.bc_cols[0] == 1


def test_utils_known_answers():
    """isochrones_amd.utils against values computed with the reference's isochrones.utils in the authoring
    container (addmags with and without uncertainties, distance, band_pairs)."""
    from isochrones_amd import utils
    assert utils.addmags(10.0, 11.5) == 9.756693015728262
    assert utils.addmags(10.0, 11.5, 12.25) == 9.652601139068326
    m, u = utils.addmags((10.0, 0.02), (11.0, 0.05))
    assert (m, u) == (9.636148842226767, 0.02004642829733418)
    assert utils.addmags((10.0, 0.02), 11.0)[1] == 0.01426743937691573
    assert utils.distance((0.6, 100), (1.2, 200)) == 1.4318007458582984
    assert utils.fast_addmags([10.0, 11.0]) == 9.636148842226767
    assert utils.band_pairs("JHK") == [("J", "K"), ("H", "K")]


def test_pdf_array_is_pdf_element_by_element():
    """Prior.pdf_array (the vectorised form start-point generation uses) returns exactly what pdf returns per element,
    bounds, NaN and the break of the Chabrier prior included."""
    fam = [priors.FlatPrior((0, 2)), priors.FlatLogPrior((1, 3)), priors.PowerLawPrior(-2.35, (1, 100)),
           priors.GaussianPrior(0.3, 0.2), priors.GaussianPrior(0.3, 0.2, bounds=(0.0, 1.0)), priors.LogNormalPrior(0.1, 0.5),
           priors.ChabrierPrior(bounds=(0.1, 300)), priors.FehPrior(halo_fraction=0.05, bounds=(-4, 0.5)), priors.FehPrior(local=False),
           priors.AgePrior(), priors.DistancePrior(3000), priors.AVPrior((0, 0.5))]
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-5, 320, 400), [0.0, 1.0, 100.0, 0.1, 300.0, 0.5, -4.0, 2.0, 3.0, np.nan, 5.0, 10.15]])
    for p in fam:
        with np.errstate(all="ignore"):
            want = np.array([p.pdf(float(v)) for v in x], dtype=float)
        got = p.pdf_array(x)
        # (numpy's array pow and Python's float pow may differ in the last bit)
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.allclose(got, want, rtol=1e-14, atol=0, equal_nan=True), type(p).__name__
    s = priors.ChabrierPrior(bounds=(0.1, 300)).sample(5000, rng)
    assert s.min() >= 0.1 and s.max() <= 300 and 0.2 < np.median(s) < 0.7


def test_integration_md_stub_mirrors_the_header_structs():
    """The ctypes structures spelled out in INTEGRATION.md section 1 (what a maintainer of the reference would copy) have the
    field names, offsets and sizes of the library's iso_prior / iso_model_desc; the GPU suite runs the stub itself."""
    import re
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.findall(r"```python\n(.*?)```", text[text.index("## 1."):text.index("## 2.")], flags=re.S)[0]
    block = block.replace('_lib = C.CDLL("libiso_hip.so")', "_lib = None").replace("import ctypes as C, numpy as np, torch",
                                                                                   "import ctypes as C, numpy as np")
    stub = types.ModuleType("integration_stub")
    exec(compile(block, "INTEGRATION.md#1", "exec"), stub.__dict__)
    for mine, theirs in ((stub.Prior, _cabi.IsoPrior), (stub.ModelDesc, _cabi.IsoModelDesc)):
        assert ctypes.sizeof(mine) == ctypes.sizeof(theirs)
        assert [n for n, _ in mine._fields_] == [n for n, _ in theirs._fields_]
        assert all(getattr(mine, n).offset == getattr(theirs, n).offset for n, _ in mine._fields_)
