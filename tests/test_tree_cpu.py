"""CPU: the host-side observation-tree builder (isochrones_amd/observation.py) and the oracle's
generic-model restatement against golden data produced by the reference's own
ObservationTree + StarModel (docs/multiple.ipynb configurations + keyword form)."""
import os

import numpy as np
import pytest

import isochrones_amd as ia
from isochrones_amd.observation import Observation, ObservationTree, Source
from oracle import oracle as orc
from tests import _fixtures as fx

TREE_CASES = ["tree_resolved", "tree_resolved_unassoc", "tree_triple1", "tree_triple2", "tree_double_binary",
              "tree_kwargs_single", "tree_kwargs_binary", "tree_kwargs_triple", "tree_out_of_order"]
# star.ini fixtures (tests/golden/ini/, layouts of the reference's tests/star1..star4) with the N / index
# variants of the reference's tests/test_ini.py
INI_CASES = ["ini_single", "ini_binary", "ini_binary_unassoc", "ini_triple", "ini_triple_unassoc1",
             "ini_triple_unassoc2", "ini_triple_b", "ini_triple_b_unassoc2", "ini_flat", "ini_flat_N2"]
INI_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ini")


def build_notebook_tree(name):
    """docs/multiple.ipynb cell 21: three unresolved 2MASS bands + a resolved relative AO K image."""
    obs = ObservationTree(name=name)
    for band, m in zip("JHK", (12.11, 11.74, 11.68)):
        o = Observation("2MASS", band, 4)
        o.add_source(Source(m, 0.02))
        obs.add_observation(o)
    o = Observation("AO", "K", 0.1)
    o.add_source(Source(0.0, 0.02, separation=0, pa=0, relative=True, is_reference=True))
    o.add_source(Source(2.43, 0.02, separation=0.2, pa=100, relative=True, is_reference=False))
    obs.add_observation(o)
    return obs


def make_tree_model(meta):
    iso_meta = dict(kind="iso", limits=meta["limits"], eep_bounds=meta["eep_bounds"])
    ic = fx.make_ic(iso_meta)
    if meta.get("ini"):
        return ic, ia.TreeStarModel.from_ini(ic, folder=os.path.join(INI_DIR, meta["ini"]), **meta["from_ini_kwargs"])
    kw = {k: (tuple(v) if isinstance(v, list) and k not in ("N", "index") else v) for k, v in meta["kwargs"].items()}
    obs = build_notebook_tree("t") if meta["built"] else None
    return ic, ia.TreeStarModel(ic, obs=obs, **kw)


@pytest.mark.parametrize("case", TREE_CASES + INI_CASES)
def test_tree_structure_and_oracle_vs_reference(case):
    g = fx.load(case)
    ic, mod = make_tree_model(g["meta"])
    _check_tree_case(g, ic, mod)


def _check_tree_case(g, ic, mod):
    meta = g["meta"]
    # --- structure: same parameters, leaves, and per-node blending as the reference's tree ---
    assert list(mod.param_names) == meta["param_names"]
    assert mod.obs.leaf_labels == meta["leaf_labels"]
    assert mod.obs.systems == meta["systems"] and {str(k): v for k, v in mod.obs.Nstars.items()} == meta["Nstars"]
    mine = []
    for n in mod.obs.obs_nodes():
        mine.append(dict(band=n.observation.band, relative=n.source.relative, mag=n.source.mag, unc=n.source.e_mag,
                         leaves=[l.label for l in n.leaves()],
                         ref_leaves=[l.label for l in n.reference.leaves()] if n.reference is not None else None))
    assert mine == meta["nodes"]
    # --- numbers: oracle on the flattened tree vs the reference's StarModel ---
    desc = mod.tree_desc()
    oic = fx.make_oracle_ic(ic)
    post, prior, like = orc.tree_lnpost(oic, desc, g["pars"].T.copy())
    fx.assert_close(prior, g["lnprior"], 1e-11, atol=1e-12, what="lnprior")
    fx.assert_close(like, g["lnlike"], 1e-11, atol=1e-11, what="lnlike")
    fx.assert_close(post, g["lnpost"], 1e-11, atol=1e-11, what="lnpost")
    fx.assert_close(mod.prior_transform(g["cube_in"]), g["cube_out"], 1e-15, what="prior_transform")
    # mnest_prior (reference starmodel.py:644-656): the box transform, then every system's EEPs in descending
    # order - 32 random (non-degenerate) cubes through the reference's own method
    rows = g["mnest_in"].copy()
    for r in rows:
        mod.mnest_prior(r, len(r), len(r))
    assert np.array_equal(rows, g["mnest_out"])
    assert np.array_equal(mod.mnest_transform(g["mnest_in"]), g["mnest_out"])
    if max(meta["Nstars"].values()) > 1:                       # the sort matters for these cases
        assert not np.array_equal(mod.prior_transform(g["mnest_in"]), g["mnest_out"])
    assert np.isfinite(g["lnpost"]).sum() > 50


def test_keyword_tree_equals_basic_model_on_the_oracle():
    """reference tests/test_likelihood.py: StarModel (tree) == BasicStarModel for unresolved systems
    once the priors are identical."""
    g = fx.load("tree_kwargs_binary")
    meta = g["meta"]
    ic, tree = make_tree_model(meta)
    basic = ia.BinaryStarModel(ic, J=(13.3, 0.05), K=(12.9, 0.05), parallax=(2.0, 0.1))
    basic.set_bounds(distance=(0, 10000))                      # the generic model keeps the default distance bound
    basic.set_bounds(mass=(0.1, 100.0))                        # ... and the default Chabrier bounds
    oic = fx.make_oracle_ic(ic)
    pars = g["pars"]
    ok = np.all(pars[:, 4:6] > 0, axis=1)
    a = orc.tree_lnpost(oic, tree.tree_desc(), pars[ok].T.copy())
    b = oic.lnpost(basic.model_desc(), pars[ok].T.copy())
    fin = np.isfinite(b[0])
    assert fin.sum() > 50
    assert np.allclose(a[0][fin], b[0][fin], rtol=1e-10, atol=1e-9)
    assert np.allclose(a[1][fin], b[1][fin], rtol=1e-10, atol=1e-9)


def test_tree_from_df_and_errors():
    import pandas as pd
    df = pd.DataFrame(dict(name=["2MASS", "2MASS", "AO", "AO"], band=["J", "K", "K", "K"], resolution=[4.0, 4.0, 0.1, 0.1],
                           mag=[12.1, 11.7, 0.0, 2.4], e_mag=[0.02] * 4, separation=[0, 0, 0, 0.2], pa=[0, 0, 0, 100],
                           relative=[False, False, True, True]))
    t = ObservationTree.from_df(df, name="x")
    assert [str(o) for o in t.observations] == ["2MASS-J", "2MASS-K", "AO-K"]
    t.define_models(None, N=1, index=[0, 1])
    assert t.leaf_labels == ["0_0", "1_0"] and t.systems == [0, 1]
    assert t.param_description == ["eep_0_0", "age_0", "feh_0", "distance_0", "AV_0",
                                   "eep_1_0", "age_1", "feh_1", "distance_1", "AV_1"]
    with pytest.raises(ValueError):
        t.add_spectroscopy(label="3_0", Teff=(5000, 100))
    with pytest.raises(ValueError):
        t.add_parallax((1.0, 0.1), system=5)
    t.add_limit(logg=(3.5, None))
    prog = t.program(["J", "K"])
    assert prog["limits"][0]["hi"] == np.inf and len(prog["terms"]) == 3          # the AO reference source adds no term
    assert [x["mask"] for x in prog["terms"]] == [2, 3, 3] and prog["terms"][0]["ref_mask"] == 1


def _isotrack_objects():
    from isochrones_amd.interp import DFInterpolator
    from isochrones_amd.models import BolometricCorrectionGrid, IsochroneGrid, IsochroneInterpolator
    g = fx.load("isotrack")
    meta = g["meta"]
    track = fx.make_ic(dict(kind="track", limits=meta["limits_track"], eep_bounds=meta["eep_bounds"]))
    (_, _, _), (bg, bax, bands) = fx.table_parts("iso")
    iso = IsochroneInterpolator(
        IsochroneGrid(DFInterpolator.from_arrays(g["iso_grid"], [g["iso_ax0"], g["iso_ax1"], g["iso_ax2"]], meta["iso_columns"]),
                      limits={k: tuple(v) for k, v in meta["limits_iso"].items()}),
        BolometricCorrectionGrid(DFInterpolator.from_arrays(bg, bax, bands), bands=bands), bands=bands,
        eep_bounds=meta["eep_bounds"])
    obs = {k: tuple(v) for k, v in meta["obs"].items()}
    return g, iso, track, obs


def test_isotrack_oracle_composition_vs_reference():
    """IsoTrackModel (starmodel.py:2010-2104) = track prior + age prior; iso likelihood + track
    likelihood with the parallax term once — composed here from the oracle's two evaluations."""
    import math
    g, iso, track, obs = _isotrack_objects()
    mod = ia.IsoTrackModel(iso, track, **obs)
    p = g["pars"]
    iso_p = np.column_stack([p[:, 0], p[:, 2], p[:, 3], p[:, 4], p[:, 5]])
    trk_p = np.column_stack([p[:, 1], p[:, 0], p[:, 3], p[:, 4], p[:, 5]])
    t = fx.make_oracle_ic(track).lnpost(mod._track_model.model_desc(), trk_p.T.copy())
    i = fx.make_oracle_ic(iso).lnpost(mod._iso_model.model_desc(), iso_p.T.copy())
    lo, hi = mod._priors["age"].bounds
    age = p[:, 2]
    with np.errstate(all="ignore"):
        ln_age = np.where((age < lo) | (age > hi), -np.inf, math.log(math.log(10) / (10 ** hi - 10 ** lo)) + age * math.log(10))
        prior = t[1] + ln_age
        like = i[2] + t[2]
        post = np.where(np.isfinite(prior), prior + like, -np.inf)
    fx.assert_close(prior, g["lnprior"], 1e-11, atol=1e-12, what="lnprior")
    fx.assert_close(like, g["lnlike"], 1e-11, atol=1e-11, what="lnlike")
    fx.assert_close(post, g["lnpost"], 1e-11, atol=1e-11, what="lnpost")


def test_tree_to_df_round_trip_and_print_ascii(capsys):
    """to_df / from_df round trip (reference observation.py:796-835) and the text rendering of the hierarchy."""
    tree = build_notebook_tree("t")
    df = tree.to_df()
    assert list(df.columns) == ["name", "band", "resolution", "mag", "e_mag", "separation", "pa", "relative"]
    assert len(df) == sum(len(o.sources) for o in tree.observations)
    again = ia.ObservationTree.from_df(df, name="t")
    assert again.to_df().sort_values(list(df.columns)).reset_index(drop=True).equals(
        df.sort_values(list(df.columns)).reset_index(drop=True))
    tree.print_ascii()
    out = capsys.readouterr().out
    assert out.count("\n") >= len(df) and ("|-- " in out or "+-- " in out)


def test_ini_reader_and_checks_of_reference_test_ini(tmp_path):
    """reference tests/test_ini.py: n_params / systems / Nstars of every fixture + variant, and the
    reader's handling of comments, quoting, lists and malformed lines."""
    from isochrones_amd import ini
    expect = {"ini_single": (5, [0], {0: 1}), "ini_binary": (6, [0], {0: 2}), "ini_binary_unassoc": (10, [0, 1], {0: 1, 1: 1}),
              "ini_triple": (7, [0], {0: 3}), "ini_triple_unassoc1": (11, [0, 1], {0: 2, 1: 1}),
              "ini_triple_unassoc2": (11, [0, 1], {0: 1, 1: 2}), "ini_triple_b": (7, [0], {0: 3}),
              "ini_triple_b_unassoc2": (11, [0, 1], {0: 1, 1: 2})}
    for case, (npar, systems, nstars) in expect.items():
        meta = fx.load(case)["meta"]
        ic, mod = make_tree_model(meta)
        assert len(mod.param_names) == npar and mod.n_params == npar
        assert mod.obs.systems == systems and mod.obs.Nstars == nstars
        assert mod.name == meta["ini"]
    assert ia.TreeStarModel.get_bands(os.path.join(INI_DIR, "binary", "star.ini")) == ["G", "BP", "RP", "J", "H", "K"]
    _, tri = make_tree_model(fx.load("ini_triple")["meta"])
    assert tri.bounds("AV") == (0, 0.9)
    # the reader
    f = tmp_path / "star.ini"
    f.write_text("# c\nname = 'a # b'  # trailing\nTeff = 5000, 100,\nN = 2\n\n[ s1 ]\nK = 1, 0.1\nlist = a, 'b c'\n")
    scalars, sections = ini.read_ini(str(f))
    assert scalars == {"name": "a # b", "Teff": ["5000", "100"], "N": "2"}
    assert sections == {"s1": {"K": ["1", "0.1"], "list": ["a", "b c"]}}
    assert ini.parse_value(scalars["Teff"]) == [5000.0, 100.0] and ini.parse_value("x") == "x"
    f.write_text("Teff 5000\n")
    with pytest.raises(ValueError, match="expected 'key = value'"):
        ini.read_ini(str(f))
    f.write_text("[s]\nK = 12\n")
    with pytest.raises(ValueError, match="magnitude, uncertainty"):
        ini.observation_rows(ini.read_ini(str(f))[1])
    # pardict round trip
    p = list(np.arange(11.0))
    _, m = make_tree_model(fx.load("ini_triple_unassoc1")["meta"])
    assert m.obs.pardict2p(m.obs.p2pardict(p)) == p


def test_saved_models_describe_the_same_device_program(tmp_path):
    """persist.py on the host side only: an (unfitted) basic model and the ini-built tree models come back with
    the same measurements, priors, bounds and — what the kernels actually see — the same flattened descriptors."""
    import ctypes as C
    from isochrones_amd import priors

    def raw(desc):
        return bytes(C.string_at(C.addressof(desc), C.sizeof(desc)))

    meta = fx.load("ini_flat")["meta"]
    ic = fx.make_ic(dict(kind="iso", limits=meta["limits"], eep_bounds=meta["eep_bounds"]))
    b = ia.BinaryStarModel.from_ini(ic, folder=os.path.join(INI_DIR, "flat"))
    b.set_prior(AV=priors.GaussianPrior(0.2, 0.1, bounds=(0, 1)), distance=priors.PowerLawPrior(2.0, (1.0, 800.0)))
    b.set_bounds(eep=(210, 600))
    b._priors["eep"].orig_prior = priors.LogNormalPrior(0.0, 0.5)
    f = str(tmp_path / "b.npz")
    b.save(f)
    back = ia.StarModel.load_hdf(f, ic=ic)
    assert type(back) is ia.BinaryStarModel and back.kwargs == b.kwargs and back._bounds == b._bounds
    assert type(back._priors["eep"].orig_prior) is priors.LogNormalPrior
    assert raw(back.model_desc()) == raw(b.model_desc())
    for case in ("ini_triple_unassoc1", "ini_binary", "ini_single"):
        _, t = make_tree_model(fx.load(case)["meta"])
        t.set_prior(feh=priors.FlatPrior((-0.5, 0.3)))
        g = str(tmp_path / (case + ".npz"))
        t.save(g)
        tb = ia.TreeStarModel.load(g, ic=t.ic)
        assert tb.param_names == t.param_names and tb.obs.leaf_labels == t.obs.leaf_labels
        assert tb._bounds == t._bounds and tb.obs.spectroscopy == t.obs.spectroscopy
        d0, d1 = t.tree_desc(), tb.tree_desc()
        pars = fx.load(case)["pars"].T.copy()
        oic = fx.make_oracle_ic(t.ic)
        a, b_ = orc.tree_lnpost(oic, d0, pars)[0], orc.tree_lnpost(oic, d1, pars)[0]
        assert np.array_equal(a, b_, equal_nan=True) and np.isfinite(a).sum() > 20
    with pytest.raises(IOError):
        ia.StarModel.load_hdf(str(tmp_path / "missing.npz"))
    with pytest.raises(IOError):
        t.save_hdf(g)                                  # exists, no overwrite / append


def test_define_models_on_leaves_selected_by_pattern():
    """reference ObservationTree.define_models(leaves=<pattern>) / select_leaves (observation.py:234-249, 1020-1023)."""
    a, b, c = build_notebook_tree("a"), build_notebook_tree("b"), build_notebook_tree("c")
    a.define_models(N=[1, 2], index=[0, 1])
    b.define_models(leaves="AO", N=[1, 2], index=[0, 1])                 # the resolved image's two sources
    c.define_models(leaves=r"K=\(.*\[0\.10\]", N=[1, 2], index=[0, 1])    # the reference's spelling of a node label
    assert a.leaf_labels == b.leaf_labels == c.leaf_labels == ["0_0", "1_0", "1_1"]
    assert [n.label for n in a.root.select_leaves("2MASS")] == [n.label for n in a.root.select_leaves("AO")]
    assert a.root.select_leaves("nothing") == []
    with pytest.raises(ValueError, match="no observation node"):
        a.define_models(leaves="nothing")
    one = build_notebook_tree("d").root.select_leaves(r"@\(0\.20, 100")
    assert len(one) == 1 and one[0].source.separation == 0.2
    assert one[0].reference_label == "AO K=(2.43, 0.02) @(0.20, 100 [0.10])"


def test_reference_spellings_of_tree_queries_and_prior_names():
    from isochrones_amd import priors
    t = build_notebook_tree("x")
    t.define_models(N=1, index=[0, 1])
    assert t.N_model_nodes == 2 and len(t.get_obs_leaves()) == 2 and t.get_leaf("1_0").label == "1_0"
    assert len(t.select_observations("AO-K")) == 2 and len(t.select_observations("2MASS-J")) == 1
    assert len(t.get_obs_nodes()) == 5 and [n.label for n in t.get_model_nodes()] == ["0_0", "1_0"]
    assert [n.label for n in t.select_leaves("AO")] == [n.label for n in t.obs_leaf_nodes] and t.trim() is None
    t.clear_models()
    assert t.N_model_nodes == 0 and t.get_leaf("0_0") is None
    assert priors.BoundedPrior is priors.Prior and priors.EEP_prior.__name__ == "EEPPrior"
    with pytest.raises(AttributeError):
        priors.NoSuchPrior


def test_from_ini_with_an_obsfile(tmp_path):
    """star.ini with ``obsfile = obs.csv`` (reference starmodel.py:262-275, 427-428): the photometry table
    comes from the csv, spectroscopy / N / index from the ini."""
    meta = fx.load("tree_resolved_unassoc")["meta"]
    ic, direct = make_tree_model(meta)
    folder = tmp_path / "KOI-7"
    folder.mkdir()
    build_notebook_tree("x").to_df().to_csv(folder / "obs.csv", index=False)
    kw = meta["kwargs"]
    lines = ["obsfile = obs.csv", "index = 0, 1"]
    lines += ["%s = %r, %r" % (k, v[0], v[1]) for k, v in kw.items() if k not in ("N", "index")]
    (folder / "star.ini").write_text("\n".join(lines) + "\n")
    mod = ia.StarModel.from_ini(ic, folder=str(folder))
    assert mod.name == "KOI-7" and mod.param_names == direct.param_names and mod.obs.leaf_labels == direct.obs.leaf_labels
    g = fx.load("tree_resolved_unassoc")
    oic = fx.make_oracle_ic(ic)
    a = orc.tree_lnpost(oic, mod.tree_desc(), g["pars"].T.copy())[0]
    fx.assert_close(a, g["lnpost"], 1e-11, atol=1e-11, what="lnpost of the obsfile model vs the reference golden")


def test_saved_model_priors_are_plain_data(tmp_path):
    """The priors entry of a saved model is JSON (class name + constructor parameters): every prior family round-trips,
    nothing is unpickled, an older pickled entry or a record naming anything but a prior class is refused."""
    import json
    import pickle
    from isochrones_amd import persist, priors as P
    meta = fx.load("ini_flat")["meta"]
    ic = fx.make_ic(dict(kind="iso", limits=meta["limits"], eep_bounds=meta["eep_bounds"]))
    mod = ia.BasicStarModel.from_ini(ic, folder=os.path.join(INI_DIR, "flat"))
    f = str(tmp_path / "m.npz")
    mod.save(f)
    with np.load(f, allow_pickle=False) as z:
        arrays = {k: z[k] for k in z.files}
    assert arrays["priors"].dtype.kind == "U" and "cls" in str(arrays["priors"])
    assert ia.BasicStarModel.load(f, ic=ic).kwargs == mod.kwargs
    # every family, including bounds set after construction (GaussianPrior: bounds truncate, reference priors.py:131-140)
    g = P.GaussianPrior(0.0, 0.1)
    g.bounds = (-1.0, 0.5)
    assert g.bounded == 1 and g.lnpdf(2.0) == -np.inf and g.pdf(2.0) == 0.0 and g.desc().bounded == 1
    assert g.norm == 1.0 and np.isclose(P.GaussianPrior(0.0, 0.1, bounds=(0.0, 5.0)).norm, 0.5)
    with pytest.raises(ValueError, match="integral test failed"):           # reference bounds setter, priors.py:123-129
        P.GaussianPrior(0.0, 0.1).bounds = (0.0, 5.0)
    fam = [P.FlatPrior((0, 2)), P.FlatLogPrior((1, 3)), P.PowerLawPrior(-2.35, (1, 100)), g, P.GaussianPrior(1.0, 2.0),
           P.LogNormalPrior(0.1, 0.5), P.ChabrierPrior(bounds=(0.1, 300)), P.FehPrior(halo_fraction=0.05, bounds=(-4, 0.5)),
           P.FehPrior(local=False), P.AgePrior(), P.DistancePrior(3000), P.AVPrior((0, 0.5)), P.QPrior(), P.SalpeterPrior()]
    for p in fam:
        spec = json.loads(json.dumps(P.prior_to_spec(p)))
        q = P.prior_from_spec(spec)
        assert type(q) is type(p) and q.bounds == p.bounds and q.bounded == p.bounded
        a, b = p.desc(), q.desc()
        assert all(getattr(a, k) == getattr(b, k) or (np.isnan(getattr(a, k)) and np.isnan(getattr(b, k)))
                   for k, _ in a._fields_), type(p).__name__
    # refused: the pickled form of older containers (whatever it holds), unknown classes, dotted names
    arrays["priors"] = np.frombuffer(pickle.dumps({"mass": os.getcwd, "x": os.system}, protocol=4), dtype=np.uint8)
    np.savez(str(tmp_path / "evil.npz"), **arrays)
    with pytest.raises(ValueError, match="pickled form"):
        ia.BasicStarModel.load(str(tmp_path / "evil.npz"), ic=ic)
    for bad in ({"cls": "np.ctypeslib.os.system"}, {"cls": "Prior"}, {"cls": "EEP_prior"}, "FlatPrior", {"bounds": [0, 1]}):
        arrays["priors"] = np.array(json.dumps({"mass": bad}))
        np.savez(str(tmp_path / "evil2.npz"), **arrays)
        with pytest.raises(ValueError, match="not a prior record"):
            ia.BasicStarModel.load(str(tmp_path / "evil2.npz"), ic=ic)
    assert not hasattr(persist, "pickle")


def _random_tree_spec(rng):
    """A random set of observations the way a user would enter them: 1-3 unresolved catalogue bands, optionally a
    seeing-limited image that splits a wide companion, optionally an AO image that also splits a close one."""
    wide, close = bool(rng.random() < 0.6), bool(rng.random() < 0.6)
    if not (wide or close) and rng.random() < 0.5:
        close = True
    spec = []
    for band in rng.choice(["J", "H", "K", "G", "V"], int(rng.integers(1, 4)), replace=False):
        spec.append(("2MASS", str(band), 4.0, [(float(rng.uniform(9.5, 12.5)), 0.02, 0.0, 0.0, False, False)]))
    pos = [(0.0, 0.0)]
    if close:
        pos.append((float(rng.uniform(0.2, 0.45)), float(rng.uniform(0, 360))))
    if wide:
        pos.append((float(rng.uniform(1.8, 3.0)), float(rng.uniform(0, 360))))
    if wide and rng.random() < 0.7:                      # seeing-limited: the close pair (if any) stays blended
        rel = bool(rng.random() < 0.5)
        srcs = [(0.0 if rel else float(rng.uniform(10, 12)), 0.02, 0.0, 0.0, rel, rel),
                (float(rng.uniform(0.5, 3.0)) + (0.0 if rel else 11.0), 0.03, pos[-1][0], pos[-1][1], rel, False)]
        spec.append(("seeing", str(rng.choice(["K", "G"])), 1.0, srcs))
    if len(pos) > 1 and (close or rng.random() < 0.5):   # AO: every component on its own
        rel = bool(rng.random() < 0.6)
        srcs = []
        for k, (sep, pa) in enumerate(pos):
            dm = 0.0 if k == 0 else float(rng.uniform(0.5, 3.5))
            srcs.append(((dm if rel else 10.5 + dm), 0.02, sep, pa, rel, rel and k == 0))
        spec.append(("AO", str(rng.choice(["K", "H"])), 0.1, srcs))
    return spec


def _build_tree(api, spec, name):
    t = api.ObservationTree(name=name)
    for inst, band, res, srcs in spec:
        o = api.Observation(inst, band, res)
        for mag, e, sep, pa, rel, isref in srcs:
            o.add_source(api.Source(mag, e, separation=sep, pa=pa, relative=rel, is_reference=isref))
        t.add_observation(o)
    return t


@pytest.mark.parametrize("seed", [201, 202, 203, 204, 205, 206, 3041, 3042])
def test_random_trees_against_the_reference_itself(seed, tmp_path, monkeypatch):
    """Container-only (needs /root/reference): random observation sets (unresolved bands, a seeing-limited image, an AO
    image; relative or absolute photometry) and random N / index assignments are handed to the reference's
    ObservationTree + StarModel and to this build's; structure, flattened terms and the oracle's numbers must agree as
    for the committed cases."""
    if not os.path.isdir("/root/reference/isochrones"):
        pytest.skip("the reference tree is only present in the build container")
    import isochrones_amd.observation as mine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(os.path.join(root, "oracle"))
    import make_golden as mg
    import ref_harness as rh
    fx.tables()
    monkeypatch.setattr(mg, "OUT", str(tmp_path))
    monkeypatch.setattr(fx, "GOLDEN", str(tmp_path))
    sm, obs_mod = rh.ref("starmodel"), rh.ref("observation")
    rng = np.random.default_rng(seed)
    iso, bc = mg.small_iso(), mg.small_bc()
    axes = iso[1]
    limits = mg.limits_of("iso", axes)
    done = 0
    for k in range(4):
        spec = _random_tree_spec(rng)
        n_fine = max(len(srcs) for _, _, _, srcs in spec)
        N = [int(rng.choice([1, 1, 2])) for _ in range(n_fine)]
        if rng.random() < 0.5:
            index = [0] * n_fine
        else:
            index = [int(v) for v in rng.permutation(n_fine)] if rng.random() < 0.6 else [int(min(j, 1)) for j in range(n_fine)]
        if sum(N[j] for j in range(n_fine) if index[j] == index[0]) > 3:
            N = [1] * n_fine
        kw = dict(N=N if n_fine > 1 else N[0], index=index if n_fine > 1 else index[0])
        if rng.random() < 0.5: kw["parallax"] = (float(rng.choice([2.0, 5.0])), 0.05)
        if rng.random() < 0.4: kw["Teff"] = (float(rng.uniform(5000, 6500)), 100)
        if rng.random() < 0.3: kw["logg"] = (float(rng.uniform(4.0, 4.6)), 0.15)
        if rng.random() < 0.3: kw["AV"] = (0.2, 0.1)
        name = "fresh_tree_%d_%d" % (seed, k)
        ic_ref = rh.make_ref_ic("iso", iso, bc, limits, (axes[2][0], axes[2][-1]))
        try:
            ref_mod = sm.StarModel(ic_ref, obs=_build_tree(obs_mod, spec, name), **kw)
        except Exception:                # a layout the reference itself does not take
            with pytest.raises(Exception):
                ic0 = fx.make_ic(dict(kind="iso", limits={k_: list(map(float, v)) for k_, v in limits.items()},
                                      eep_bounds=[float(axes[2][0]), float(axes[2][-1])]))
                ia.TreeStarModel(ic0, obs=_build_tree(mine, spec, name), **kw)
            continue
        mg._emit_tree_case(name, ref_mod, kw, True, rng, axes, limits, obs_mod)
        g = fx.load(name)
        meta = g["meta"]
        ic = fx.make_ic(dict(kind="iso", limits=meta["limits"], eep_bounds=meta["eep_bounds"]))
        kw2 = {k_: (tuple(v) if isinstance(v, list) and k_ not in ("N", "index") else v) for k_, v in meta["kwargs"].items()}
        mod = ia.TreeStarModel(ic, obs=_build_tree(mine, spec, name), **kw2)
        if np.isfinite(g["lnpost"]).sum() <= 50:          # the draw left too few finite points for the last check
            g["lnpost"] = g["lnpost"].copy()
        _check_tree_case(g, ic, mod)
        done += 1
    assert done >= 2
