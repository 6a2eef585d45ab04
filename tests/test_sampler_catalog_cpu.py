"""CPU tests of the host logic around the hot path: the stretch-move samplers (on CPU tensors with
toy log-densities), the batch_starfit sharding rule, and the N>1 catalog path under a real
world_size-2 `gloo` process group (fit function injected so no GPU is needed)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

import isochrones_amd as ia
from isochrones_amd.catalog import BatchedEnsembleSampler, result_columns
from isochrones_amd.sampler import EnsembleSampler, summarize_chain

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rule_matches_batch_starfit():
    """scripts/batch_starfit:60-62: `awk "NR % NPROCS == i"` with 1-based NR."""
    for world in (1, 2, 3, 8):
        owners = [ia.shard_of(i, world) for i in range(50)]
        assert owners == [(i + 1) % world for i in range(50)]
        got = np.sort(np.concatenate([ia.shard_indices(50, r, world) for r in range(world)]))
        assert np.array_equal(got, np.arange(50))          # a partition: every star exactly once
        for r in range(world):
            assert all(ia.shard_of(int(i), world) == r for i in ia.shard_indices(50, r, world))


def test_ensemble_sampler_recovers_gaussian():
    mu = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)
    sig = torch.tensor([0.5, 2.0, 0.1], dtype=torch.float64)
    lnp = lambda x: -0.5 * (((x - mu) / sig) ** 2).sum(dim=1)
    s = EnsembleSampler(40, 3, lnp, seed=3, device="cpu")
    p0 = mu + 0.1 * torch.randn(40, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    pos, prob = s.run_mcmc(p0, 300, store=False)
    s.reset()
    s.run_mcmc(pos, 600, lnprob0=prob)
    assert s.chain.shape == (40, 600, 3) and s.lnprobability.shape == (40, 600)
    flat = s.flatchain.numpy()
    assert np.allclose(flat.mean(axis=0), mu.numpy(), atol=0.15 * sig.numpy().max())
    assert np.allclose(flat.std(axis=0), sig.numpy(), rtol=0.15)
    acc = s.acceptance_fraction.numpy()
    assert 0.2 < acc.mean() < 0.9
    row = summarize_chain(s.flatchain, s.flatlnprobability)
    assert row.shape == (10,) and abs(row[0] - 1.0) < 0.1


def test_sampler_rejects_bad_start_and_nonfinite_proposals():
    lnp = lambda x: torch.where(x[:, 0] > 0, -0.5 * (x ** 2).sum(dim=1), torch.full((x.shape[0],), -float("inf"),
                                                                                dtype=torch.float64))
    s = EnsembleSampler(8, 2, lnp, seed=0, device="cpu")
    with pytest.raises(ValueError):
        s.run_mcmc(-torch.ones(8, 2, dtype=torch.float64), 1)
    pos, prob = s.run_mcmc(torch.rand(8, 2, dtype=torch.float64) + 0.1, 200)
    assert bool((s.flatchain[:, 0] > 0).all())          # -inf proposals are never accepted


def test_batched_sampler_independent_stars():
    """S stars with different means advance in lock-step and do not mix."""
    S, W, D = 5, 16, 2
    centres = torch.arange(S, dtype=torch.float64)[:, None] * torch.tensor([10.0, -5.0], dtype=torch.float64)
    def lnp(x, sid):
        return -0.5 * ((x - centres[sid.long()]) ** 2).sum(dim=1)
    s = BatchedEnsembleSampler(S, W, D, lnp, seed=5, device="cpu")
    pos = centres[:, None, :] + 0.1 * torch.randn(S, W, D, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    l0 = s.lnpost_all(pos)
    s.run(pos, l0, 200)
    chain, lnps = s.run(pos, l0, 300, keep=True)
    assert chain.shape == (S, W, 300, D) and lnps.shape == (S, W, 300)
    means = chain.reshape(S, -1, D).mean(dim=1)
    assert torch.allclose(means, centres, atol=0.2)
    stds = chain.reshape(S, -1, D).std(dim=1)
    assert torch.allclose(stds, torch.ones_like(stds), rtol=0.2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_worker(rank, world, port, n_stars, out_dir):
    sys.path.insert(0, ROOT)
    import pandas as pd
    import torch.distributed as dist
    import isochrones_amd as ia_
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    df = pd.DataFrame({"V_mag": np.linspace(8, 12, n_stars), "V_mag_unc": 0.02}, index=["s%03d" % i for i in range(n_stars)])
    cat = ia_.StarCatalog(df, bands=["V"])
    seen = []

    def fake_fit(catalog, ic, indices, N=1, **kw):
        seen.extend(int(i) for i in indices)
        rows = np.zeros((len(indices), 3 * (N + 4) + 3))
        rows[:, 0] = np.asarray(indices) * 10.0 + 1.0          # a value that identifies the star
        rows[:, -3] = rank                                      # who fitted it
        rows[:, -1] = 1.0
        return rows

    res = ia_.fit_catalog(cat, ic=None_IC(), N=1, fit_fn=fake_fit)
    assert seen == [int(i) for i in ia_.shard_indices(n_stars, rank, world)]
    res.to_pickle(os.path.join(out_dir, "res%d.pkl" % rank))
    dist.barrier()
    dist.destroy_process_group()


class None_IC:
    param_names = ("mass", "eep", "feh", "distance", "AV")


@pytest.mark.parametrize("n_stars", [7, 16])
def test_fit_catalog_world2_gloo(tmp_path, n_stars):
    import pandas as pd
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gloo_worker, args=(2, port, n_stars, str(tmp_path)), nprocs=2, join=True)
    r0 = pd.read_pickle(tmp_path / "res0.pkl")
    r1 = pd.read_pickle(tmp_path / "res1.pkl")
    assert r0.equals(r1)                                        # all-gather: same table on every rank
    assert list(r0.columns) == result_columns(None_IC.param_names)
    assert np.array_equal(r0.iloc[:, 0].values, np.arange(n_stars) * 10.0 + 1.0)     # rows landed in order
    assert np.array_equal(r0["lnpost_max"].values, (np.arange(n_stars) + 1) % 2)     # fitted by rank (i+1)%2
    assert list(r0.index) == ["s%03d" % i for i in range(n_stars)]


def test_star_catalog_schema():
    import pandas as pd
    df = pd.DataFrame({"J_mag": [9.0, 10.0], "J_mag_unc": [0.02, 0.03], "K_mag": [8.5, 9.4], "K_mag_unc": [0.02, 0.02],
                       "parallax": [5.0, 2.0], "parallax_unc": [0.1, 0.1]})
    cat = ia.StarCatalog(df, props=["parallax"])
    assert cat.bands == ("J", "K") and len(cat) == 2
    with pytest.raises(ValueError):
        ia.StarCatalog(df.drop(columns=["K_mag_unc"]))
    ic = ia.synthetic_isochrone(bands=("J", "K"), ages=[9.0, 9.5, 10.0], fehs=[-0.5, 0.0, 0.5], eeps=np.arange(300., 340.))
    v, u = cat.get_measurement("parallax")
    assert list(v) == [5.0, 2.0] and list(u) == [0.1, 0.1]
    assert [b for b, _ in cat.iter_bands()] == ["J", "K"] and [p for p, _ in cat.iter_props()] == ["parallax"]
    mods = list(cat.iter_models(ic, N=2))
    assert len(mods) == 2 and mods[0].N == 2 and mods[1].bands == ["J", "K"]
    assert mods[1].kwargs["parallax"] == (2.0, 0.1) and mods[1].bounds("distance") == (0, 1000.0)


def test_vectorised_catalog_descriptors_equal_per_model_descriptors():
    """CatalogPosterior.build_descs fills iso_model_desc records column-wise; they must be byte-
    identical to the records of one BasicStarModel per row (incl. missing parallax / Teff)."""
    import ctypes as C
    import pandas as pd
    from isochrones_amd import _cabi
    from isochrones_amd.catalog import CatalogPosterior
    rng = np.random.default_rng(2)
    n = 40
    df = pd.DataFrame({"G_mag": 10 + rng.random(n), "G_mag_unc": 0.01 + 0.01 * rng.random(n),
                       "RP_mag": 9 + rng.random(n), "RP_mag_unc": 0.02,
                       "parallax": 1 + 5 * rng.random(n), "parallax_unc": 0.05,
                       "Teff": 5000 + 1000 * rng.random(n), "Teff_unc": 80.0})
    df.loc[3, "parallax"] = np.nan
    df.loc[5, "parallax"] = -0.2
    df.loc[7, "Teff"] = np.nan
    cat = ia.StarCatalog(df, bands=["G", "RP"], props=["parallax", "Teff"])
    ic = ia.synthetic_track(bands=("G", "RP"), fehs=[-1, 0, .5], masses=[.5, 1, 2], eeps=np.arange(1., 50.))
    arr, template = CatalogPosterior.build_descs(cat, ic)
    assert arr.shape == (n,) and template.bands == ["G", "RP"]
    for i in range(n):
        want = bytes(cat.model(i, ic).model_desc())
        got = arr[i].tobytes()
        if got != want:
            d = cat.model(i, ic).model_desc()
            w = np.frombuffer(d, dtype=arr.dtype)[0]
            for name in arr.dtype.names:
                a, b = np.asarray(arr[i][name]), np.asarray(w[name])
                assert a.tobytes() == b.tobytes() or (name in ("plx_val", "plx_unc") and not w["has_parallax"]), (i, name, a, b)
    assert arr["has_parallax"][3] == 0 and arr["prior_distance"]["hi"][3] == 10000.0
    assert np.isclose(arr["prior_distance"]["hi"][5], 2000 / 0.05)
    assert np.isnan(arr["spec_val"][7, 0])
    # catalog-wide prior objects (reference catalog.py:117-124) reach both forms identically
    from isochrones_amd import priors as P
    cat.set_prior(distance=P.GaussianPrior(300.0, 100.0, bounds=(1.0, 900.0)), feh=P.FlatPrior((-0.8, 0.3)),
                  AV=P.PowerLawPrior(0.5, (0.0, 1.0)))
    arr2, _ = CatalogPosterior.build_descs(cat, ic)
    for i in (0, 3, 5, 11):
        w = np.frombuffer(cat.model(i, ic).model_desc(), dtype=arr2.dtype)[0]
        for name in arr2.dtype.names:
            a, b = np.asarray(arr2[i][name]), np.asarray(w[name])
            assert a.tobytes() == b.tobytes() or (name in ("plx_val", "plx_unc") and not w["has_parallax"]), (i, name, a, b)
    assert arr2["prior_distance"]["hi"][5] == 900.0 and arr2["prior_distance"]["kind"][0] == _cabi.PRIOR_GAUSS
    # the column form (what from_catalog sends to iso_catalog_create_columns) carries the same per-star numbers
    cat0 = ia.StarCatalog(df, bands=["G", "RP"], props=["parallax", "Teff"])
    cols, tmpl = CatalogPosterior.build_columns(cat0, ic)
    assert np.array_equal(cols["mag_val"], arr["mag_val"][:, :2]) and np.array_equal(cols["mag_unc"], arr["mag_unc"][:, :2])
    assert np.array_equal(cols["spec_val"], arr["spec_val"], equal_nan=True)
    assert np.array_equal(cols["has_plx"], arr["has_parallax"]) and np.array_equal(cols["dist_hi"], arr["prior_distance"]["hi"])
    has = cols["has_plx"] != 0
    assert np.array_equal(cols["plx_val"][has], arr["plx_val"][has]) and tmpl.bands == ["G", "RP"]
    cols2, _ = CatalogPosterior.build_columns(cat, ic)            # catalog-wide distance prior: its own bounds stay
    assert cols2["dist_hi"] is None


def test_fit_catalog_checkpoint_resume(tmp_path):
    """A rerun loads finished shards instead of refitting (single process, injected fit)."""
    import pandas as pd
    df = pd.DataFrame({"V_mag": np.linspace(8, 12, 9), "V_mag_unc": 0.02}, index=["s%d" % i for i in range(9)])
    cat = ia.StarCatalog(df, bands=["V"])
    calls = []

    def fake_fit(catalog, ic, indices, N=1, **kw):
        calls.append(len(indices))
        rows = np.zeros((len(indices), 3 * (N + 4) + 3))
        rows[:, 0] = np.asarray(indices) + 0.5
        return rows

    a = ia.fit_catalog(cat, ic=None_IC(), fit_fn=fake_fit, checkpoint_dir=str(tmp_path))
    b = ia.fit_catalog(cat, ic=None_IC(), fit_fn=fake_fit, checkpoint_dir=str(tmp_path))
    assert calls == [9] and a.equals(b)
    cat2 = ia.StarCatalog(df.iloc[:7], bands=["V"])          # different catalog -> checkpoint ignored
    ia.fit_catalog(cat2, ic=None_IC(), fit_fn=fake_fit, checkpoint_dir=str(tmp_path))
    assert calls == [9, 7]


def test_fit_catalog_checkpoint_depends_on_settings_and_data(tmp_path):
    """A stored shard is reused only for the same stars, measurements and fit settings."""
    import pandas as pd
    df = pd.DataFrame({"V_mag": np.linspace(8, 12, 6), "V_mag_unc": 0.02}, index=["s%d" % i for i in range(6)])
    calls = []

    def fake_fit(catalog, ic, indices, N=1, **kw):
        calls.append(dict(kw))
        return np.zeros((len(indices), 3 * (N + 4) + 3))

    run = lambda frame, **kw: ia.fit_catalog(ia.StarCatalog(frame, bands=["V"]), ic=None_IC(), fit_fn=fake_fit,
                                             checkpoint_dir=str(tmp_path), **kw)
    run(df, nwalkers=32, niter=100, seed=1)
    run(df, nwalkers=32, niter=100, seed=1)
    assert len(calls) == 1                                    # reused
    run(df, nwalkers=32, niter=100, seed=2)                   # other seed
    run(df, nwalkers=64, niter=100, seed=2)                   # other walker count
    df2 = df.copy()
    df2.loc["s3", "V_mag"] += 0.01                            # a measurement changed
    run(df2, nwalkers=64, niter=100, seed=2)
    assert len(calls) == 4
    run(df2, nwalkers=64, niter=100, seed=2)
    assert len(calls) == 4


def _failing_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import warnings
    import pandas as pd
    import torch.distributed as dist
    import isochrones_amd as ia_
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 9
    df = pd.DataFrame({"V_mag": np.linspace(8, 12, n), "V_mag_unc": 0.02}, index=["s%03d" % i for i in range(n)])
    cat = ia_.StarCatalog(df, bands=["V"])

    def fit(catalog, ic, indices, N=1, **kw):
        if rank == 1:
            raise ValueError("no star of the batch has all of the catalog's bands")
        rows = np.ones((len(indices), 3 * (N + 4) + 3))
        rows[:, 0] = np.asarray(indices)
        return rows

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = ia_.fit_catalog(cat, ic=None_IC(), fit_fn=fit)
    assert any("rank 1" in str(x.message) for x in w)
    res.to_pickle(os.path.join(out_dir, "res%d.pkl" % rank))
    with open(os.path.join(out_dir, "err%d.txt" % rank), "w") as f:
        f.write(repr(res.attrs["shard_errors"]))
    try:
        ia_.fit_catalog(cat, ic=None_IC(), fit_fn=fit, strict=True)
        raised = False
    except RuntimeError as e:
        raised = "rank 1" in str(e)
    with open(os.path.join(out_dir, "strict%d.txt" % rank), "w") as f:
        f.write(str(raised))
    dist.barrier()
    dist.destroy_process_group()


def test_fit_catalog_a_failing_rank_does_not_hang_the_others(tmp_path):
    """One rank's fit raises: it still enters the all-gather with NaN / ok = 0 rows, the other rank's stars arrive,
    every rank sees the error text; strict=True raises on every rank after the exchange."""
    import pandas as pd
    import torch.multiprocessing as mp
    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = pd.read_pickle(tmp_path / "res0.pkl"), pd.read_pickle(tmp_path / "res1.pkl")
    assert r0.equals(r1)
    owner = (np.arange(9) + 1) % 2
    assert np.array_equal(r0["ok"].values, (owner == 0).astype(float))
    assert np.isnan(r0.iloc[owner == 1, :-1].values).all() and np.array_equal(r0.iloc[owner == 0, 0].values, np.flatnonzero(owner == 0))
    for r in (0, 1):
        assert "no star of the batch" in open(tmp_path / ("err%d.txt" % r)).read()
        assert open(tmp_path / ("strict%d.txt" % r)).read() == "True"


def _bcast_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import isochrones_amd as ia_
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ic = None
    if rank == 0:           # only rank 0 "loads" the tables
        ic = ia_.synthetic_isochrone(bands=("J", "K"), ages=[9.0, 9.5, 10.0], fehs=[-0.5, 0.0, 0.5],
                                     eeps=np.arange(300., 340.), eep_bounds=(300, 339),
                                     limits=dict(age=(9.0, 10.0), feh=(-0.5, 0.5)))
    got = ia_.broadcast_interpolator(ic, src=0)
    np.savez(os.path.join(out_dir, "ic%d.npz" % rank), grid=got.model_grid.interp.grid, bc=got.bc_grid.interp.grid,
             ax0=got.model_grid.interp.index_columns[0], bcax3=got.bc_grid.interp.index_columns[3],
             cols=np.array(got.model_grid.interp.columns), bands=np.array(got.bands),
             lim=np.array(got.model_grid.get_limits("age")), eb=np.array(got.eep_bounds), kind=got.kind)
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_interpolator_world2_gloo(tmp_path):
    """the tables loaded on rank 0 reach every rank through one broadcast."""
    import torch.multiprocessing as mp
    mp.spawn(_bcast_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "ic0.npz"), np.load(tmp_path / "ic1.npz")
    for k in a.files:
        assert np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == "f" else np.array_equal(a[k], b[k]), k
    assert b["grid"].shape == (3, 3, 40, 16) and list(b["bands"]) == ["J", "K"] and int(b["kind"]) == 1


def test_fit_catalog_single_process_errors_propagate_and_strict_false_isolates():
    """One process: a failing fit_fn / a wrong result width is the caller's error (strict defaults to True there);
    strict=False keeps the per-shard isolation (NaN rows, ok = 0, RuntimeWarning)."""
    import pandas as pd
    df = pd.DataFrame({"V_mag": np.linspace(8, 12, 5), "V_mag_unc": 0.02}, index=["s%d" % i for i in range(5)])
    cat = ia.StarCatalog(df, bands=["V"])

    def bad_kw(catalog, ic, indices, N=1):
        return np.zeros((len(indices), 18))

    with pytest.raises(TypeError):                               # misspelled keyword reaches the caller as TypeError
        ia.fit_catalog(cat, ic=None_IC(), fit_fn=bad_kw, nwalkerz=32)
    with pytest.raises(ValueError, match="expected"):
        ia.fit_catalog(cat, ic=None_IC(), fit_fn=lambda c, i, idx, N=1: np.zeros((len(idx), 3)))
    with pytest.warns(RuntimeWarning, match="shard"):
        res = ia.fit_catalog(cat, ic=None_IC(), fit_fn=bad_kw, nwalkerz=32, strict=False)
    assert res["ok"].eq(0).all() and 0 in res.attrs["shard_errors"]
    assert set(res.attrs["timings"]) >= {"fit_s", "gather_s", "world"}


def test_checkpoint_digest_covers_the_interpolator_and_flags_unstable_settings(tmp_path):
    import pandas as pd
    from isochrones_amd.catalog import _shard_fingerprint
    df = pd.DataFrame({"V_mag": np.linspace(8, 12, 5), "V_mag_unc": 0.02}, index=["s%d" % i for i in range(5)])
    cat = ia.StarCatalog(df, bands=["V"])
    mine = np.arange(5)

    class TrackIC(None_IC):
        eep_replaces, bands = "age", ("V",)

    class IsoIC(None_IC):
        eep_replaces, bands = "mass", ("V",)

    a = _shard_fingerprint(cat, mine, 1, dict(nwalkers=32), TrackIC())
    assert a == _shard_fingerprint(cat, mine, 1, dict(nwalkers=32), TrackIC())
    assert a != _shard_fingerprint(cat, mine, 1, dict(nwalkers=32), IsoIC())          # other parametrisation
    IsoIC.eep_replaces = "age"
    IsoIC.bands = ("V", "J")
    assert a != _shard_fingerprint(cat, mine, 1, dict(nwalkers=32), IsoIC())          # other bands
    # the digest reads the arrays the fit reads (the catalog's snapshot), not the live frame
    cat.df = cat.df.assign(V_mag=99.0)                 # a new frame: the measurement arrays are what they were
    assert a == _shard_fingerprint(cat, mine, 1, dict(nwalkers=32), TrackIC())
    cat.measurements["V"][0][2] += 0.5
    assert a != _shard_fingerprint(cat, mine, 1, dict(nwalkers=32), TrackIC())
    # arrays are hashed by content; an object whose repr is its address is flagged
    k = dict(model_kwargs=dict(w=np.arange(3.0)))
    assert _shard_fingerprint(cat, mine, 1, k, TrackIC()) == _shard_fingerprint(cat, mine, 1, dict(model_kwargs=dict(w=np.arange(3.0))), TrackIC())
    with pytest.warns(RuntimeWarning, match="no stable representation"):
        _shard_fingerprint(cat, mine, 1, dict(callback=object()), TrackIC())
    # where the MIST caches are mounted is not part of the digest (the tables' content hash is), real-vs-synthetic is
    TrackIC.data_source = "/data/one/mist/tracks/full_grid_v1.2_vvcrit0.4.npz"
    b = _shard_fingerprint(cat, mine, 1, dict(nwalkers=32), TrackIC())
    TrackIC.data_source = "/mnt/elsewhere/full_grid_v1.2_vvcrit0.4.npz"
    assert b == _shard_fingerprint(cat, mine, 1, dict(nwalkers=32), TrackIC())
    TrackIC.data_source = "synthetic"
    assert b != _shard_fingerprint(cat, mine, 1, dict(nwalkers=32), TrackIC())


def test_checkpoint_digest_follows_the_start_point_method(monkeypatch):
    """the start-point kernel and the framework version draw different random numbers: a shard fitted with one is not the
    shard of the other"""
    import pandas as pd
    from isochrones_amd.catalog import _shard_fingerprint
    df = pd.DataFrame({"V_mag": np.linspace(8, 12, 5), "V_mag_unc": 0.02}, index=["s%d" % i for i in range(5)])
    cat = ia.StarCatalog(df, bands=["V"])
    monkeypatch.delenv("ISOCHRONES_AMD_START", raising=False)
    a = _shard_fingerprint(cat, np.arange(5), 1, dict(nwalkers=32), None_IC())
    monkeypatch.setenv("ISOCHRONES_AMD_START", "kernel")
    assert a == _shard_fingerprint(cat, np.arange(5), 1, dict(nwalkers=32), None_IC())
    monkeypatch.setenv("ISOCHRONES_AMD_START", "torch")
    assert a != _shard_fingerprint(cat, np.arange(5), 1, dict(nwalkers=32), None_IC())


# ---- eight ranks: the shape of the driver's scaling run (scripts/batch_starfit:60-62) -----------------------------
def _world8_worker(rank, world, port, n_stars, out_dir):
    """What one rank of `bench.py --gpus 8` does around its kernels: receive the tables from rank 0, fit its share of a
    10^4-star catalog (star i -> rank (i + 1) % 8), meet the others in ONE all-gather.  Rank 5's fit fails."""
    sys.path.insert(0, ROOT)
    import warnings
    import pandas as pd
    import torch.distributed as dist
    import isochrones_amd as ia_
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ic = None
    if rank == 0:
        ic = ia_.synthetic_isochrone(bands=("G", "BP", "RP"), ages=[9.0, 9.5, 10.0], fehs=[-0.5, 0.0, 0.5],
                                     eeps=np.arange(300., 340.), eep_bounds=(300, 339), limits=dict(age=(9.0, 10.0), feh=(-0.5, 0.5)))
    ic = ia_.broadcast_interpolator(ic, src=0)
    df = pd.DataFrame({"G_mag": np.linspace(8, 12, n_stars), "G_mag_unc": 0.02}, index=["s%05d" % i for i in range(n_stars)])
    cat = ia_.StarCatalog(df, bands=["G"])
    mine = ia_.shard_indices(n_stars, rank, world)
    seen = []

    def fit(catalog, ic_, indices, N=1, **kw):
        seen.extend(int(i) for i in indices)
        if rank == 5:
            raise RuntimeError("HIP error on this rank")
        rows = np.zeros((len(indices), 3 * (N + 4) + 3))
        rows[:, 0] = np.asarray(indices) * 2.0 + 1.0
        rows[:, -3] = rank
        rows[:, -1] = 1.0
        return rows

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = ia_.fit_catalog(cat, ic, N=1, fit_fn=fit)
    assert seen == [int(i) for i in mine]
    assert any("rank 5" in str(x.message) for x in w)
    assert res.attrs["timings"]["world"] == world and res.attrs["timings"]["stars_of_this_rank"] == len(mine)
    if rank in (0, 5, 7):
        res.to_pickle(os.path.join(out_dir, "res%d.pkl" % rank))
        np.save(os.path.join(out_dir, "grid%d.npy" % rank), ic.model_grid.interp.grid)
        with open(os.path.join(out_dir, "err%d.txt" % rank), "w") as f:
            f.write(repr(sorted(res.attrs["shard_errors"])))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_broadcast_fit_and_gather_with_uneven_shards_and_one_failing_rank(tmp_path):
    """World size 8 on gloo: the rank count of the driver's scaling run, a catalog whose size 8 does not divide (10 003
    stars: five ranks get 1 250, three get 1 251), one rank whose fit raises.  Every rank ends with the same table; the
    failing rank's stars are NaN / ok = 0, everyone else's arrive; nobody hangs."""
    import pandas as pd
    import torch.multiprocessing as mp
    n = 10_003
    mp.spawn(_world8_worker, args=(8, _free_port(), n, str(tmp_path)), nprocs=8, join=True)
    r0, r5, r7 = (pd.read_pickle(tmp_path / ("res%d.pkl" % r)) for r in (0, 5, 7))
    assert r0.equals(r5) and r0.equals(r7) and len(r0) == n
    owner = (np.arange(n) + 1) % 8                              # batch_starfit's rule: awk 'NR % NPROCS == TASK', NR = i + 1
    assert sorted(np.bincount(owner)) == [1250] * 5 + [1251] * 3
    ok = owner != 5
    assert np.array_equal(r0["ok"].values, ok.astype(float))
    assert np.array_equal(r0.iloc[ok, 0].values, np.flatnonzero(ok) * 2.0 + 1.0)
    assert np.array_equal(r0["lnpost_max"].values[ok], owner[ok].astype(float))
    assert np.isnan(r0.iloc[~ok, :-1].values).all()
    for r in (0, 5, 7):
        assert open(tmp_path / ("err%d.txt" % r)).read() == "[5]"
        assert np.array_equal(np.load(tmp_path / ("grid%d.npy" % r)), np.load(tmp_path / "grid0.npy"), equal_nan=True)
