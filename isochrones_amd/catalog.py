"""Catalog path: many independent stars, sharded over the GPUs of a node.

Reference semantics being reproduced:

* ``StarCatalog`` (isochrones/catalog.py:19-139): a DataFrame with ``{band}_mag``,
  ``{band}_mag_unc`` and ``{prop}``, ``{prop}_unc`` columns; ``iter_models(ic, N)`` yields one
  Single/Binary/TripleStarModel per row.
* ``scripts/batch_starfit:60-62``: line NR of the star list goes to worker ``NR % NPROCS``
  (NR is 1-based), workers never talk to each other, results are per-star.

Here one process drives one GPU (``torch.distributed``, backend ``nccl`` = RCCL on ROCm, ``gloo``
in the CPU tests).  Every rank builds the posteriors of *its* stars only and samples all of them
in lock-step: S stars x W walkers are one ``iso_catalog_lnpost`` launch per half-step (each row
carries its star's index).  There is no collective inside the sampling loop; the only exchange
is one all-gather of fixed-size per-star result rows at the end.
"""
from __future__ import annotations

import ctypes as C
import re

import os

import numpy as np

from . import _cabi, device as dev
from .starmodel import BasicStarModel


def shard_of(i: int, world: int) -> int:
    """Worker that owns star ``i`` (0-based): batch_starfit's ``NR % NPROCS`` with NR = i + 1."""
    return (i + 1) % world


def shard_indices(n: int, rank: int, world: int) -> np.ndarray:
    idx = np.arange(n)
    return idx[(idx + 1) % world == rank]


class StarCatalog:
    """Measurements of many stars, held column-wise.

    The reference's ``StarCatalog`` (isochrones/catalog.py:19-139) wraps a DataFrame whose columns follow the
    ``<band>_mag`` / ``<band>_mag_unc`` and ``<prop>`` / ``<prop>_unc`` naming and yields one star model per row.
    The same frame, names and accessors are accepted here, but the object is organised around what the batched
    device path consumes: every measurement is pulled out once as a (values, uncertainties) pair of float arrays
    (``measurements``), so building the per-star columns of :class:`CatalogPosterior` never touches a row object;
    per-row models are only made on request (``model(i)`` / ``iter_models``).

    df        frame with one row per star (index = star names)
    bands     photometric bands; default: every ``X`` with an ``X_mag`` column
    props     further measured quantities (``Teff``, ``logg``, ``feh``, ``parallax``, ...)
    no_uncs   accept a frame without uncertainty columns (uncertainties are then NaN)"""

    _MAG = re.compile(r"(.+)_mag$")

    def __init__(self, df, bands=None, props=None, no_uncs=False):
        self.df = df
        names = [str(c) for c in df.columns]
        self.bands = tuple(bands) if bands is not None else tuple(m.group(1) for m in map(self._MAG.match, names) if m)
        self.props = tuple(props or ())
        self._prior_settings = {}
        self.measurements = {}
        for key, column in [(b, b + "_mag") for b in self.bands] + [(q, q) for q in self.props]:
            missing = [c for c in (column, column + "_unc") if c not in names]
            if missing and not no_uncs:
                what = ("{} not in DataFrame!" if missing[0] == column else "{0} uncertainty ({0}_unc) not in DataFrame!")
                raise ValueError(what.format(column))
            val = df[column].to_numpy(dtype=float) if column in names else np.full(len(df), np.nan)
            unc = df[column + "_unc"].to_numpy(dtype=float) if column + "_unc" in names else np.full(len(df), np.nan)
            self.measurements[key] = (val, unc)

    @property
    def band_cols(self):
        return tuple(b + "_mag" for b in self.bands)

    def __len__(self):
        return len(self.df)

    def get_measurement(self, prop, values=False):
        """(values, uncertainties) of a band (``"J"`` or ``"J_mag"``) or property."""
        m = self._MAG.match(prop)
        key = m.group(1) if m and m.group(1) in self.measurements else prop
        return self.measurements[key]

    def iter_bands(self, **kwargs):
        return ((b, self.measurements[b]) for b in self.bands)

    def iter_props(self, **kwargs):
        return ((q, self.measurements[q]) for q in self.props)

    def set_prior(self, **kwargs):
        """Prior objects every model built from this catalog receives (reference: catalog.py:117-124).  A catalog is fitted
        inside the resident kernels, so these have to be families the device evaluates."""
        from .priors import is_host_prior
        for prop, prior in kwargs.items():
            if prop != "eep" and is_host_prior(prior):
                raise NotImplementedError("prior %r for %r is not evaluable on the device: a catalog fit needs one of "
                                          "isochrones_amd.priors.DEVICE_PRIOR_TYPES (single models accept any Prior)" % (prior, prop))
        self._prior_settings.update(kwargs)

    def model(self, i, ic, N=1, **kwargs):
        """The 1-3 star model of row ``i`` (what the reference's ``iter_models`` yields for it)."""
        obs = {key: (float(v[i]), float(u[i])) for key, (v, u) in self.measurements.items()}
        mod = BasicStarModel(ic, N=N, name=self.df.index[i], **obs, **kwargs)
        if self._prior_settings:
            mod.set_prior(**self._prior_settings)
        return mod

    def iter_models(self, ic=None, N=1, indices=None, **kwargs):
        if ic is None:
            from .models import get_ichrone
            ic = get_ichrone("mist", bands=self.bands)
        return (self.model(int(i), ic, N=N, **kwargs) for i in (range(len(self)) if indices is None else indices))

    def write_ini(self, ic=None, root=".", N=1, nest_directories=True, clobber=True):
        """One ``<name>/star.ini`` folder per star for the per-folder drivers (see starfit.write_catalog_ini)."""
        from .starfit import write_catalog_ini
        return write_catalog_ini(self, ic=ic, root=root, N=N, nest_directories=nest_directories, clobber=clobber)


class CatalogPosterior:
    """Device-resident posteriors of many stars sharing bands and multiplicity.

    Build it from a list of models, or — without creating one Python object per star — from a
    :class:`StarCatalog` with :meth:`from_catalog` (descriptor records filled column-wise)."""

    def __init__(self, ic, models=None, device=None, _descs=None, _template=None, _columns=None):
        self.ic = ic
        self.device = dev.current_device() if device is None else device
        h = C.c_void_p()
        if _columns is not None:
            # template descriptor + plain per-star columns; the device fills the per-star constant blocks
            self.models = None
            template = _template
            col = _columns
            n = int(col["mag_val"].shape[0])
            D = template.n_params
            d0 = template.model_desc()
            self.bounds_lo = np.tile(np.array([d0.bound_lo[j] for j in range(D)]), (n, 1))
            self.bounds_hi = np.tile(np.array([d0.bound_hi[j] for j in range(D)]), (n, 1))
            if col["dist_hi"] is not None:
                self.bounds_hi[:, list(template.param_names).index("distance")] = col["dist_hi"]
            has = col["has_plx"] != 0
            self.parallax = np.where(has, col["plx_val"], np.nan)
            self.parallax_unc = np.where(has, col["plx_unc"], np.nan)
            self._desc_array = None
            dp = C.POINTER(C.c_double)
            ptr = lambda a: a.ctypes.data_as(dp)
            _cabi.check(_cabi.lib().iso_catalog_create_columns(
                ic.handle(self.device), C.byref(d0), n, ptr(col["mag_val"]), ptr(col["mag_unc"]), ptr(col["spec_val"]),
                ptr(col["spec_unc"]), col["has_plx"].ctypes.data_as(C.POINTER(C.c_int32)), ptr(col["plx_val"]),
                ptr(col["plx_unc"]), ptr(col["dist_hi"]) if col["dist_hi"] is not None else None, C.byref(h)))
            self.n_models = n
        else:
            if _descs is None:
                if not models:
                    raise ValueError("no models")
                self.models = list(models)
                template = self.models[0]
                descs = (_cabi.IsoModelDesc * len(self.models))(*[m.model_desc() for m in self.models])
                arr = np.frombuffer(descs, dtype=np.dtype(_cabi.IsoModelDesc))
            else:
                self.models = None
                template = _template
                arr = _descs
                descs = (_cabi.IsoModelDesc * arr.shape[0]).from_buffer(arr)
            self.n_models = int(arr.shape[0])
            D = template.n_params
            self.bounds_lo = np.array(arr["bound_lo"][:, :D])
            self.bounds_hi = np.array(arr["bound_hi"][:, :D])
            has = arr["has_parallax"] != 0
            self.parallax = np.where(has, arr["plx_val"], np.nan)
            self.parallax_unc = np.where(has, arr["plx_unc"], np.nan)
            self._desc_array = arr
            _cabi.check(_cabi.lib().iso_catalog_create(ic.handle(self.device), descs, self.n_models, C.byref(h)))
        self.template = template
        self.n_params = template.n_params
        self.param_names = template.param_names
        self._h = h

    @classmethod
    def from_catalog(cls, catalog, ic, N=1, indices=None, device=None, **model_kwargs):
        """One template model + per-star columns straight from the DataFrame (no per-star Python objects and no
        per-star descriptors on the host: ``iso_catalog_create_columns`` builds the constant blocks on the device)."""
        cols, template = cls.build_columns(catalog, ic, N=N, indices=indices, **model_kwargs)
        return cls(ic, device=device, _columns=cols, _template=template)

    @staticmethod
    def _template_and_rows(catalog, ic, N, indices, model_kwargs):
        idx = np.arange(len(catalog)) if indices is None else np.asarray(indices, dtype=int)
        if idx.size == 0:
            raise ValueError("no stars")
        # template = the first star that has every band (stars lacking some get NaN entries).  The rows are taken from
        # the catalog's column arrays (StarCatalog.measurements), not through a DataFrame copy
        class _Rows:
            def __init__(self, cat, rows):
                self.cat, self.rows = cat, rows

            def pair(self, key):
                v, u = self.cat.measurements[key]
                return v[self.rows], u[self.rows]
        sub = _Rows(catalog, idx)
        complete = np.ones(idx.size, dtype=bool)
        for b in catalog.bands:
            v, u = sub.pair(b)
            complete &= ~(np.isnan(v) | np.isnan(u))
        if not complete.any():
            raise ValueError("no star of the batch has all of the catalog's bands")
        template = catalog.model(int(idx[int(np.argmax(complete))]), ic, N=N, **model_kwargs)
        if len(template.bands) != len([b for b in catalog.bands if b in ic.bc_grid.bands]):
            raise ValueError("template star lacks some of the catalog's bands")
        if "nu_max" in catalog.props or "delta_nu" in catalog.props:
            raise ValueError("asteroseismic terms are not batched")
        return idx, sub, template

    @staticmethod
    def build_columns(catalog, ic, N=1, indices=None, **model_kwargs):
        """Per-star columns for ``iso_catalog_create_columns`` (host only, numpy): magnitudes [n, nb] (NaN =
        band not observed), spectroscopic values [n, 3] (NaN = absent), parallax flags / values, and the
        parallax-dependent upper distance bound (reference: starmodel.py:1465-1475)."""
        idx, df, template = CatalogPosterior._template_and_rows(catalog, ic, N, indices, model_kwargs)
        n, bands = idx.size, template.bands
        mag_val = np.empty((n, len(bands))); mag_unc = np.empty((n, len(bands)))
        for j, b in enumerate(bands):
            v, u = df.pair(b)
            missing = np.isnan(v) | np.isnan(u)
            mag_val[:, j] = np.where(missing, np.nan, v)
            mag_unc[:, j] = np.where(missing, 1.0, u)
        spec_val = np.full((n, 3), np.nan); spec_unc = np.full((n, 3), np.nan)
        for q, name in enumerate(("Teff", "logg", "feh")):
            if name in catalog.props:
                v, u = df.pair(name)
                missing = np.isnan(v) | np.isnan(u)
                spec_val[:, q] = np.where(missing, np.nan, v)
                spec_unc[:, q] = np.where(missing, np.nan, u)
        has = np.zeros(n, dtype=np.int32); plx_val = np.zeros(n); plx_unc = np.ones(n)
        dist_hi = None
        if "parallax" in catalog.props:
            v, u = df.pair("parallax")
            ok = ~(np.isnan(v) | np.isnan(u))
            has = ok.astype(np.int32)
            plx_val = np.where(ok, v, 0.0)
            plx_unc = np.where(ok, u, 1.0)
            if "max_distance" not in model_kwargs and "distance" not in catalog._prior_settings:
                with np.errstate(divide="ignore", invalid="ignore"):
                    dist_hi = np.ascontiguousarray(np.where(ok & (v > 0), 1.0 / v * 2000,
                                                            np.where(ok & (v < 0), 1.0 / np.abs(u) * 2000, 10000.0)))
        cols = dict(mag_val=np.ascontiguousarray(mag_val), mag_unc=np.ascontiguousarray(mag_unc),
                    spec_val=np.ascontiguousarray(spec_val), spec_unc=np.ascontiguousarray(spec_unc),
                    has_plx=np.ascontiguousarray(has), plx_val=np.ascontiguousarray(plx_val),
                    plx_unc=np.ascontiguousarray(plx_unc), dist_hi=dist_hi)
        return cols, template

    @staticmethod
    def build_descs(catalog, ic, N=1, indices=None, **model_kwargs):
        """Vectorised descriptor records (host only): one template model (row ``indices[0]``)
        supplies priors, bands and flags; magnitudes, spectroscopic values, parallaxes and the
        parallax-dependent distance bound (reference: starmodel.py:1465-1475) are filled per star
        with numpy.  Returns (structured array of iso_model_desc, template model)."""
        idx = np.arange(len(catalog)) if indices is None else np.asarray(indices, dtype=int)
        if idx.size == 0:
            raise ValueError("no stars")
        # template = the first star that has every band (stars lacking some get NaN entries below)
        sub = catalog.df.iloc[idx]
        complete = np.ones(idx.size, dtype=bool)
        for b in catalog.bands:
            complete &= sub["{}_mag".format(b)].notna().to_numpy() & sub["{}_mag_unc".format(b)].notna().to_numpy()
        if not complete.any():
            raise ValueError("no star of the batch has all of the catalog's bands")
        template = catalog.model(int(idx[int(np.argmax(complete))]), ic, N=N, **model_kwargs)
        d0 = template.model_desc()
        dt = np.dtype(_cabi.IsoModelDesc)
        arr = np.empty(idx.size, dtype=dt)
        arr[:] = np.frombuffer(d0, dtype=dt)[0]
        df = catalog.df.iloc[idx]
        bands = template.bands
        if len(bands) != len([b for b in catalog.bands if b in ic.bc_grid.bands]):
            raise ValueError("template star lacks some of the catalog's bands")
        for j, b in enumerate(bands):
            v = df["{}_mag".format(b)].to_numpy(float)
            u = df["{}_mag_unc".format(b)].to_numpy(float)
            # a star without a measurement in this band: NaN value = "skip this term" for the catalog kernels
            # (the reference drops NaN measurements when it builds that star's model, starmodel.py:1427-1433)
            missing = np.isnan(v) | np.isnan(u)
            arr["mag_val"][:, j] = np.where(missing, np.nan, v)
            arr["mag_unc"][:, j] = np.where(missing, 1.0, u)
        for q, name in enumerate(("Teff", "logg", "feh")):
            if name in catalog.props:
                v = df[name].to_numpy(float)
                u = df[name + "_unc"].to_numpy(float)
                missing = np.isnan(v) | np.isnan(u)
                arr["spec_val"][:, q] = np.where(missing, np.nan, v)
                arr["spec_unc"][:, q] = np.where(missing, np.nan, u)
        if "nu_max" in catalog.props or "delta_nu" in catalog.props:
            raise ValueError("asteroseismic terms are not batched")
        i_d = list(template.param_names).index("distance")
        if "parallax" in catalog.props:
            v = df["parallax"].to_numpy(float)
            u = df["parallax_unc"].to_numpy(float)
            has = ~(np.isnan(v) | np.isnan(u))
            arr["has_parallax"] = has.astype(np.int32)
            arr["plx_val"] = np.where(has, v, 0.0)
            arr["plx_unc"] = np.where(has, u, 1.0)
            # (a distance prior installed through set_prior keeps its own bounds, as in the per-star models)
            if "max_distance" not in model_kwargs and "distance" not in catalog._prior_settings:
                default_hi = 10000.0
                with np.errstate(divide="ignore", invalid="ignore"):
                    hi = np.where(has & (v > 0), 1.0 / v * 2000, np.where(has & (v < 0), 1.0 / np.abs(u) * 2000,
                                                                           default_hi))
                arr["prior_distance"]["hi"] = hi
                arr["bound_hi"][:, i_d] = hi
        return arr, template

    def close(self):
        if getattr(self, "_h", None) is not None:
            _cabi.lib().iso_catalog_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lnpost(self, pars, star_id):
        """pars: CUDA float64 [n, n_params]; star_id: CUDA int32 [n] -> CUDA float64 [n]."""
        import torch
        if pars.dim() != 2 or pars.shape[1] != self.n_params or pars.dtype != torch.float64:
            raise ValueError("pars must be float64 [n, %d]" % self.n_params)
        if star_id.dtype != torch.int32 or star_id.numel() != pars.shape[0]:
            raise ValueError("star_id must be int32 [n]")
        pars = pars.contiguous()
        star_id = star_id.contiguous()
        n = pars.shape[0]
        out = dev.empty_f64((n,), self.device)
        if n:
            _cabi.check(_cabi.lib().iso_catalog_lnpost(self._h, dev.ptr(star_id), dev.ptr(pars), self.n_params, 1, n,
                                                       dev.ptr(out), dev.stream_ptr(self.device)))
        return out


class BatchedEnsembleSampler:
    """Stretch-move ensembles of S independent stars advanced in lock-step (one lnpost launch per
    half-step for all S x W/2 proposals).  ``lnpost_fn(pars[n, D], star_id[n]) -> [n]``."""

    def __init__(self, n_stars, nwalkers, ndim, lnpost_fn, a=2.0, seed=0, device=None):
        import torch
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("need an even number of walkers, at least 2*ndim")
        self.S, self.W, self.D, self.a, self.lnpost_fn = int(n_stars), int(nwalkers), int(ndim), float(a), lnpost_fn
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        h = self.W // 2
        self._sid_half = torch.arange(self.S, device=self.device, dtype=torch.int32).repeat_interleave(h)
        self._sid_full = torch.arange(self.S, device=self.device, dtype=torch.int32).repeat_interleave(self.W)
        self.naccepted = torch.zeros(self.S, self.W, dtype=torch.float64, device=self.device)
        self.iterations = 0

    def lnpost_all(self, pos):
        return self.lnpost_fn(pos.reshape(self.S * self.W, self.D), self._sid_full).view(self.S, self.W)

    def _half(self, pos, lnp, lo, hi, clo, chi):
        import torch
        S, h, D = self.S, self.W // 2, self.D
        idx = torch.randint(0, h, (S, h), generator=self.gen, device=self.device)
        u = torch.rand(S, h, generator=self.gen, device=self.device, dtype=torch.float64)
        z = ((self.a - 1.0) * u + 1.0) ** 2 / self.a
        xk = pos[:, lo:hi, :]
        xj = pos[:, clo:chi, :].gather(1, idx[..., None].expand(S, h, D))
        prop = xj + z[..., None] * (xk - xj)
        lnp_new = self.lnpost_fn(prop.reshape(S * h, D), self._sid_half).view(S, h)
        lnq = (D - 1) * torch.log(z) + lnp_new - lnp[:, lo:hi]
        logu = torch.log(torch.rand(S, h, generator=self.gen, device=self.device, dtype=torch.float64))
        acc = (logu < lnq) & torch.isfinite(lnp_new)
        pos[:, lo:hi, :] = torch.where(acc[..., None], prop, xk)
        lnp[:, lo:hi] = torch.where(acc, lnp_new, lnp[:, lo:hi])
        self.naccepted[:, lo:hi] += acc

    def run(self, pos, lnp, nsteps, keep=False):
        """Advance in place; with keep=True returns (chain [S, W, nsteps, D], lnprob [S, W, nsteps])."""
        import torch
        h = self.W // 2
        chain = torch.empty(self.S, self.W, nsteps, self.D, dtype=torch.float64, device=self.device) if keep else None
        lnps = torch.empty(self.S, self.W, nsteps, dtype=torch.float64, device=self.device) if keep else None
        for it in range(int(nsteps)):
            self._half(pos, lnp, 0, h, h, self.W)
            self._half(pos, lnp, h, self.W, 0, h)
            self.iterations += 1
            if keep:
                chain[:, :, it, :] = pos
                lnps[:, :, it] = lnp
        return chain, lnps


def initial_positions(post: CatalogPosterior, nwalkers, rng_seed=0, oversample=8, max_tries=6, method=None):
    """[S, W, D] start points with finite lnpost: draw ``oversample * W`` candidates per star inside
    the parameter bounds (distance centred on the parallax when there is one), evaluate them,
    keep each star's best W.  Stars that never reach W finite candidates are returned
    in ``failed`` (their rows hold NaN) — per-star failure isolation, as the reference's
    try/except around each star (isochrones/starfit.py:155-159).

    ``method`` (default: ``$ISOCHRONES_AMD_START`` or "kernel"): "kernel" = one launch of the start-point kernel
    (``iso_catalog_start_points``: one workgroup per star draws with Philox, evaluates with the catalog kernels' lnpost and
    selects in LDS; no candidate leaves the chip, no host synchronisation); "torch" = the framework version of rounds 1-3
    (candidates [D, S, K] in memory, one catalog-kernel launch and a dozen sort / topk / gather passes per attempt), kept
    as the A/B counterpart and for ensembles the kernel has no LDS for (more than 1024 walkers).  Both draw from the same
    candidate distribution; the random numbers differ - so the method is part of a stored shard's digest."""
    import torch
    method = method or os.environ.get("ISOCHRONES_AMD_START", "kernel")
    if method not in ("kernel", "torch"):
        raise ValueError("initial_positions: method must be 'kernel' or 'torch'")
    if method == "kernel" and hasattr(_cabi.lib(), "iso_catalog_start_points"):
        S, D, W = post.n_models, post.n_params, int(nwalkers)
        device = torch.device("cuda", post.device)
        best = torch.empty(S, W, D, dtype=torch.float64, device=device)
        best_lnp = torch.empty(S, W, dtype=torch.float64, device=device)
        failed = torch.empty(S, dtype=torch.int32, device=device)
        rc = _cabi.lib().iso_catalog_start_points(post._h, W, int(oversample), int(max_tries), int(rng_seed) & 0xFFFFFFFFFFFFFFFF,
                                                  dev.ptr(best), dev.ptr(best_lnp), dev.ptr(failed), dev.stream_ptr(post.device))
        if rc == 0:
            return best, best_lnp, failed.to(torch.bool)
        if rc != _cabi.ERR_INVALID:              # a HIP failure is an error, not a reason to change the random numbers
            _cabi.check(rc)
        import warnings
        warnings.warn("initial_positions: no start-point kernel for this shape (%s); using the framework version, whose "
                      "random numbers differ" % (_cabi.lib().iso_last_error() or b"").decode(), RuntimeWarning, stacklevel=2)
    S, D, W = post.n_models, post.n_params, nwalkers
    device = torch.device("cuda", post.device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(rng_seed))
    f64 = dict(dtype=torch.float64, device=device)
    lo = torch.as_tensor(post.bounds_lo, **f64)
    hi = torch.as_tensor(post.bounds_hi, **f64)
    names = list(post.param_names)
    i_d = names.index("distance")
    plx = torch.as_tensor(post.parallax, **f64)
    plx_e = torch.as_tensor(post.parallax_unc, **f64)
    n_eep = sum(1 for n in names if n.startswith("eep"))
    # the ranges candidates are drawn from: the stars' bounds cut to the tables (as the kernel does, fast/start_points.h: a
    # prior without finite bounds would give log(0) / inf maps, and nothing outside the table has a posterior)
    ic = post.ic
    max_, may = ic.model_grid.interp.index_columns, ic.bc_grid.interp.index_columns
    cuts = {"eep": (float(max_[2][0]), float(max_[2][-1])), "distance": (0.0, 1.0e5),
            "AV": (float(may[3][0]), float(may[3][-1])) if len(post.template.bands) else (0.0, 10.0)}
    if "mass" in names:          # evolution tracks: table axes (feh, mass, eep)
        cuts.update(feh=(float(max_[0][0]), float(max_[0][-1])), mass=(float(max_[1][0]), float(max_[1][-1])))
    else:                        # isochrones: table axes (age, feh, eep)
        cuts.update(age=(float(max_[0][0]), float(max_[0][-1])), feh=(float(max_[1][0]), float(max_[1][-1])))
    for j, nm in enumerate(names):
        c_lo, c_hi = cuts["eep" if nm.startswith("eep") else nm]
        lo[:, j].clamp_(min=c_lo)
        hi[:, j].clamp_(max=c_hi)
    K = oversample * W
    # per-star affine maps u -> a + b u of every parameter; mass and a prior-drawn distance are then exponentiated
    a, b = lo.clone(), hi - lo
    log_cols = torch.zeros(S, D, dtype=torch.bool, device=device)
    if "mass" in names:                             # log-uniform in mass
        j = names.index("mass")
        a[:, j], b[:, j] = torch.log(lo[:, j]), torch.log(hi[:, j] / lo[:, j])
        log_cols[:, j] = True
    dlo = torch.clamp(lo[:, i_d], min=1.0)
    d0 = 1000.0 / plx
    rel = torch.clamp(plx_e / plx, min=1e-3, max=0.3)
    use_plx = (plx > 0) & torch.isfinite(d0)
    a[:, i_d] = torch.where(use_plx, d0 * (1.0 - 4.0 * rel), torch.log(dlo))         # d0 (1 + 4 rel (2u - 1)) | log-uniform
    b[:, i_d] = torch.where(use_plx, 8.0 * rel * d0, torch.log(hi[:, i_d] / dlo))
    log_cols[:, i_d] = ~use_plx
    cand = torch.empty(D, S, K, **f64)
    best = torch.full((S, W, D), float("nan"), **f64)
    best_lnp = torch.full((S, W), -float("inf"), **f64)
    sid = torch.arange(S, device=device, dtype=torch.int32).repeat_interleave(K)
    out = dev.empty_f64((S * K,), post.device)
    for attempt in range(max_tries):
        for j in range(D):
            plane = cand[j]
            plane.uniform_(0.0, 1.0, generator=gen)
            plane.mul_(b[:, j, None]).add_(a[:, j, None])
            if bool(log_cols[:, j].all()):
                plane.exp_()
            elif bool(log_cols[:, j].any()):
                rows = log_cols[:, j]
                plane[rows] = torch.exp(plane[rows])
        if n_eep > 1:                               # eep_0 >= eep_1 >= eep_2
            cand[:n_eep] = torch.sort(cand[:n_eep], dim=0, descending=True).values
        _cabi.check(_cabi.lib().iso_catalog_lnpost(post._h, dev.ptr(sid), dev.ptr(cand), 1, S * K, S * K, dev.ptr(out),
                                                   dev.stream_ptr(post.device)))
        lnp = out.view(S, K)
        lnp.masked_fill_(~torch.isfinite(lnp), -float("inf"))
        if attempt == 0:                            # nothing kept yet: the best W of the K new ones
            top = torch.topk(lnp, W, dim=1)
            best_lnp = top.values
            best = torch.stack([cand[j].gather(1, top.indices) for j in range(D)], dim=-1)
        else:
            top = torch.topk(torch.cat([best_lnp, lnp], dim=1), W, dim=1)
            best_lnp = top.values
            best = torch.stack([torch.cat([best[..., j], cand[j]], dim=1).gather(1, top.indices) for j in range(D)], dim=-1)
        if bool(torch.isfinite(best_lnp).all()):
            break
    failed = ~torch.isfinite(best_lnp).all(dim=1)
    best[failed] = float("nan")
    return best.contiguous(), best_lnp.contiguous(), failed


RESULT_STATS = ("median", "p16", "p84")


def result_columns(param_names):
    cols = []
    for p in param_names:
        cols += ["%s_%s" % (p, s) for s in RESULT_STATS]
    return cols + ["lnpost_max", "acceptance", "ok"]


def fit_stars_gpu(catalog: StarCatalog, ic, indices, N=1, nwalkers=32, nburn=150, niter=100, seed=0,
                  model_kwargs=None, fused=True, timings=None, max_stars_per_batch=200_000, return_chains=False,
                  replay_record=None):
    """Fit the stars ``indices`` of the catalog on the current GPU; returns [len(indices), 3*D+3]
    float64 numpy rows (result_columns order).  ``return_chains=True`` (fused sampler, one batch): also the
    stored chain [S, W, niter, D] and its lnpost values [S, W, niter] as CUDA tensors.  ``replay_record`` (a dict, tests):
    filled with what a move-by-move replay of the SAMPLING run needs - the ensembles as burn-in left them (``pos`` [S, W, D],
    ``lnp`` [S, W]), the sampler's ``seed``, the step counter the run starts at (``step0`` = nburn), the start points."""
    import torch
    import time as _time

    def _mark(name, _t=[_time.perf_counter()]):
        if timings is not None:
            torch.cuda.synchronize()
            now = _time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - _t[0]
            _t[0] = now

    if len(indices) == 0:
        return np.empty((0, 3 * (N + 4) + 3))
    if return_chains and (not fused or len(indices) > max_stars_per_batch):
        raise ValueError("return_chains needs the fused sampler and at most max_stars_per_batch stars")
    if len(indices) > max_stars_per_batch:
        # bound the device memory of the stored chains (S x W x niter x D doubles): fit the shard in slices
        parts = [fit_stars_gpu(catalog, ic, indices[k:k + max_stars_per_batch], N=N, nwalkers=nwalkers, nburn=nburn,
                               niter=niter, seed=seed + 7919 * (k // max_stars_per_batch), model_kwargs=model_kwargs,
                               fused=fused, timings=timings, max_stars_per_batch=max_stars_per_batch)
                 for k in range(0, len(indices), max_stars_per_batch)]
        return np.concatenate(parts, axis=0)
    post = CatalogPosterior.from_catalog(catalog, ic, N=N, indices=indices, **(model_kwargs or {}))
    _mark("build_posteriors")
    D = post.n_params
    pos, lnp, failed = initial_positions(post, nwalkers, rng_seed=seed)
    _mark("initial_positions")
    good = ~failed
    lean = (fused and hasattr(_cabi.lib(), "iso_catalog_patch_failed")
            and os.environ.get("ISOCHRONES_AMD_CATALOG_LEAN", "1") != "0")       # (0: the host-checked path of rounds 1-5, A/B and tests)
    if lean:
        # the whole fit stays on the stream: failed stars borrow the first good star's walkers with lnpost 0 (one small
        # launch instead of three device-to-host round trips and a dozen indexing launches), no start point is tested on the
        # host, and a batch without any good star shows up as rows that all say ok = 0
        fi = failed.to(torch.int32)
        _cabi.check(_cabi.lib().iso_catalog_patch_failed(post._h, int(nwalkers), dev.ptr(pos), dev.ptr(lnp), dev.ptr(fi),
                                                         dev.stream_ptr(post.device)))
    else:
        if not bool(good.any()):
            # no star of the batch found a start point: nothing to sample, every row reports ok = 0
            post.close()
            out = np.full((post.n_models, 3 * D + 3), np.nan)
            out[:, 3 * D + 2] = 0.0
            if return_chains:
                raise ValueError("no star of the batch has a start point with a finite lnpost")
            return out
        # failed stars get a copy of a good star's walkers so the batch stays rectangular
        if bool(failed.any()) and bool(good.any()):
            src = int(torch.nonzero(good)[0])
            pos[failed] = pos[src].clone()      # (a view of the tensor written to: torch refuses the aliasing)
            lnp[failed] = lnp[src].clone()
    if fused:
        from .sampler import FusedEnsembleSampler
        if not lean and bool(failed.any()) and bool(good.any()):
            # a failed star keeps its own (hopeless) posterior: its borrowed walkers simply never move
            lnp = torch.where(failed[:, None], torch.zeros_like(lnp), lnp)
        sampler = FusedEnsembleSampler(post, nwalkers, seed=seed + 1)
        if replay_record is not None:
            replay_record.update(start_pos=pos.clone(), start_lnp=lnp.clone(), failed=failed.clone())
        pos, lnp = sampler.run_mcmc(pos, nburn, lnprob0=lnp, store=False, check=not lean, inplace=lean)
        _mark("burn_in")
        if replay_record is not None:
            replay_record.update(pos=pos.clone(), lnp=lnp.clone(), seed=seed + 1, step0=int(nburn))
        sampler.reset()
        sampler.run_mcmc(pos, niter, lnprob0=lnp, store=True, check=not lean, inplace=lean)
        _mark("sampling")
        chain, lnps = sampler.chain, sampler.lnprobability        # [S, W, niter, D], [S, W, niter]
        acc_frac = sampler.acceptance_fraction.mean(dim=1)
    else:
        sampler = BatchedEnsembleSampler(post.n_models, nwalkers, D, post.lnpost, seed=seed + 1,
                                         device=torch.device("cuda", post.device))
        if bool(failed.any()) and bool(good.any()):
            # evaluate failed stars with the borrowed star's id so their lnpost stays finite
            sid_map = torch.arange(post.n_models, device=pos.device, dtype=torch.int32)
            sid_map[failed] = src
            sampler._sid_half = sid_map.repeat_interleave(nwalkers // 2)
            sampler._sid_full = sid_map.repeat_interleave(nwalkers)
        sampler.run(pos, lnp, nburn)
        sampler.naccepted.zero_()
        sampler.iterations = 0
        chain, lnps = sampler.run(pos, lnp, niter, keep=True)
        acc_frac = sampler.naccepted.mean(dim=1) / max(sampler.iterations, 1)
    if fused:
        q = sampler.quantiles((0.5, 0.16, 0.84))                                                # [S, D, 3], LDS sort per (star, parameter)
    else:
        # quantiles along a contiguous last axis (sorting a strided middle axis is several times slower)
        flat = chain.reshape(post.n_models, nwalkers * niter, D).permute(0, 2, 1).contiguous()  # [S, D, W*niter]
        srt = torch.sort(flat, dim=2).values
        m = srt.shape[2]
        pick = torch.tensor([0.5, 0.16, 0.84], dtype=torch.float64, device=flat.device) * (m - 1)
        i0 = pick.floor().long()
        i1 = torch.clamp(i0 + 1, max=m - 1)
        frac = pick - i0.to(torch.float64)
        q = srt[:, :, i0] * (1 - frac) + srt[:, :, i1] * frac                                   # [S, D, 3] (linear, as np.percentile)
    rows = torch.empty(post.n_models, 3 * D + 3, dtype=torch.float64, device=q.device)
    rows[:, : 3 * D] = q.reshape(post.n_models, 3 * D)
    if fused:       # storage order is [step][star * W + walker]: reduce over steps first (coalesced), then walkers
        rows[:, 3 * D] = sampler._lnprob.amax(dim=0).view(post.n_models, nwalkers).amax(dim=1)
    else:
        rows[:, 3 * D] = lnps.amax(dim=(1, 2))
    rows[:, 3 * D + 1] = acc_frac
    rows[:, 3 * D + 2] = good.to(torch.float64)
    rows[failed, : 3 * D + 2] = float("nan")
    out = rows.cpu().numpy()
    _mark("summaries")
    if return_chains:
        if not out[:, 3 * D + 2].any():
            post.close()
            raise ValueError("no star of the batch has a start point with a finite lnpost")
        kept = (chain.clone(), lnps.clone())
        post.close()
        return out, kept[0], kept[1]
    post.close()
    return out


#: bump when the stored result rows change meaning (columns, summaries, sampler defaults)
SHARD_FORMAT = 4


def _stable_repr(key, value):
    """A representation of a fit setting that is the same in every process: plain data as is, arrays by content.
    Anything whose default ``repr`` carries a memory address (callables, objects without ``__repr__``) has no such
    form: the shard digest then changes from run to run and the shard is refitted every time - say so once."""
    import warnings
    if value is None or isinstance(value, (bool, int, float, str, bytes)):
        return repr(value)
    if isinstance(value, np.ndarray):
        return "ndarray%s%s:%s" % (value.shape, value.dtype, value.tobytes().hex())
    if isinstance(value, (list, tuple)):
        return "[%s]" % ",".join(_stable_repr(key, v) for v in value)
    if isinstance(value, dict):
        return "{%s}" % ",".join("%r:%s" % (k, _stable_repr(key, v)) for k, v in sorted(value.items(), key=lambda kv: repr(kv[0])))
    try:
        from .priors import prior_to_spec
        return repr(prior_to_spec(value))
    except Exception:       # noqa: BLE001 - not a prior with a plain-data form
        pass
    text = repr(value)
    if " at 0x" in text:
        warnings.warn("fit_catalog: the setting %r has no stable representation (%s); a stored shard can never match "
                      "it and will be refitted on every run" % (key, type(value).__name__), RuntimeWarning, stacklevel=3)
    return text


def _shard_fingerprint(catalog, mine, N, fit_kwargs, ic=None):
    """Digest of everything a stored shard depends on: the stars' names and the measurement arrays the fit reads
    (``StarCatalog.measurements`` - the snapshot taken at construction, which is what ``CatalogPosterior`` consumes, not
    the live DataFrame), the multiplicity, the fit settings, the catalog's prior settings, the interpolator (its
    parametrisation, bands and table shapes) and the format of the stored rows."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr(("format", SHARD_FORMAT)).encode())
    h.update(repr([str(catalog.df.index[i]) for i in mine]).encode())
    for key in sorted(catalog.measurements):
        v, u = catalog.measurements[key]
        h.update(key.encode() + b"\0" + np.ascontiguousarray(v[mine], dtype=np.float64).tobytes()
                 + np.ascontiguousarray(u[mine], dtype=np.float64).tobytes())
    h.update(repr((int(N), tuple(catalog.bands), tuple(catalog.props))).encode())
    h.update(repr(sorted((k, _stable_repr(k, v)) for k, v in fit_kwargs.items() if k != "timings")).encode())
    h.update(repr(sorted((k, _stable_repr(k, v)) for k, v in catalog._prior_settings.items())).encode())
    h.update(repr(_ic_signature(ic)).encode())
    h.update(repr(("start", os.environ.get("ISOCHRONES_AMD_START", "kernel"))).encode())     # the two methods draw different numbers
    return h.hexdigest()


def _table_content_hash(interp):
    """sha256 over a table's axes and a strided sample of its values (at most ~2^16 doubles, NaN padding included):
    tables of the same shape - the synthetic MIST-shaped fallback and the real grid, two grid versions, two vvcrit
    values - differ in it, and it costs a few milliseconds even on the 724-MB track table."""
    import hashlib
    grid = getattr(interp, "grid", None)
    if grid is None:
        return None
    h = hashlib.sha256()
    for ax in getattr(interp, "index_columns", ()) or ():
        h.update(np.ascontiguousarray(ax, dtype=np.float64).tobytes())
    flat = np.asarray(grid).reshape(-1)
    step = max(1, flat.size // 65521)          # a prime number of samples: no resonance with the table's strides
    h.update(np.ascontiguousarray(flat[::step], dtype=np.float64).tobytes())
    h.update(repr((flat.size, step)).encode())
    return h.hexdigest()


def _ic_signature(ic):
    """What identifies the interpolator a shard was fitted with: parametrisation, bands, table shapes, EEP bounds, the
    kind of source the tables came from ('synthetic' or 'mist' - not the path: a shard stays valid when $ISOCHRONES moves
    or another host mounts it elsewhere) and a content hash of both tables (whatever of these the object has - fit_fn may
    be handed any stand-in)."""
    sig = [type(ic).__name__]
    for name in ("eep_replaces", "bands", "eep_bounds", "param_names"):
        v = getattr(ic, name, None)
        sig.append((name, tuple(v) if isinstance(v, (list, tuple)) else v))
    src = getattr(ic, "data_source", None)
    sig.append(("data_source", src if src in (None, "synthetic") else "mist"))
    for name in ("model_grid", "bc_grid"):
        interp = getattr(getattr(ic, name, None), "interp", None)
        grid = getattr(interp, "grid", None)
        sig.append((name, tuple(grid.shape) if grid is not None else None, _table_content_hash(interp)))
    return tuple(sig)


def fit_catalog(catalog: StarCatalog, ic, N=1, fit_fn=None, checkpoint_dir=None, strict=None, **fit_kwargs):
    """Shard the catalog over the ranks of the default process group (star i -> rank (i+1) % P),
    fit every shard with ``fit_fn`` (default: :func:`fit_stars_gpu`) and all-gather the per-star
    result rows.  Returns a DataFrame indexed like ``catalog.df`` on every rank.

    Failure isolation (runs with more than one rank): a rank whose shard cannot be fitted (an exception in ``fit_fn``)
    contributes NaN rows with ``ok = 0`` and still takes part in every collective, so the other ranks' stars are not lost
    and nobody waits for a rank that has left; the error texts come back in ``result.attrs["shard_errors"]``
    ({rank: message}) with a RuntimeWarning on every rank - or, with ``strict=True``, as a RuntimeError raised on every
    rank after the exchange.  (The reference wraps each star's fit in try/except, isochrones/starfit.py:155-159.)
    In a single-process run there is nobody to wait for, so ``strict`` defaults to True there and the original exception
    propagates (a misspelled keyword or an unwritable directory is an error, not an all-NaN table).

    ``checkpoint_dir``: every rank stores its finished shard there (``shard_{rank}of{world}.npz``)
    and a rerun loads it instead of refitting - the reference's "skip stars whose results already exist"
    (isochrones/starfit.py:66-77).  A stored shard is reused only if the stars, their measurements, the interpolator
    and the fit settings are the ones it was made with (a digest of all of them is stored next to the rows).

    ``result.attrs["timings"]``: this rank's seconds in the fit (``fit_s``) and in the result exchange (``gather_s``), and
    ``phases``: the fit by phase (``build_s`` per-star blocks, ``start_s`` start points, ``sample_s`` burn-in + sampling,
    ``summary_s`` quantile summaries and the copy of the rows to the host)."""
    import time as _time
    import warnings
    import pandas as pd
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    if strict is None:
        strict = world == 1
    n = len(catalog)
    mine = shard_indices(n, rank, world)
    fit_fn = fit_fn or fit_stars_gpu
    width = 3 * (N + 4) + 3
    rows, ckpt, error = None, None, None
    phases = {}
    t_fit = _time.perf_counter()

    def fit_shard():
        import os
        rows, ckpt = None, None
        if checkpoint_dir is not None:
            os.makedirs(checkpoint_dir, exist_ok=True)
            ckpt = os.path.join(checkpoint_dir, "shard_%dof%d.npz" % (rank, world))
            digest = _shard_fingerprint(catalog, mine, N, fit_kwargs, ic)
            if os.path.exists(ckpt):
                try:
                    with np.load(ckpt, allow_pickle=False) as z:
                        if np.array_equal(z["indices"], mine) and str(z["digest"]) == digest:
                            rows = z["rows"]
                except Exception:        # noqa: BLE001 - an unreadable shard file is refitted
                    rows = None
        if rows is None:
            kw = dict(fit_kwargs)
            if fit_fn is fit_stars_gpu and "timings" not in kw:
                kw["timings"] = phases           # where this rank's fit spends its time (attrs["timings"]["phases"])
            rows = np.asarray(fit_fn(catalog, ic, mine, N=N, **kw), dtype=np.float64)
            if rows.shape != (len(mine), width):
                raise ValueError("fit_fn returned rows of shape %s, expected %s" % (rows.shape, (len(mine), width)))
            if ckpt is not None:
                tmp = ckpt + ".tmp.npz"
                np.savez(tmp, indices=mine, rows=rows, digest=np.array(digest))
                os.replace(tmp, ckpt)
        return rows

    if world == 1 and strict:
        rows = fit_shard()                # single process: errors are the caller's, as they come
    else:
        try:
            rows = fit_shard()
        except Exception as e:           # noqa: BLE001 - this rank's stars are lost, the job is not
            error = "%s: %s" % (type(e).__name__, e)
            rows = np.full((len(mine), width), np.nan)
            rows[:, -1] = 0.0
    if torch.cuda.is_available() and distributed and dist.get_backend() == "nccl":
        torch.cuda.synchronize()
    t_gather = _time.perf_counter()
    full = np.full((n, width), np.nan)
    errors = {}
    if distributed:
        cap = (n + world - 1) // world + 1                       # fixed-size exchange buffers
        backend = dist.get_backend()
        devt = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        buf = torch.full((cap, width + 1), float("nan"), dtype=torch.float64, device=devt)
        buf[: len(mine), 0] = torch.as_tensor(mine, dtype=torch.float64)
        buf[: len(mine), 1:] = torch.as_tensor(rows, dtype=torch.float64)
        gathered = torch.empty((world, cap, width + 1), dtype=torch.float64, device=devt)
        if backend == "nccl":
            dist.all_gather_into_tensor(gathered, buf)           # one RCCL all-gather into one buffer
        else:
            parts = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(parts, buf)
            gathered = torch.stack(parts)
        g = gathered.reshape(world * cap, width + 1).cpu().numpy()
        ok = np.isfinite(g[:, 0])
        full[g[ok, 0].astype(int)] = g[ok, 1:]
        # the error texts: one flag reduction first, the (slower) object exchange only when some rank failed
        flag = torch.tensor([1.0 if error is not None else 0.0], dtype=torch.float64, device=devt)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if float(flag[0]) > 0:
            texts = [None] * world
            dist.all_gather_object(texts, error)
            errors = {r: t for r, t in enumerate(texts) if t is not None}
    else:
        full[mine] = rows
        if error is not None:
            errors = {0: error}
    t_end = _time.perf_counter()
    if errors:
        msg = "fit_catalog: shard(s) failed - " + "; ".join("rank %d: %s" % kv for kv in sorted(errors.items()))
        if strict:
            raise RuntimeError(msg)
        warnings.warn(msg, RuntimeWarning)
    names = (ic.param_names if N == 1 else tuple(["eep_%d" % i for i in range(N)] + list(ic.param_names[1:])))
    out = pd.DataFrame(full, index=catalog.df.index, columns=result_columns(names))
    out.attrs["shard_errors"] = errors
    out.attrs["timings"] = {"fit_s": t_gather - t_fit, "gather_s": t_end - t_gather, "world": world, "rank": rank,
                            "backend": dist.get_backend() if distributed else None, "stars_of_this_rank": int(len(mine)),
                            # this rank's shard by phase: per-star blocks, start points, burn-in + sampling, summaries (+ D2H)
                            "phases": {"build_s": phases.get("build_posteriors", 0.0), "start_s": phases.get("initial_positions", 0.0),
                                       "sample_s": phases.get("burn_in", 0.0) + phases.get("sampling", 0.0),
                                       "summary_s": phases.get("summaries", 0.0)}}
    return out


def synthetic_catalog(ic, n_stars, bands=None, seed=0, mag_unc=0.02, with_parallax=True, device=None):
    """Catalog of single stars drawn over the table (mags synthesised with this build's own
    interp_mag + Gaussian noise); returns (StarCatalog, truth DataFrame).  SURVEY 8d cfg 5."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    bands = list(bands or ic.bands)
    cols = {}
    truth = None
    need = n_stars
    chunks = []
    while need > 0:
        m = int(need * 1.6) + 64
        if ic.eep_replaces == "age":
            p = np.array([np.exp(rng.uniform(np.log(0.6), np.log(2.5), m)), rng.uniform(202, 605, m),
                          rng.uniform(-1.0, 0.4, m), np.exp(rng.uniform(np.log(50), np.log(1500), m)),
                          rng.uniform(0.0, 0.6, m)])
        else:
            p = np.array([rng.uniform(202, 605, m), rng.uniform(8.5, 10.0, m), rng.uniform(-1.0, 0.4, m),
                          np.exp(rng.uniform(np.log(50), np.log(1500), m)), rng.uniform(0.0, 0.6, m)])
        T, g, f, mags = ic.interp_mag(list(p), bands)
        ok = np.isfinite(mags).all(axis=1) & np.isfinite(T)
        chunks.append((p[:, ok], T[ok], g[ok], f[ok], mags[ok]))
        need -= int(ok.sum())
    p = np.concatenate([c[0] for c in chunks], axis=1)[:, :n_stars]
    T, g, f = (np.concatenate([c[k] for c in chunks])[:n_stars] for k in (1, 2, 3))
    mags = np.concatenate([c[4] for c in chunks])[:n_stars]
    for j, b in enumerate(bands):
        cols["%s_mag" % b] = mags[:, j] + mag_unc * rng.standard_normal(n_stars)
        cols["%s_mag_unc" % b] = np.full(n_stars, mag_unc)
    props = []
    if with_parallax:
        plx = 1000.0 / p[3]
        cols["parallax"] = plx * (1 + 0.02 * rng.standard_normal(n_stars))
        cols["parallax_unc"] = 0.02 * plx
        props.append("parallax")
    cols["Teff"] = T + 80 * rng.standard_normal(n_stars)
    cols["Teff_unc"] = np.full(n_stars, 80.0)
    props.append("Teff")
    df = pd.DataFrame(cols, index=["star%05d" % i for i in range(n_stars)])
    truth = pd.DataFrame(p.T, columns=list(ic.param_names), index=df.index)
    return StarCatalog(df, bands=bands, props=props), truth


def broadcast_interpolator(ic=None, src=0, rebuild_on_src=False, timings=None):
    """Give every rank the interpolator that only rank ``src`` has loaded (one broadcast
    of the tables at start-up instead of P reads of the table files).  Metadata travels as a small
    pickled object, the dense float64 tables as tensors (RCCL over xGMI with the ``nccl`` backend —
    the model table is ~0.7 GB —, host memory with ``gloo``).  Returns a
    ModelGridInterpolator on every rank; without an initialised process group it returns ``ic``.

    ``rebuild_on_src``: rank ``src`` also rebuilds its interpolator from the broadcast buffers instead of returning
    the one it passed in (what a one-rank group needs to exercise the route at all: tests).  ``timings`` (dict):
    receives ``broadcast_s`` (collectives + device-to-host copies) and ``rebuild_s`` (interpolator construction)."""
    import time as _time
    import torch
    import torch.distributed as dist
    from .interp import DFInterpolator
    from .models import (BolometricCorrectionGrid, EvolutionTrackGrid, EvolutionTrackInterpolator, IsochroneGrid,
                         IsochroneInterpolator)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not rebuild_on_src):
        return ic
    t0 = _time.perf_counter()
    rank = dist.get_rank()
    # (ISOCHRONES_AMD_BROADCAST=device with gloo: CUDA tensors through gloo's staged broadcast - how the receiving side of the
    # device route is exercised with more than one rank on a box with one GPU, which RCCL refuses to share between ranks)
    nccl = dist.get_backend() == "nccl"
    use_gpu = nccl or (os.environ.get("ISOCHRONES_AMD_BROADCAST") == "device" and torch.cuda.is_available())
    devt = torch.device("cuda", torch.cuda.current_device()) if use_gpu else torch.device("cpu")
    meta = [None]
    if rank == src:
        m, b = ic.model_grid.interp, ic.bc_grid.interp
        meta[0] = dict(kind=ic.kind, bands=list(ic.bands), eep_bounds=tuple(ic.eep_bounds),
                       model=dict(shape=m.grid.shape, columns=list(m.columns), names=list(m.index_names),
                                  limits=dict(ic.model_grid._limits)),
                       bc=dict(shape=b.grid.shape, columns=list(b.columns), names=list(b.index_names),
                               bands=list(ic.bc_grid.bands or b.columns)))
    dist.broadcast_object_list(meta, src=src, device=devt if nccl else None)
    meta = meta[0]
    keep = rank != src or rebuild_on_src

    # Over RCCL the tables travel device to device and STAY there: a receiving rank hands the tensor that arrived to the
    # library as it is (DFInterpolator.from_device -> iso_table_create_from_device, one device-to-device copy) instead of
    # downloading 0.7 GB to a numpy array and uploading it again when the interpolator is first used; the host copy
    # behind `.grid` is made only if somebody asks for it (checkpoint digests, table ingest).  gloo: host arrays, as before.
    # ISOCHRONES_AMD_BROADCAST=host keeps the round trip on RCCL as well (A/B, tests).
    on_device = use_gpu and os.environ.get("ISOCHRONES_AMD_BROADCAST", "device") != "host"

    def bcast(arr, shape, table=False):
        t = (torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float64), device=devt) if rank == src
             else torch.empty(shape, dtype=torch.float64, device=devt))
        dist.broadcast(t, src=src)
        if not keep:
            return None
        return t if (table and on_device) else t.cpu().numpy()

    tables = {}
    for key in ("model", "bc"):
        shape = tuple(meta[key]["shape"])
        srcobj = (ic.model_grid.interp if key == "model" else ic.bc_grid.interp) if rank == src else None
        grid = bcast(srcobj.grid if srcobj is not None else None, shape, table=True)
        axes = [bcast(srcobj.index_columns[d] if srcobj is not None else None, (shape[d],)) for d in range(len(shape) - 1)]
        tables[key] = (grid, axes)
    if use_gpu:
        torch.cuda.synchronize()
    t1 = _time.perf_counter()
    if timings is not None:
        timings["broadcast_s"] = t1 - t0
        timings["broadcast_bytes"] = int(sum(8 * int(np.prod(meta[k]["shape"])) for k in ("model", "bc")))
        timings["tables_stay_on_device"] = bool(on_device)
    if not keep:
        return ic
    mg_cls, ic_cls = ((EvolutionTrackGrid, EvolutionTrackInterpolator) if meta["kind"] == _cabi.KIND_TRACK
                      else (IsochroneGrid, IsochroneInterpolator))
    def table_of(key):
        grid, axes = tables[key]
        make = DFInterpolator.from_device if dev.is_tensor(grid) else DFInterpolator.from_arrays
        return make(grid, axes, meta[key]["columns"], meta[key]["names"])
    mg = mg_cls(table_of("model"), limits=meta["model"]["limits"])
    bcg = BolometricCorrectionGrid(table_of("bc"), bands=meta["bc"]["bands"])
    out = ic_cls(mg, bcg, bands=meta["bands"], eep_bounds=meta["eep_bounds"])
    if timings is not None:
        timings["rebuild_s"] = _time.perf_counter() - t1
    return out
