"""``ModelGridInterpolator`` — a stellar-model table bound to a bolometric-correction table.

Keeps the reference's call surface for the hot path (isochrones/models.py:253-718):
``interp_value(pars, props)``, ``interp_mag(pars, bands)``, ``__call__``, the property shortcuts
(``mass``, ``Teff``, ``logg`` ...), ``param_names``, ``param_index_order``, ``eep_replaces``,
``eep_bounds``, ``model_grid`` / ``bc_grid`` with ``.interp`` = :class:`DFInterpolator` and
``get_limits``.  Evaluation happens in libiso_hip (kernels K3 ``interp_nd`` and K4
``interp_mag``); inputs may be scalars (reference behaviour: returns scalars/1-D arrays),
numpy arrays (broadcast, returns numpy) or CUDA tensors (returns CUDA tensors, nothing leaves
the device).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _cabi, device as dev, grids
from .interp import DFInterpolator


from .interp import HOST_CALL_ROWS


class TableGrid:
    """Holder of one dense table — the slice of the reference's ``Grid`` API
    (isochrones/grid.py:10-144) that the numeric path reads."""

    eep_replaces = None
    bounds = ()

    def __init__(self, interp: DFInterpolator, limits=None, bands=None):
        self.interp = interp
        self._limits = dict(self.bounds)
        self._limits.update(limits or {})
        self.bands = list(bands) if bands is not None else None

    @property
    def columns(self):
        return self.interp.columns

    def get_limits(self, prop):
        if prop not in self._limits:
            col = self.interp.grid[..., self.interp.column_index[prop]]
            self._limits[prop] = (float(np.nanmin(col)), float(np.nanmax(col)))
        return self._limits[prop]


class EvolutionTrackGrid(TableGrid):
    """index (initial_feh, initial_mass, EEP) — reference: isochrones/mist/models.py:164-173"""
    eep_replaces = "age"
    bounds = (("age", (5, 10.13)), ("feh", (-4, 0.5)), ("eep", (0, 1710)), ("mass", (0.1, 300)))

    @property
    def fehs(self):
        return self.interp.index_columns[0]

    @property
    def masses(self):
        return self.interp.index_columns[1]


class IsochroneGrid(TableGrid):
    """index (log10 age, feh, EEP) — reference: isochrones/mist/models.py:88-99"""
    eep_replaces = "mass"
    bounds = (("age", (5, 10.13)), ("feh", (-4, 0.5)), ("eep", (0, 1710)), ("mass", (0.1, 300)))

    @property
    def ages(self):
        return self.interp.index_columns[0]

    @property
    def fehs(self):
        return self.interp.index_columns[1]


class BolometricCorrectionGrid(TableGrid):
    """index (Teff, logg, [Fe/H], Av), one column per band — reference: isochrones/bc.py:9-118"""


class ModelGridInterpolator:
    param_names = None
    eep_replaces = None
    _param_index_order = (1, 2, 0, 3, 4)
    eep_bounds = (0, 1710)
    kind = None

    def __init__(self, model_grid: TableGrid, bc_grid: TableGrid, bands=None, eep_bounds=None):
        self.model_grid = model_grid
        self.bc_grid = bc_grid
        self.bands = list(bands) if bands is not None else list(bc_grid.interp.columns)
        self.param_index_order = list(self._param_index_order)
        if eep_bounds is not None:
            self.eep_bounds = tuple(eep_bounds)
        self._handles = {}
        self._handle_tables = {}
        self._generation = 0        # bumped whenever an iso_ic is destroyed (models bound to it rebuild)
        ci = model_grid.interp.column_index
        missing = [c for c in ("Teff", "logg", "feh", "Mbol") if c not in ci]
        if missing:
            raise ValueError("model table lacks column(s) %s" % missing)
        prior_names = ("age", "dt_deep") if self.eep_replaces == "age" else ("mass", "dm_deep")
        self._cols = [ci[c] for c in ("Teff", "logg", "feh", "Mbol")]
        self._prior_cols = [ci.get(prior_names[0], -1), ci.get(prior_names[1], -1)]
        if -1 in self._prior_cols:
            self._prior_cols = [-1, -1]
        self._astero_cols = [ci.get("nu_max", -1), ci.get("delta_nu", -1)]
        if -1 in self._astero_cols:
            self._astero_cols = [-1, -1]

    # -- limits -----------------------------------------------------------------------------
    minfeh = property(lambda s: s.model_grid.get_limits("feh")[0])
    maxfeh = property(lambda s: s.model_grid.get_limits("feh")[1])
    mineep = property(lambda s: s.model_grid.get_limits("eep")[0])
    maxeep = property(lambda s: s.model_grid.get_limits("eep")[1])
    minage = property(lambda s: s.model_grid.get_limits("age")[0])
    maxage = property(lambda s: s.model_grid.get_limits("age")[1])
    minmass = property(lambda s: s.model_grid.get_limits("mass")[0])
    maxmass = property(lambda s: s.model_grid.get_limits("mass")[1])

    @property
    def fehs(self):
        return self.model_grid.fehs

    @property
    def ages(self):
        if self.eep_replaces != "mass":
            raise AttributeError("Age is not a dimension of this model grid!")
        return self.model_grid.ages

    @property
    def masses(self):
        if self.eep_replaces != "age":
            raise AttributeError("Mass is not a dimension of this model grid!")
        return self.model_grid.masses

    # companion grid of the other parametrisation (reference: EvolutionTrackInterpolator.iso /
    # IsochroneInterpolator.track, models.py:676-709); set by the factories
    _companion_factory = None

    def _companion(self):
        if getattr(self, "_companion_obj", None) is None:
            if self._companion_factory is None:
                raise ValueError("{} has no companion grid type".format(type(self).__name__))
            self._companion_obj = self._companion_factory()
        return self._companion_obj

    # -- device residency -------------------------------------------------------------------
    def handle(self, device=None):
        """iso_ic* for `device`: both tables uploaded once + packed hot-column table built."""
        if device is None:
            device = dev.current_device()
        mg = self.model_grid.interp.handle(device)
        bc = self.bc_grid.interp.handle(device)
        h = self._handles.get(device)
        tables = (self.model_grid.interp._generation, self.bc_grid.interp._generation)
        if h is not None and self._handle_tables.get(device) != tables:
            _cabi.lib().iso_ic_destroy(h)      # a table was rebuilt (add_column): rebind
            self._generation += 1
            h = None
        if h is None:
            ctx = dev.context(device)
            keep0, cols = dev.i32_array(self._cols)          # keep the arrays alive across the call
            keep1, pcols = dev.i32_array(self._prior_cols)
            keep2, acols = dev.i32_array(self._astero_cols)
            h = C.c_void_p()
            _cabi.check(_cabi.lib().iso_ic_create(ctx, mg, bc, self.kind, cols, pcols, acols, C.byref(h)))
            self._handles[device] = h
            self._handle_tables[device] = tables
        return h

    def release(self):
        for h in self._handles.values():
            _cabi.lib().iso_ic_destroy(h)
        self._handles = {}
        self._handle_tables = {}
        self._generation = getattr(self, "_generation", 0) + 1
        for h in getattr(self, "_eep_handles", {}).values():
            _cabi.lib().iso_eep_table_destroy(h)
        self._eep_handles = {}

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # -- hot path ---------------------------------------------------------------------------
    def initialize(self, pars=None):
        """Make the interpolator ready for use (reference: models.py:349-358 triggers the numba compilation with one
        call; here: upload / pack the tables on the current device and run one evaluation)."""
        if pars is None:
            lo_e, hi_e = self.eep_bounds
            mid = 0.5 * (lo_e + hi_e)
            if self.eep_replaces == "age":
                m = self.model_grid.masses
                pars = [float(m[len(m) // 2]), mid, float(self.fehs[len(self.fehs) // 2]), 100.0, 0.1]
            else:
                a = self.model_grid.ages
                pars = [mid, float(a[len(a) // 2]), float(self.fehs[len(self.fehs) // 2]), 100.0, 0.1]
        self.handle()
        return self.interp_mag(pars, self.bands)

    @property
    def name(self):
        return type(self).__name__

    def interp_value(self, pars, props):
        """Interpolate model-table columns ``props`` at ``pars`` (in ``param_names`` order; only
        the first three are used).  reference: isochrones/models.py:390-400"""
        i0, i1, i2 = self.param_index_order[:3]
        p = [pars[i0], pars[i1], pars[i2]]
        p = [float(x) if isinstance(x, (np.floating, np.integer)) else x for x in p]
        return self.model_grid.interp(p, props)

    def _band_cols(self, bands):
        key = tuple(bands)
        cache = self.__dict__.setdefault("_band_col_cache", {})
        if key not in cache:
            cols = self.bc_grid.interp.columns
            cache[key] = np.array([cols.index(b) for b in bands], dtype=np.int32)
        return cache[key]

    def interp_mag_device(self, pars, bands=None, device=None):
        """pars: CUDA float64 tensor [5, N] (SoA) -> (Teff[N], logg[N], feh[N], mags[N, nb]) tensors."""
        bands = self.bands if bands is None else bands
        if device is None:
            device = pars.device.index
        if pars.dim() != 2 or pars.shape[0] != 5 or pars.dtype.itemsize != 8 or not pars.is_contiguous():
            raise ValueError("pars must be a contiguous float64 [5, N] tensor (%s)" % ", ".join(self.param_names))
        n = pars.shape[1]
        nb = len(bands)
        if nb > _cabi.ISO_MAX_BANDS:
            raise ValueError("at most %d bands per call" % _cabi.ISO_MAX_BANDS)
        Teff = dev.empty_f64((n,), device)
        logg = dev.empty_f64((n,), device)
        feh = dev.empty_f64((n,), device)
        mags = dev.empty_f64((n, nb), device)
        if n:
            bc_cols, bcp = dev.i32_array(self._band_cols(bands))
            _cabi.check(_cabi.lib().iso_interp_mag(self.handle(device), dev.ptr(pars), 1, n, n, bcp, nb,
                                                   dev.ptr(Teff), dev.ptr(logg), dev.ptr(feh), dev.ptr(mags),
                                                   dev.stream_ptr(device)))
        return Teff, logg, feh, mags

    def _scalar_interp_mag(self, pars, bands):
        """interp_mag of five plain numbers (the reference's scalar form, models.py:416-431): handle, band columns, one
        buffer [pars | Teff logg feh | mags] and its addresses are kept per thread and band list (revalidated by the
        interpolator's generation number); the C call is answered by the context's resident service wave."""
        for x in pars:
            if not isinstance(x, (float, int, np.floating, np.integer)) or isinstance(x, bool):
                return None
        tls = self.__dict__.get("_scalar_tls")
        if tls is None:
            import threading
            tls = self.__dict__.setdefault("_scalar_tls", threading.local())
        cache = tls.__dict__.get("c")
        gen = (self._generation, self.model_grid.interp._generation, self.bc_grid.interp._generation)
        if cache is None or cache[0] != gen:
            cache = tls.c = (gen, {})
        key = tuple(bands) if bands else ()
        c = cache[1].get(key)
        if c is None:
            nb = len(key)
            if nb > _cabi.ISO_MAX_BANDS:
                return None
            h = self.handle(dev.current_device())
            gen = (self._generation, self.model_grid.interp._generation, self.bc_grid.interp._generation)
            if cache[0] != gen:                      # (handle() rebound the interpolator)
                cache = tls.c = (gen, {})
            buf = np.empty(8 + nb)
            base = buf.ctypes.data
            keep, bcp = dev.i32_array(self._band_cols(list(key)))
            c = cache[1][key] = (h, buf, base, keep, bcp, nb, _cabi.lib().iso_interp_mag_host)
        buf = c[1]
        buf[0], buf[1], buf[2], buf[3], buf[4] = pars
        base = c[2]
        rc = c[6](c[0], base, 1, c[4], c[5], base + 40, base + 48, base + 56, base + 64)
        if rc:
            _cabi.check(rc)
        out = buf[5:].copy()
        return out[0], out[1], out[2], out[3:]

    def interp_mag(self, pars, bands):
        """(Teff, logg, feh, mags) at ``pars`` = the five ``param_names``.
        reference: isochrones/models.py:402-445 (scalar form -> mags.interp_mag, array form ->
        mags.interp_mags)."""
        tp = type(pars)
        if (tp is list or tp is tuple) and len(pars) == 5 and type(pars[0]) in (float, int, np.float64):
            r = self._scalar_interp_mag(pars, bands)
            if r is not None:
                return r
        bands = list(bands) if bands else []
        if dev.is_tensor(pars) and pars.is_cuda:
            p = pars.double()
            if p.dim() == 1:
                p = p[:, None]
            return self.interp_mag_device(p.contiguous(), bands)
        if not isinstance(pars, np.ndarray) and any(dev.is_tensor(x) and x.is_cuda for x in pars):
            import torch
            device = next(x.device.index for x in pars if dev.is_tensor(x) and x.is_cuda)
            xs = torch.broadcast_tensors(*[dev.to_device_f64(x, device) for x in pars])
            p = torch.stack([x.reshape(-1) for x in xs]).contiguous()
            return self.interp_mag_device(p, bands, device)
        device = dev.current_device()
        scalar = False
        nb = len(bands)
        if (nb <= _cabi.ISO_MAX_BANDS and isinstance(pars, (list, tuple)) and len(pars) == 5
                and all(isinstance(x, (float, int, np.floating, np.integer)) for x in pars)):
            # five plain numbers (the reference's scalar form): one buffer [pars | Teff logg feh | mags], one C call
            buf = np.empty(8 + nb)
            buf[:5] = pars
            base = buf.ctypes.data
            key = tuple(bands)
            cached = self._band_ptr_cache.get(key) if hasattr(self, "_band_ptr_cache") else None
            if cached is None:
                if not hasattr(self, "_band_ptr_cache"):
                    self._band_ptr_cache = {}
                cached = self._band_ptr_cache[key] = dev.i32_array(self._band_cols(bands))
            _cabi.check(_cabi.lib().iso_interp_mag_host(self.handle(device), base, 1, cached[1], nb, base + 40, base + 48,
                                                        base + 56, base + 64))
            return buf[5], buf[6], buf[7], buf[8:]
        try:
            arr = np.atleast_1d(pars).astype(float).squeeze()
            if arr.ndim > 1 or arr.shape != (5,):
                raise ValueError
            scalar = True
            p = arr.reshape(5, 1)
        except (TypeError, ValueError):
            if len(pars) != 5:
                raise ValueError("interp_mag needs the five parameters %s" % (self.param_names,))
            b = np.broadcast(*pars)
            p = np.array([np.resize(x, b.shape).astype(float).ravel() for x in pars])
        n = p.shape[1]
        if n <= HOST_CALL_ROWS and len(bands) <= _cabi.ISO_MAX_BANDS:
            rows = np.ascontiguousarray(p.T)
            Teff, logg, feh = np.empty(n), np.empty(n), np.empty(n)
            mags = np.empty((n, len(bands)))
            dp = C.POINTER(C.c_double)
            bc_cols, bcp = dev.i32_array(self._band_cols(bands))
            _cabi.check(_cabi.lib().iso_interp_mag_host(self.handle(device), rows.ctypes.data_as(dp), n, bcp, len(bands),
                                                        Teff.ctypes.data_as(dp), logg.ctypes.data_as(dp),
                                                        feh.ctypes.data_as(dp), mags.ctypes.data_as(dp)))
        else:
            Teff, logg, feh, mags = self.interp_mag_device(dev.to_device_f64(p, device), bands, device)
            Teff, logg, feh, mags = (t.cpu().numpy() for t in (Teff, logg, feh, mags))
        if scalar:
            return Teff[0], logg[0], feh[0], mags[0]
        return Teff, logg, feh, mags

    # property shortcuts (reference: models.py:358-388)
    def _prop(self, prop, *pars):
        return np.squeeze(self.interp_value(pars, [prop]))

    def mass(self, *pars): return self._prop("mass", *pars)
    def initial_mass(self, *pars): return self._prop("initial_mass", *pars)
    def radius(self, *pars): return self._prop("radius", *pars)
    def Teff(self, *pars): return self._prop("Teff", *pars)
    def logg(self, *pars): return self._prop("logg", *pars)
    def feh(self, *pars): return self._prop("feh", *pars)
    def density(self, *pars): return self._prop("density", *pars)
    def nu_max(self, *pars): return self._prop("nu_max", *pars)
    def delta_nu(self, *pars): return self._prop("delta_nu", *pars)

    # -- (mass, age, feh) -> EEP ("next" row f2; reference: models.py:501-542) --------------
    def _eep_handle(self, device):
        if self.eep_replaces != "age":
            raise NotImplementedError("get_eep needs the evolution-track parametrisation (as the reference)")
        h = getattr(self, "_eep_handles", {}).get(device)
        if h is None:
            from .ingest import ragged_age_arrays
            dfi = self.model_grid.interp
            ages, lengths = ragged_age_arrays(dfi, "age")
            self._age_grid, self._array_lengths = ages, lengths
            dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
            fehs, masses, eeps = dfi.index_columns
            h = C.c_void_p()
            _cabi.check(_cabi.lib().iso_eep_table_create(
                dev.context(device), ages.ctypes.data_as(dp), lengths.ctypes.data_as(ip), fehs.ctypes.data_as(dp),
                fehs.size, masses.ctypes.data_as(dp), masses.size, ages.shape[1], float(eeps[0]), C.byref(h)))
            if not hasattr(self, "_eep_handles"):
                self._eep_handles = {}
            self._eep_handles[device] = h
        return h

    def max_eep(self, mass, feh):
        """Last populated EEP shared by the tracks that bracket (mass, feh) (reference: models.py:498,
        mist/eep.py hard-codes the MIST values; here it is read off the table)."""
        if self.eep_replaces != "age":
            raise NotImplementedError("max_eep needs the evolution-track parametrisation")
        self._eep_handle(dev.current_device())
        dfi = self.model_grid.interp
        fehs, masses, eeps = dfi.index_columns
        i = int(np.clip(np.searchsorted(fehs, feh, side="right") - 1, 0, fehs.size - 2))
        j = int(np.clip(np.searchsorted(masses, mass, side="right") - 1, 0, masses.size - 2))
        L = self._array_lengths.reshape(fehs.size, masses.size)[i:i + 2, j:j + 2]
        return float(eeps[int(L.min()) - 1]) if L.min() > 0 else float("nan")

    def mass_age_resid(self, eep, mass, age, feh):
        raise NotImplementedError

    def get_eep_accurate(self, mass, age, feh, eep0=300, resid_tol=0.02, method="nelder-mead", return_object=False,
                         return_nan=False, **kwargs):
        """Nelder-Mead refinement of the EEP against ``mass_age_resid`` (reference: models.py:544-578; a
        host-side scalar loop over the device interpolator, as in the reference)."""
        from scipy.optimize import minimize
        eeps_to_try = [min(self.max_eep(mass, feh) - 20, 600), 100, 200]
        while np.isnan(self.mass_age_resid(eep0, mass, age, feh)):
            try:
                eep0 = eeps_to_try.pop()
            except IndexError:
                if return_nan:
                    return np.nan
                raise ValueError("eep0 gives nan for all initial guesses! {}".format((mass, age, feh)))
        result = minimize(lambda e: self.mass_age_resid(float(np.ravel(e)[0]), mass, age, feh), eep0, method=method,
                          options=kwargs)
        if return_object:
            return result
        if result.success and result.fun < resid_tol ** 2:
            return float(np.ravel(result.x)[0])
        if return_nan:
            return np.nan
        raise RuntimeError("EEP minimization not successful: {}".format((mass, age, feh)))

    def get_eep(self, mass, age, feh, accurate=False, **kwargs):
        """EEP of a star of given (mass, log10 age, feh): bilinear blend over the four neighbouring
        tracks of the first EEP whose age exceeds ``age`` (reference ``interp_eep(s)``).  Scalars
        -> float, arrays -> numpy, CUDA tensors -> CUDA tensor.  ``accurate=True`` refines a scalar
        result with :meth:`get_eep_accurate` starting from the fast estimate, as the reference."""
        if accurate:
            eep0 = self.get_eep(mass, age, feh)
            if np.ndim(eep0) == 0:
                return self.get_eep_accurate(mass, age, feh, eep0=eep0 if np.isfinite(eep0) else 300, **kwargs)
            b = np.broadcast(mass, age, feh)           # arrays: one refinement per star, as the reference
            m, a, f = [np.resize(x, b.shape).astype(float).ravel() for x in (mass, age, feh)]
            return np.array([self.get_eep_accurate(mi, ai, fi, eep0=e if np.isfinite(e) else 300, **kwargs)
                             for mi, ai, fi, e in zip(m, a, f, np.ravel(eep0))])
        if type(mass) in (float, int, np.float64) and type(age) in (float, int, np.float64) and type(feh) in (float, int, np.float64):
            # the call form of the reference's notebooks: three plain numbers, one C call (the context's resident service
            # wave); buffers and the table handle kept per thread
            tls = self.__dict__.get("_eep_tls")
            if tls is None:
                import threading
                tls = self.__dict__.setdefault("_eep_tls", threading.local())
            c = tls.__dict__.get("c")
            if c is None or c[0] != self._generation or c[1] != self.model_grid.interp._generation:
                h = self._eep_handle(dev.current_device())
                buf = (C.c_double * 4)()
                base = C.addressof(buf)
                c = tls.c = (self._generation, self.model_grid.interp._generation, h, buf, base, _cabi.lib().iso_interp_eep_host)
            buf = c[3]
            buf[0], buf[1], buf[2] = age, feh, mass
            base = c[4]
            rc = c[5](c[2], base, base + 8, base + 16, 1, base + 24)
            if rc:
                _cabi.check(rc)
            return buf[3]
        args = [mass, age, feh]
        if any(dev.is_tensor(a) and a.is_cuda for a in args):
            import torch
            device = next(a.device.index for a in args if dev.is_tensor(a) and a.is_cuda)
            m, a, f = [t.reshape(-1).contiguous() for t in
                       torch.broadcast_tensors(*[dev.to_device_f64(x, device) for x in args])]
            out = dev.empty_f64((m.numel(),), device)
            _cabi.check(_cabi.lib().iso_interp_eep(self._eep_handle(device), dev.ptr(a), dev.ptr(f), dev.ptr(m),
                                                   m.numel(), dev.ptr(out), dev.stream_ptr(device)))
            return out
        device = dev.current_device()
        scalar = all(isinstance(x, (float, int, np.floating, np.integer)) for x in args)
        if scalar:                                 # the call form of the reference's notebooks: no array machinery
            one = C.c_double * 1
            out1 = one()
            _cabi.check(_cabi.lib().iso_interp_eep_host(self._eep_handle(device), one(age), one(feh), one(mass), 1, out1))
            return out1[0]
        b = np.broadcast(*args)
        hm, ha, hf = [np.ascontiguousarray(np.atleast_1d(np.resize(x, b.shape)).astype(float).ravel()) for x in args]
        if hm.size <= HOST_CALL_ROWS:
            res = np.empty(hm.size)
            dp = C.POINTER(C.c_double)
            _cabi.check(_cabi.lib().iso_interp_eep_host(self._eep_handle(device), ha.ctypes.data_as(dp), hf.ctypes.data_as(dp),
                                                        hm.ctypes.data_as(dp), hm.size, res.ctypes.data_as(dp)))
            return float(res[0]) if scalar else res
        m, a, f = [dev.to_device_f64(x, device) for x in (hm, ha, hf)]
        out = dev.empty_f64((m.numel(),), device)
        _cabi.check(_cabi.lib().iso_interp_eep(self._eep_handle(device), dev.ptr(a), dev.ptr(f), dev.ptr(m),
                                               m.numel(), dev.ptr(out), dev.stream_ptr(device)))
        res = out.cpu().numpy()
        return float(res[0]) if scalar else res

    def generate(self, mass, age, feh, props="all", bands=None, eeps=None, return_df=True, return_dict=False,
                 distance=10, AV=0, all_As=False, **kwargs):
        """Model columns + magnitudes of stars given (mass, log10 age, feh): a DataFrame (default) or, with
        ``return_dict``, a dict of arrays; ``all_As`` adds the per-band extinctions ``A_<band>``; other keywords
        (``accurate=True`` ...) go to :meth:`get_eep` (reference: models.py:580-631)."""
        import pandas as pd
        mass, age, feh, distance, AV = [np.atleast_1d(a).astype(float).ravel()
                                        for a in np.broadcast_arrays(mass, age, feh, distance, AV)]
        bands = self.bands if bands is None else list(bands)
        if eeps is None:
            eeps = np.atleast_1d(self.get_eep(mass, age, feh, **kwargs))
        cols = list(self.model_grid.interp.columns) if (isinstance(props, str) and props == "all") else list(props)
        values = np.atleast_2d(self.interp_value([mass, eeps, feh], cols))
        out = pd.DataFrame(values, columns=cols)
        if bands:
            _, _, _, mags = self.interp_mag([mass, eeps, feh, distance, AV], bands)
            for j, b in enumerate(bands):
                out["{}_mag".format(b)] = np.atleast_2d(mags)[:, j]
        out["distance"] = distance
        out["AV"] = AV
        out["initial_feh"] = feh
        out["requested_age"] = age
        if all_As and bands:
            _, _, _, true_mags = self.interp_mag([mass, eeps, feh, distance, np.zeros_like(AV)], bands)
            for j, b in enumerate(bands):
                out["A_{}".format(b)] = out["{}_mag".format(b)].values - np.atleast_2d(true_mags)[:, j]
        if return_dict or not return_df:
            return {c: out[c].values for c in out.columns}
        return out

    def generate_binary(self, mass_A, mass_B, age, feh, bands=None, **kwargs):
        """Two coeval stars: per-component columns suffixed _0 / _1 plus the combined magnitudes
        (reference: models.py:633-661)."""
        import pandas as pd
        bands = self.bands if bands is None else list(bands)
        mass_A, mass_B = np.broadcast_arrays(mass_A, mass_B)
        a = self.generate(mass_A, age, feh, bands=bands, **kwargs)
        b = self.generate(mass_B, age, feh, bands=bands, **kwargs)
        out = pd.concat([a.rename(columns={c: c + "_0" for c in a.columns}),
                         b.rename(columns={c: c + "_1" for c in b.columns})], axis=1)
        for band in bands:
            m0 = a["{}_mag".format(band)].values
            m1 = np.where(np.isnan(b["{}_mag".format(band)].values), np.inf, b["{}_mag".format(band)].values)
            out["{}_mag".format(band)] = -2.5 * np.log10(10 ** (-0.4 * m0) + 10 ** (-0.4 * m1))
        return out

    def isochrone(self, age, feh=0.0, eep_range=None, distance=10.0, AV=0.0, dropna=True):
        """All columns + magnitudes along one isochrone (reference: models.py:484-493)."""
        if self.eep_replaces != "mass":
            raise NotImplementedError("isochrone() needs the isochrone parametrisation")
        if eep_range is None:
            eep_range = self.model_grid.get_limits("eep")
        eeps = np.arange(*eep_range, dtype=float)
        df = self(eeps, age, feh, distance=distance, AV=AV)
        return df.dropna() if dropna else df

    def model_value(self, mass, age, feh, props):
        props = [props] if isinstance(props, str) else list(props)
        eep = self.get_eep(mass, age, feh)
        return np.squeeze(self.interp_value([mass, eep, feh], props))

    def model_mag(self, mass, age, feh, distance=10.0, AV=0.0, bands=None):
        bands = self.bands if bands is None else list(bands)
        eep = self.get_eep(mass, age, feh)
        return np.squeeze(self.interp_mag([mass, eep, feh, distance, AV], bands)[3])

    def __call__(self, p1, p2, p3, distance=10.0, AV=0.0):
        """All model columns + every band's magnitude, as a DataFrame
        (reference: isochrones/models.py:471-482)."""
        import pandas as pd
        p1, p2, p3, dist, AV = [np.atleast_1d(a).astype(float).ravel()
                                for a in np.broadcast_arrays(p1, p2, p3, distance, AV)]
        pars = [p1, p2, p3, dist, AV]
        prop_cols = list(self.model_grid.interp.columns)
        props = self.interp_value(pars, prop_cols)
        _, _, _, mags = self.interp_mag(pars, self.bands)
        cols = prop_cols + ["{}_mag".format(b) for b in self.bands]
        values = np.concatenate([np.atleast_2d(props), np.atleast_2d(mags)], axis=1)
        return pd.DataFrame(values, columns=cols)


class EvolutionTrackInterpolator(ModelGridInterpolator):
    """(mass, eep, feh, distance, AV) over a (feh, mass, eep) table — reference: models.py:664-688"""
    param_names = ("mass", "eep", "feh", "distance", "AV")
    eep_replaces = "age"
    _param_index_order = (2, 0, 1, 3, 4)
    kind = _cabi.KIND_TRACK

    @property
    def iso(self):
        return self._companion()

    def mass_age_resid(self, eep, mass, age, feh):
        return (age - self.interp_value([mass, eep, feh], ["age"])) ** 2


class IsochroneInterpolator(ModelGridInterpolator):
    """(eep, age, feh, distance, AV) over an (age, feh, eep) table — reference: models.py:691-718"""
    param_names = ("eep", "age", "feh", "distance", "AV")
    eep_replaces = "mass"
    _param_index_order = (1, 2, 0, 3, 4)
    kind = _cabi.KIND_ISO

    @property
    def track(self):
        return self._companion()

    def mass_age_resid(self, eep, mass, age, feh):
        return (mass - self.interp_value([eep, age, feh], ["initial_mass"])) ** 2

    def max_eep(self, mass, feh):
        return self.track.max_eep(mass, feh)

    def get_eep(self, mass, age, feh, accurate=False, **kwargs):
        """Fast estimate from the companion track grid (reference: IsochroneInterpolator delegates to
        ``self.track``), optionally refined against this grid's own initial_mass column."""
        eep0 = self.track.get_eep(mass, age, feh)
        if not accurate:
            return eep0
        return self.get_eep_accurate(mass, age, feh, eep0=eep0 if np.isfinite(eep0) else 300, **kwargs)

    def generate(self, *args, **kwargs):
        return self.track.generate(*args, **kwargs)


# ------------------------------------------------------------------------------------------
# factories over the synthetic MIST-shaped tables (real MIST data needs the network)
# ------------------------------------------------------------------------------------------

def _bc(bands, bc_axes=None):
    g, ax, cols = grids.synthetic_bc_grid(tuple(bands), bc_axes)
    return BolometricCorrectionGrid(DFInterpolator.from_arrays(g, ax, cols, ["Teff", "logg", "[Fe/H]", "Av"]),
                                    bands=bands)


def synthetic_track(bands=grids.DEFAULT_BANDS, fehs=None, masses=None, eeps=None, bc_axes=None,
                    limits=None, eep_bounds=None, ragged=True):
    """MIST_EvolutionTrack-shaped interpolator over the synthetic tables."""
    g, ax, cols = grids.synthetic_track_grid(fehs, masses, eeps, ragged=ragged)
    mg = EvolutionTrackGrid(DFInterpolator.from_arrays(g, ax, cols, ["initial_feh", "initial_mass", "EEP"]),
                            limits=limits)
    ic = EvolutionTrackInterpolator(mg, _bc(bands, bc_axes), bands=bands, eep_bounds=eep_bounds)
    ic.data_source = "synthetic"
    if fehs is None and masses is None and eeps is None:          # full-size tables: the companion is well defined
        ic._companion_factory = lambda: synthetic_isochrone(bands, bc_axes=bc_axes)
    return ic


def synthetic_isochrone(bands=grids.DEFAULT_BANDS, ages=None, fehs=None, eeps=None, bc_axes=None,
                        limits=None, eep_bounds=None, ragged=True):
    """MIST_Isochrone-shaped interpolator over the synthetic tables."""
    g, ax, cols = grids.synthetic_iso_grid(ages, fehs, eeps, ragged=ragged)
    mg = IsochroneGrid(DFInterpolator.from_arrays(g, ax, cols, ["log10_isochrone_age_yr", "feh", "EEP"]),
                       limits=limits)
    ic = IsochroneInterpolator(mg, _bc(bands, bc_axes), bands=bands, eep_bounds=eep_bounds)
    ic.data_source = "synthetic"
    if ages is None and fehs is None and eeps is None:
        ic._companion_factory = lambda: synthetic_track(bands, bc_axes=bc_axes)
    return ic


def get_ichrone(models="mist", bands=None, default=False, tracks=False, basic=False, **kwargs):
    """The reference's factory (isochrones/isochrone.py:48-78).  An interpolator object passed as ``models`` is returned
    as it is.

    ``"mist"``: the MIST grids from the reference's data directory, ``$ISOCHRONES`` (default ``~/.isochrones``): its
    ``full_grid*.npz`` caches and BC frames, see :mod:`isochrones_amd.mist`.  When that directory holds no MIST caches
    this build can read, a ``UserWarning`` says so (naming what is missing) and the MIST-shaped *synthetic* tables of
    :mod:`isochrones_amd.grids` are returned: same axes, columns and ragged structure, invented physics - good for
    exercising the path, not for astrophysics.  ``"synthetic"`` asks for those tables by name (no warning).
    ``default`` / ``basic`` select variants of the reference's grids that coincide here."""
    if isinstance(models, ModelGridInterpolator):
        return models
    if models not in ("mist", "synthetic"):
        raise ValueError("Unknown stellar models: {}".format(models))
    bands = grids.DEFAULT_BANDS if not bands else tuple(bands)
    if models == "mist":
        from . import mist
        grid_kw = {k: kwargs[k] for k in ("version", "vvcrit", "kind") if k in kwargs}
        try:
            return mist.load_mist(bands, tracks=tracks, **grid_kw)
        except mist.MistDataNotFound as e:
            import warnings
            # synthetic tables only when NOTHING is there; a cache that exists but cannot be used is the user's to fix
            if not mist.nothing_there(tracks=tracks, **grid_kw):
                raise
            warnings.warn("get_ichrone('mist'): no MIST tables under %s (%s) - using synthetic MIST-shaped tables "
                          "(isochrones_amd.grids: same axes and columns, invented physics)" % (mist.data_root(), e),
                          UserWarning, stacklevel=2)
        kwargs = {k: v for k, v in kwargs.items() if k not in grid_kw}
    ic = synthetic_track(bands, **kwargs) if tracks else synthetic_isochrone(bands, **kwargs)
    return ic
