"""TEST INFRASTRUCTURE — ctypes wrapper around oracle/libiso_oracle.so (the C restatement of the
reference's hot path, oracle/iso_oracle.c).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product (isochrones_amd/) never does."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from isochrones_amd import _cabi

_HERE = os.path.dirname(os.path.abspath(__file__))
# ISO_ORACLE_LIB: load another build of the same sources (the sanitizer build of tests/test_oracle_sanitizers.py)
_LIBPATH = os.environ.get("ISO_ORACLE_LIB") or os.path.join(_HERE, "libiso_oracle.so")
_lib = None

ORC_MAX_DIM = 4


class _Table(C.Structure):
    _fields_ = [
        ("ndim", C.c_int),
        ("shape", C.c_int64 * (ORC_MAX_DIM + 1)),
        ("grid", C.POINTER(C.c_double)),
        ("axes", C.POINTER(C.c_double) * ORC_MAX_DIM),
    ]


class _IC(C.Structure):
    _fields_ = [
        ("model", _Table),
        ("bc", _Table),
        ("kind", C.c_int),
        ("i_Teff", C.c_int32), ("i_logg", C.c_int32), ("i_feh", C.c_int32), ("i_Mbol", C.c_int32),
        ("i_prior_val", C.c_int32), ("i_prior_deriv", C.c_int32),
        ("i_numax", C.c_int32), ("i_dnu", C.c_int32),
    ]


def build(force=False):
    src = os.path.join(_HERE, "iso_oracle.c")
    if force or not os.path.exists(_LIBPATH) or os.path.getmtime(_LIBPATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", os.path.basename(_LIBPATH)])
    return _LIBPATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIBPATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.orc_interp.argtypes = [C.POINTER(_Table), C.POINTER(dp), C.c_int64, ip, C.c_int, dp, C.c_int]
        L.orc_interp.restype = None
        L.orc_interp_mag.argtypes = [C.POINTER(_IC), dp, C.c_int64, C.c_int64, C.c_int64, ip, C.c_int,
                                     dp, dp, dp, dp, C.c_int]
        L.orc_interp_mag.restype = None
        L.orc_lnpost.argtypes = [C.POINTER(_IC), C.POINTER(_cabi.IsoModelDesc), dp, C.c_int64, C.c_int64,
                                 C.c_int64, dp, dp, dp, C.c_int]
        L.orc_lnpost.restype = None
        L.orc_unit_cube.argtypes = [C.POINTER(_cabi.IsoModelDesc), C.c_int, dp, C.c_int64, C.c_int64, C.c_int64]
        L.orc_unit_cube.restype = None
        L.orc_max_threads.restype = C.c_int
        i64p = C.POINTER(C.c_int64)
        L.orc_interp_eep.argtypes = [dp, dp, dp, C.c_int64, dp, C.c_int64, dp, C.c_int64, dp, i64p, C.c_int64, dp]
        L.orc_interp_eep.restype = None
        L.orc_tree_lnpost.argtypes = [C.POINTER(_IC), C.POINTER(_cabi.IsoTreeDesc), dp, C.c_int64, C.c_int64, C.c_int64,
                                      dp, dp, dp, C.c_int]
        L.orc_tree_lnpost.restype = None
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class OracleTable:
    """Dense N-D table + axes (the arrays a reference DFInterpolator holds)."""

    def __init__(self, grid, axes):
        self.grid = np.ascontiguousarray(grid, dtype=np.float64)
        self.axes = [np.ascontiguousarray(a, dtype=np.float64) for a in axes]
        self.ndim = len(self.axes)
        assert self.grid.ndim == self.ndim + 1
        t = _Table()
        t.ndim = self.ndim
        for d, n in enumerate(self.grid.shape):
            t.shape[d] = n
        t.grid = _dp(self.grid)
        for d, a in enumerate(self.axes):
            assert a.size == self.grid.shape[d]
            t.axes[d] = _dp(a)
        self.c = t

    def interp(self, xs, icols, nthreads=1):
        xs = [np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64) for x in xs]
        n = xs[0].size
        icols = np.ascontiguousarray(icols, dtype=np.int32)
        out = np.empty((n, icols.size))
        xp = (C.POINTER(C.c_double) * self.ndim)(*[_dp(x) for x in xs])
        lib().orc_interp(C.byref(self.c), xp, n, _ip(icols), icols.size, _dp(out), nthreads)
        return out


class OracleIC:
    def __init__(self, kind, model: OracleTable, bc: OracleTable, cols, prior_cols=(-1, -1),
                 astero_cols=(-1, -1)):
        self.model, self.bc, self.kind = model, bc, kind
        ic = _IC()
        ic.model = model.c
        ic.bc = bc.c
        ic.kind = kind
        ic.i_Teff, ic.i_logg, ic.i_feh, ic.i_Mbol = [int(c) for c in cols]
        ic.i_prior_val, ic.i_prior_deriv = [int(c) for c in prior_cols]
        ic.i_numax, ic.i_dnu = [int(c) for c in astero_cols]
        self.c = ic

    def interp_mag(self, pars, bc_cols, nthreads=1):
        """pars [5, N] (SoA) -> Teff[N], logg[N], feh[N], mags[N, nb]"""
        pars = np.ascontiguousarray(pars, dtype=np.float64)
        assert pars.shape[0] == 5
        n = pars.shape[1]
        bc_cols = np.ascontiguousarray(bc_cols, dtype=np.int32)
        T, g, f = np.empty(n), np.empty(n), np.empty(n)
        mags = np.empty((n, bc_cols.size))
        lib().orc_interp_mag(C.byref(self.c), _dp(pars), 1, n, n, _ip(bc_cols), bc_cols.size,
                             _dp(T), _dp(g), _dp(f), _dp(mags), nthreads)
        return T, g, f, mags

    def lnpost(self, desc: _cabi.IsoModelDesc, pars, nthreads=1, parts=True):
        """pars [n_params, N] (SoA) -> (lnpost, lnprior, lnlike) each [N]"""
        pars = np.ascontiguousarray(pars, dtype=np.float64)
        assert pars.shape[0] == desc.n_stars + 4
        n = pars.shape[1]
        post = np.empty(n)
        if parts:
            prior, like = np.empty(n), np.empty(n)
            lib().orc_lnpost(C.byref(self.c), C.byref(desc), _dp(pars), 1, n, n, _dp(post), _dp(prior),
                             _dp(like), nthreads)
            return post, prior, like
        lib().orc_lnpost(C.byref(self.c), C.byref(desc), _dp(pars), 1, n, n, _dp(post), None, None, nthreads)
        return post


def tree_lnpost(oic, desc, pars, nthreads=1):
    """Generic (observation-tree) model: pars [n_params, N] (SoA) -> (lnpost, lnprior, lnlike)."""
    pars = np.ascontiguousarray(pars, dtype=np.float64)
    n = pars.shape[1]
    post, prior, like = np.empty(n), np.empty(n), np.empty(n)
    lib().orc_tree_lnpost(C.byref(oic.c), C.byref(desc), _dp(pars), 1, n, n, _dp(post), _dp(prior), _dp(like), nthreads)
    return post, prior, like


def unit_cube(desc, kind, cube):
    """cube [N, n_params] row-major, transformed in place and returned."""
    cube = np.ascontiguousarray(cube, dtype=np.float64)
    n, npar = cube.shape
    lib().orc_unit_cube(C.byref(desc), kind, _dp(cube), npar, 1, n)
    return cube


def max_threads():
    return lib().orc_max_threads()


def interp_eep(age, feh, mass, fehs, masses, age_grid, lengths):
    """reference interp_eeps(xs=age, x0s=feh, x1s=mass, ...): EEP on the ragged age arrays."""
    age, feh, mass = (np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64) for a in (age, feh, mass))
    fehs = np.ascontiguousarray(fehs, dtype=np.float64)
    masses = np.ascontiguousarray(masses, dtype=np.float64)
    age_grid = np.ascontiguousarray(age_grid, dtype=np.float64)
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    assert age_grid.shape[0] == fehs.size * masses.size and lengths.size == age_grid.shape[0]
    out = np.empty(age.size)
    lib().orc_interp_eep(_dp(age), _dp(feh), _dp(mass), age.size, _dp(fehs), fehs.size, _dp(masses), masses.size,
                         _dp(age_grid), lengths.ctypes.data_as(C.POINTER(C.c_int64)), age_grid.shape[1], _dp(out))
    return out
