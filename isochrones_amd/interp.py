"""``DFInterpolator`` — N-D multilinear interpolation of the columns of a dense table, evaluated
by the HIP kernels of libiso_hip (K3 ``interp_nd``).

Mirrors the surface of the reference class (isochrones/interp.py:571-698): attributes ``grid``,
``index_columns``, ``index_names``, ``columns``, ``column_index``, ``n_columns``, ``ndim``;
``__call__(p, cols="all")`` returns ``[k]`` for all-scalar ``p`` and ``[N, k]`` for array ``p``
(with numpy broadcasting), NaN for NaN / out-of-grid queries.  In addition ``p`` may hold CUDA
tensors, in which case the result stays on the device.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _cabi, device as dev


HOST_CALL_ROWS = 4096     # host inputs up to this many rows use the *_host entry points (one launch + one sync)
TABLE_EPOCH = [0]         # bumped whenever ANY interpolator frees its device tables (release / add_column): per-call caches
                          # that skip ic.handle() (StarModel._scalar_call, the mailbox accessors) compare this integer


def _is_scalar(v):
    return isinstance(v, (float, int)) and not isinstance(v, bool)


class DFInterpolator:
    def __init__(self, df=None, filename=None, recalc=False, is_full=False, *, grid=None,
                 index_columns=None, columns=None, index_names=None, device_grid=None):
        self.filename = filename
        self.is_full = is_full
        # a table that ARRIVED on the device (broadcast_interpolator over RCCL): the device tensor is handed to the library as
        # it is (iso_table_create_from_device) and the host copy behind `grid` is only made if somebody asks for it
        self._device_grid = device_grid
        self._grid = None
        self._handles = {}          # device index -> iso_table*
        self._generation = 0        # bumped whenever device tables are freed: dependants compare generations, not
                                    # pointer values (a new table often lands on the address of the freed one)
        if df is not None:
            self.columns = list(df.columns)
            self.index_columns = tuple(np.array(l, dtype=float) for l in df.index.levels)
            self.index_names = list(df.index.names)
            self.grid = self._make_grid(df, recalc=recalc)
        else:
            self.columns = list(columns)
            self.index_columns = tuple(np.ascontiguousarray(a, dtype=float) for a in index_columns)
            self.index_names = list(index_names) if index_names is not None else [
                "x%d" % i for i in range(len(self.index_columns))]
            if device_grid is None:
                self.grid = np.ascontiguousarray(grid, dtype=float)
            elif (not device_grid.is_cuda or device_grid.dtype.itemsize != 8 or not device_grid.dtype.is_floating_point
                  or not device_grid.is_contiguous()):
                raise ValueError("device_grid must be a contiguous float64 CUDA tensor")
        self.n_columns = len(self.columns)
        self.ndim = len(self.index_columns)
        if self.ndim not in (2, 3, 4):
            raise ValueError("DFInterpolator supports 2-, 3- and 4-dimensional tables")
        if self.grid_shape != tuple(len(a) for a in self.index_columns) + (self.n_columns,):
            raise ValueError("grid shape %s does not match axes/columns" % (self.grid_shape,))
        for a in self.index_columns:
            if a.size < 2 or not np.all(np.diff(a) > 0):
                raise ValueError("every index level needs >= 2 strictly increasing values")
        self.column_index = {c: i for i, c in enumerate(self.columns)}

    @classmethod
    def from_arrays(cls, grid, index_columns, columns, index_names=None):
        return cls(grid=grid, index_columns=index_columns, columns=columns, index_names=index_names)

    @classmethod
    def from_device(cls, device_grid, index_columns, columns, index_names=None):
        """A table whose values are a float64 CUDA tensor [n0, .., n_columns] (axes on the host)."""
        return cls(device_grid=device_grid, index_columns=index_columns, columns=columns, index_names=index_names)

    @property
    def grid(self):
        """The dense table as a C-contiguous numpy array [n0, .., n_columns] (reference: DFInterpolator.grid); for a table
        that arrived on the device the host copy is made on first use."""
        if self._grid is None and self._device_grid is not None:
            self._grid = self._device_grid.cpu().numpy()
        return self._grid

    @grid.setter
    def grid(self, value):
        self._grid = value

    @property
    def grid_shape(self):
        return tuple(self._grid.shape) if self._grid is not None else tuple(self._device_grid.shape)

    # -- table construction (reference: _make_grid, interp.py:590-614) ---------------------
    def _make_grid(self, df, recalc=False):
        if self.filename is not None and os.path.exists(self.filename) and not recalc:
            d = np.load(self.filename)
            if list(d["columns"]) != list(self.columns):
                raise ValueError("DataFrame columns do not match columns loaded from full grid!")
            return np.ascontiguousarray(d["grid"], dtype=float)
        shape = [len(l) for l in df.index.levels] + [len(df.columns)]
        values = np.asarray(df.values, dtype=float)
        if self.is_full:
            grid = values.reshape(shape)
        else:
            # scatter the (possibly ragged) rows into a NaN-initialised product grid
            grid = np.full(shape, np.nan)
            codes = [np.asarray(c) for c in df.index.codes]
            grid[tuple(codes)] = values
        grid = np.ascontiguousarray(grid)
        if self.filename is not None:
            np.savez(self.filename, grid=grid, columns=self.columns)
        return grid

    def add_column(self, values, name):
        """Append a column (reference: interp.py:616-623); device copies are rebuilt lazily."""
        newgrid = np.empty(self.grid.shape[:-1] + (self.n_columns + 1,))
        newgrid[..., :-1] = self.grid
        newgrid[..., -1] = values
        self.column_index[name] = self.n_columns
        self.n_columns += 1
        self.columns = self.columns + [name]
        self.grid = newgrid
        self._device_grid = None
        self.release()

    # -- device residency -------------------------------------------------------------------
    def handle(self, device=None):
        """iso_table* for `device` (uploaded once, then resident in HBM)."""
        if device is None:
            device = dev.current_device()
        h = self._handles.get(device)
        if h is None:
            ctx = dev.context(device)
            shape = (C.c_int64 * (self.ndim + 1))(*self.grid_shape)
            dp = C.POINTER(C.c_double)
            axes = (dp * self.ndim)(*[a.ctypes.data_as(dp) for a in self.index_columns])
            h = C.c_void_p()
            dg = self._device_grid
            if dg is not None and dg.device.index == device and hasattr(_cabi.lib(), "iso_table_create_from_device"):
                # the values are on this device already: one device-to-device copy instead of a download and an upload
                _cabi.check(_cabi.lib().iso_table_create_from_device(ctx, self.ndim, shape, dev.ptr(dg), axes, C.byref(h)))
                if self._grid is not None:
                    self._device_grid = None            # (a host copy exists: the tensor is not needed any more)
            else:
                _cabi.check(_cabi.lib().iso_table_create(ctx, self.ndim, shape, self.grid.ctypes.data_as(dp),
                                                         axes, C.byref(h)))
            self._handles[device] = h
        return h

    def release(self):
        for h in self._handles.values():
            _cabi.lib().iso_table_destroy(h)
        self._handles = {}
        self._generation += 1
        TABLE_EPOCH[0] += 1

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # -- evaluation -------------------------------------------------------------------------
    def _icols(self, cols):
        if isinstance(cols, str) and cols == "all":
            return np.arange(self.n_columns, dtype=np.int32)
        return np.array([self.column_index[c] for c in cols], dtype=np.int32)

    def interp_device(self, xs, icols, device=None):
        """xs: ndim float64 CUDA tensors of equal length N -> CUDA tensor [N, k]."""
        if len(xs) != self.ndim:
            raise ValueError("need %d coordinate tensors, got %d" % (self.ndim, len(xs)))
        if device is None:
            device = xs[0].device.index
        n = xs[0].numel()
        if any(x.numel() != n or x.dtype.itemsize != 8 or not x.is_contiguous() for x in xs):
            raise ValueError("coordinates must be contiguous float64 tensors of equal length")
        icols = np.ascontiguousarray(icols, dtype=np.int32)
        k = icols.size
        out = dev.empty_f64((n, k), device)
        if n == 0 or k == 0:
            return out
        if k > _cabi.ISO_MAX_COLS:
            parts = [self.interp_device(xs, icols[i:i + _cabi.ISO_MAX_COLS], device)
                     for i in range(0, k, _cabi.ISO_MAX_COLS)]
            import torch
            return torch.cat(parts, dim=1)
        xp = (C.c_void_p * self.ndim)(*[x.data_ptr() for x in xs])
        _cabi.check(_cabi.lib().iso_interp(self.handle(device), xp, n, icols.ctypes.data_as(C.POINTER(C.c_int32)),
                                           k, dev.ptr(out), dev.stream_ptr(device)))
        return out

    def _scalar_interp(self, p, cols):
        """One point given as plain numbers (the call form of the reference's notebooks and of optimisers that walk the
        table, interp.py:631-660): everything a call needs besides the numbers - the table handle, the column numbers, a
        coordinate buffer, an output buffer and their addresses - is kept per thread and per column list and revalidated by
        one integer comparison, so that the wrapper adds about a microsecond to the C call (which the context's resident
        service wave answers without a launch)."""
        tls = self.__dict__.get("_scalar_tls")
        if tls is None:
            import threading
            tls = self.__dict__.setdefault("_scalar_tls", threading.local())
        cache = tls.__dict__.get("c")
        if cache is None or cache[0] != self._generation:
            cache = tls.c = (self._generation, {})
        key = cols if type(cols) is str else tuple(cols)
        c = cache[1].get(key)
        if c is None:
            icols = self._icols(cols)
            if icols.size == 0 or icols.size > _cabi.ISO_MAX_COLS:
                return None
            xbuf = (C.c_double * self.ndim)()
            out = np.empty(icols.size)
            c = cache[1][key] = (self.handle(dev.current_device()), xbuf, C.addressof(xbuf), icols, icols.ctypes.data, icols.size, out,
                                 out.ctypes.data, _cabi.lib().iso_interp_host)
        xbuf = c[1]
        for d in range(self.ndim):
            xbuf[d] = p[d]
        rc = c[8](c[0], c[2], 1, c[4], c[5], c[7])
        if rc:
            _cabi.check(rc)
        return c[6].copy()

    def __call__(self, p, cols="all"):
        tp = type(p)
        if (tp is list or tp is tuple) and len(p) == self.ndim:
            t0 = type(p[0])
            if (t0 is float or t0 is int) and all(type(x) is float or type(x) is int for x in p):
                r = self._scalar_interp(p, cols)
                if r is not None:
                    return r
        icols = self._icols(cols)
        p = list(p)[: self.ndim] if len(p) > self.ndim else list(p)
        if len(p) != self.ndim:
            raise ValueError("expected %d coordinates, got %d" % (self.ndim, len(p)))
        if any(dev.is_tensor(x) and x.is_cuda for x in p):
            import torch
            device = next(x.device.index for x in p if dev.is_tensor(x) and x.is_cuda)
            xs = torch.broadcast_tensors(*[dev.to_device_f64(x, device) for x in p])
            xs = [x.reshape(-1).contiguous() for x in xs]
            return self.interp_device(xs, icols, device)
        device = dev.current_device()
        scalar = all(_is_scalar(x) for x in p)
        if scalar:
            rows = np.array([[float(x) for x in p]])
        else:
            b = np.broadcast(*p)
            rows = np.column_stack([np.atleast_1d(np.resize(x, b.shape)).astype(float).ravel() for x in p])
        n = rows.shape[0]
        if n <= HOST_CALL_ROWS and icols.size <= _cabi.ISO_MAX_COLS:
            # scalar calls / small batches: host arrays through the context's pinned staging buffer
            rows = np.ascontiguousarray(rows)
            out = np.empty((n, icols.size))
            dp = C.POINTER(C.c_double)
            _cabi.check(_cabi.lib().iso_interp_host(self.handle(device), rows.ctypes.data_as(dp), n,
                                                    icols.ctypes.data_as(C.POINTER(C.c_int32)), icols.size,
                                                    out.ctypes.data_as(dp)))
            return out[0] if scalar else out
        xs = [dev.to_device_f64(np.ascontiguousarray(rows[:, d]), device) for d in range(self.ndim)]
        return self.interp_device(xs, icols, device).cpu().numpy()
