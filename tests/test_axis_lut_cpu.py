"""Bucket tables of the LDS-staged axes (csrc/fast/axis_lut.h + lut_start() in csrc/fast/brackets.h), host logic only.

The fused kernels do not bisect an axis from scratch: a byte table indexed by the exponent / leading mantissa bits of
x (+ a shift) gives a node at or below x and a window the bracket lies in.  Whatever table the planner picks, the
integer that comes out must be the reference's (interp.py:10-35 searchsorted + :116-123 find_indices):
#{a_j <= x} - 1 clamped to [0, n - 2].  `iso_axis_bracket_host` runs the planner, the table fill and the statements
of the device bracket on the host; checked here against numpy.searchsorted on the MIST axes, on random axes of every
kind (log-spaced, uniform, clustered, negative, tiny / huge magnitudes), on every node, on the neighbours of every
node, and under every byte budget.
"""
import ctypes as C

import numpy as np
import pytest

from isochrones_amd import _cabi, grids


def _brackets(axes, which, x, budget):
    lib = C.CDLL(_cabi.library_path())
    f = lib.iso_axis_bracket_host
    f.restype = C.c_int
    axes = [np.ascontiguousarray(a, dtype=np.float64) for a in axes]
    ptrs = (C.c_void_p * len(axes))(*[a.ctypes.data for a in axes])
    ns = np.array([len(a) for a in axes], dtype=np.int32)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(len(x), dtype=np.int32)
    plan = np.empty(5 * len(axes), dtype=np.int32)
    rc = f(ptrs, C.c_void_p(ns.ctypes.data), len(axes), int(budget), int(which), C.c_void_p(x.ctypes.data),
           C.c_int64(len(x)), C.c_void_p(out.ctypes.data), C.c_void_p(plan.ctypes.data))
    assert rc == 0
    return out, plan.reshape(len(axes), 5)


def _expected(a, x):
    return np.clip(np.searchsorted(a, x, side="right") - 1, 0, len(a) - 2)


def _probes(a, rng, n_random=4000):
    """every node, the doubles next to every node, and random points inside the axis"""
    near = np.concatenate([a, np.nextafter(a, -np.inf), np.nextafter(a, np.inf)])
    x = np.concatenate([near, rng.uniform(a[0], a[-1], n_random),
                        a[0] + (a[-1] - a[0]) * rng.random(n_random) ** 4])
    return x[(x >= a[0]) & (x <= a[-1])]


def _mist_axes():
    teff, logg, feh, av = grids.bc_axes()
    return [grids.MIST_FEHS, grids.mist_masses(), teff, logg, feh, av]


def test_mist_axes_every_budget():
    rng = np.random.default_rng(5)
    axes = _mist_axes()
    for budget in (1, 8, 64, 256, 736, 1024, 4096):
        for which, a in enumerate(axes):
            x = _probes(a, rng)
            got, plan = _brackets(axes, which, x, budget)
            np.testing.assert_array_equal(got, _expected(a, x))
        assert plan[:, 0].sum() <= max(budget, len(axes))          # one byte per axis is the floor (plain bisection)
    # what the library stages for the MIST tables (736 B): 1-3 levels instead of 4-8
    _, plan = _brackets(axes, 0, axes[0][:1], 736)
    full = [int(np.ceil(np.log2(len(a)))) for a in axes]
    assert all(plan[:, 2] < np.array(full)) and max(plan[:, 2]) <= 3, plan


def test_iso_axes_and_thinned_eep_axis():
    rng = np.random.default_rng(6)
    thin = np.sort(rng.choice(grids.mist_eeps(), 700, replace=False))
    coarse = np.append(thin[::8], thin[-1])            # every 8th node of a thinned EEP axis + its last node, as staged
    axes = [grids.mist_log_ages(), grids.MIST_FEHS] + list(grids.bc_axes()) + [coarse]
    for which, a in enumerate(axes):
        x = _probes(a, rng)
        got, _ = _brackets(axes, which, x, 1024)
        np.testing.assert_array_equal(got, _expected(a, x))


@pytest.mark.parametrize("seed", range(12))
def test_random_axes(seed):
    rng = np.random.default_rng(100 + seed)
    kinds = []
    for _ in range(6):
        n = int(rng.integers(3, 257))
        kind = rng.integers(0, 6)
        if kind == 0:
            a = np.cumsum(rng.random(n) + 1e-3) + rng.uniform(-50, 50)                       # near-uniform, any sign
        elif kind == 1:
            a = 10.0 ** np.sort(rng.uniform(-3, 6, n))                                        # log-spaced, positive
        elif kind == 2:
            a = np.sort(np.concatenate([rng.normal(0, 1e-3, n // 2), rng.uniform(-5, 5, n - n // 2)]))   # clustered
        elif kind == 3:
            a = -(10.0 ** np.sort(rng.uniform(-2, 4, n)))[::-1]                               # negative, log-spaced
        elif kind == 4:
            a = np.sort(rng.uniform(0, 1, n)) * 10.0 ** rng.integers(-200, 200)               # extreme magnitudes
        else:
            a = np.arange(n) * 0.25 - 4.0                                                     # exactly uniform
        a = np.unique(a)
        if len(a) < 3:
            a = np.array([0.0, 1.0, 2.0])
        kinds.append(a)
    for budget in (16, 700, 2048):
        for which, a in enumerate(kinds):
            x = _probes(a, rng, 1500)
            got, _ = _brackets(kinds, which, x, budget)
            np.testing.assert_array_equal(got, _expected(a, x))


def test_axes_too_long_for_byte_tables_keep_the_full_bisection():
    rng = np.random.default_rng(9)
    a = np.cumsum(rng.random(1000) + 0.01)
    x = _probes(a, rng)
    got, plan = _brackets([a], 0, x, 1024)
    np.testing.assert_array_equal(got, _expected(a, x))
    assert tuple(plan[0, :2]) == (1, 1000)
