"""Synthetic MIST-shaped model / bolometric-correction tables.

The real MIST tables are downloaded by the reference on first use
(reference: isochrones/grid.py:80-101, isochrones/mist/models.py:116-124) and are not
available offline.  Everything here is a closed-form recipe: a table is a pure function of
its axis vectors, so the golden-vector generator (which feeds the *reference* code), the CPU
oracle, the parity tests and ``bench.py`` all see bit-identical tables without shipping files.

What is kept faithful to the reference is the *schema* the numeric path consumes:

* track table   index (initial_feh, initial_mass, EEP), 18 named columns incl. ``dt_deep``
  (reference: isochrones/mist/models.py:167-173,399), ragged tails NaN-padded
  (reference: isochrones/interp.py:598-609, isochrones/mist/eep.py:1-59);
* isochrone table index (log10 age, feh, EEP), 16 named columns incl. ``dm_deep``
  (reference: isochrones/mist/models.py:99, isochrones/models.py:28-41,159);
* BC table index (Teff, logg, [Fe/H], Av), one column per band
  (reference: isochrones/bc.py:25, isochrones/mist/bc.py:159-163).

The column *values* are smooth, vaguely stellar functions chosen so that the surface
(Teff, logg, feh) of most of the table falls inside the BC table and the age/mass columns
cross the prior bounds somewhere (so every branch of the posterior is exercised).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# axes
# --------------------------------------------------------------------------------------

#: MIST v1.2 initial [Fe/H] nodes (reference: isochrones/mist/models.py:39-58)
MIST_FEHS = np.array(
    [-4.0, -3.5, -3.0, -2.5, -2.0, -1.75, -1.5, -1.25, -1.0, -0.75, -0.5, -0.25, 0.0, 0.25, 0.5]
)

MIST_N_EEP = 1710  # reference: isochrones/mist/models.py:63

TRACK_COLUMNS = (
    "eep", "feh", "mass", "initial_mass", "radius", "density", "logTeff", "Teff", "logg",
    "logL", "Mbol", "delta_nu", "nu_max", "phase", "interpolated", "star_age", "age", "dt_deep",
)

ISO_COLUMNS = (
    "eep", "age", "feh", "mass", "initial_mass", "radius", "density", "logTeff", "Teff",
    "logg", "logL", "Mbol", "delta_nu", "nu_max", "phase", "dm_deep",
)

#: default MIST band short names (reference: isochrones/mist/bc.py:159)
DEFAULT_BANDS = ("J", "H", "K", "G", "BP", "RP", "W1", "W2", "W3", "TESS", "Kepler")

# every band name the synthetic BC table knows; index in this tuple seeds its coefficients
KNOWN_BANDS = DEFAULT_BANDS + ("V", "B", "g", "r", "i", "z", "U", "R", "I")

_MSUN_G = 1.98840987e33
_RSUN_CM = 6.957e10


def mist_masses() -> np.ndarray:
    """196 initial-mass nodes with the MIST v1.2 spacing pattern."""
    parts = [
        np.arange(10, 31, 5) / 100.0,        # 0.10 .. 0.30 step 0.05
        np.arange(31, 41, 1) / 100.0,        # 0.31 .. 0.40 step 0.01
        np.arange(45, 91, 5) / 100.0,        # 0.45 .. 0.90 step 0.05
        np.arange(92, 281, 2) / 100.0,       # 0.92 .. 2.80 step 0.02
        np.arange(30, 81, 2) / 10.0,         # 3.0 .. 8.0 step 0.2
        np.arange(9, 21, 1) * 1.0,           # 9 .. 20
        np.arange(22, 41, 2) * 1.0,          # 22 .. 40
        np.arange(45, 151, 5) * 1.0,         # 45 .. 150
        np.arange(175, 301, 25) * 1.0,       # 175 .. 300
    ]
    m = np.concatenate(parts)
    assert m.size == 196 and np.all(np.diff(m) > 0)
    return m


def mist_eeps() -> np.ndarray:
    return np.arange(1, MIST_N_EEP + 1, dtype=np.float64)


def mist_log_ages() -> np.ndarray:
    """107 log10(age/yr) nodes 5.0 .. 10.3 step 0.05."""
    a = np.arange(100, 207, 1) * 0.05
    assert a.size == 107
    return a


def bc_axes() -> tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """(Teff[70], logg[26], [Fe/H][18], Av[13]) axes with MIST-BC-like extents."""
    teff = np.round(2500.0 * (200000.0 / 2500.0) ** (np.arange(70) / 69.0), 3)
    logg = -4.0 + 0.5 * np.arange(26)
    feh = np.array([-4.0, -3.5, -3.0, -2.75, -2.5, -2.25, -2.0, -1.75, -1.5, -1.25, -1.0,
                    -0.75, -0.5, -0.25, 0.0, 0.25, 0.5, 0.75])
    av = np.array([0.0, 0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.6, 0.8, 1.0, 2.0, 4.0, 6.0])
    return teff, logg, feh, av


# --------------------------------------------------------------------------------------
# ragged-track model
# --------------------------------------------------------------------------------------

def track_max_eep(mass: np.ndarray, feh: np.ndarray) -> np.ndarray:
    """Last populated EEP of a (mass, feh) track — a smooth stand-in for the reference's
    lookup table (isochrones/mist/eep.py:1-59): low-mass tracks stop on the main sequence,
    massive ones at carbon burning, intermediate ones run to the white-dwarf sequence."""
    mass, feh = np.broadcast_arrays(np.asarray(mass, float), np.asarray(feh, float))
    out = np.full(mass.shape, 1710.0)
    out[mass < 0.6] = 454.0
    out[(mass >= 0.6) & (mass < 0.7)] = 808.0
    out[mass >= 6.0] = 808.0
    metal_poor = feh <= -2.5
    out[metal_poor & (mass >= 0.7) & (mass < 2.4)] = 808.0
    out[metal_poor & (mass >= 2.4) & (mass < 5.0)] = 1409.0
    return out


def iso_eep_range(log_age: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """(first, last) populated EEP of an isochrone: old isochrones have lost their pre-MS
    points, young ones have not reached the late phases yet."""
    a = np.asarray(log_age, float)
    first = np.maximum(1.0, np.round(40.0 * (a - 6.0)))
    last = np.where(a < 7.0, 808.0, np.where(a < 8.0, 1409.0, 1710.0))
    return first, last


# --------------------------------------------------------------------------------------
# column physics (toy)
# --------------------------------------------------------------------------------------

def _surface(lm, fe, y, cur_mass):
    """Shared toy stellar surface: lm = log10(initial mass), fe = initial [Fe/H],
    y in [0,1] = post-main-sequence evolution coordinate."""
    damp = 0.35 + 0.325 * (1.0 + np.tanh(2.0 * lm))
    logTeff = (3.762 + 0.28 * lm - 0.025 * lm * lm - 0.015 * fe
               - 0.22 * y * y * damp + 0.01 * np.sin(7.0 * y))
    logg = 4.43 + 0.02 * fe - 0.35 * lm - 3.6 * y ** 2.2
    feh = fe + 0.02 * np.sin(np.pi * y) * (1.0 + 0.3 * lm)
    logR = 0.5 * (np.log10(cur_mass) - (logg - 4.438))
    logL = 2.0 * logR + 4.0 * (logTeff - 3.7617)
    Teff = 10.0 ** logTeff
    radius = 10.0 ** logR
    density = cur_mass * _MSUN_G / (4.0 / 3.0 * np.pi * (radius * _RSUN_CM) ** 3)
    Mbol = 4.74 - 2.5 * logL
    delta_nu = 135.1 * np.sqrt(cur_mass / radius ** 3)
    nu_max = 3090.0 * cur_mass / radius ** 2 / np.sqrt(Teff / 5777.0)
    return dict(logTeff=logTeff, Teff=Teff, logg=logg, feh=feh, logL=logL, Mbol=Mbol,
                radius=radius, density=density, delta_nu=delta_nu, nu_max=nu_max)


def synthetic_track_grid(fehs=None, masses=None, eeps=None, ragged=True, columns=TRACK_COLUMNS):
    """Evolution-track table ``G[n_feh, n_mass, n_eep, n_col]`` (C-contiguous float64,
    last axis = column; the reference's DFInterpolator.grid layout, isochrones/interp.py:607-609).

    Returns ``(grid, (fehs, masses, eeps), columns)``.
    """
    fehs = MIST_FEHS if fehs is None else np.asarray(fehs, float)
    masses = mist_masses() if masses is None else np.asarray(masses, float)
    eeps = mist_eeps() if eeps is None else np.asarray(eeps, float)
    F = fehs[:, None, None]
    M = masses[None, :, None]
    E = eeps[None, None, :]
    x = (E - 1.0) / (MIST_N_EEP - 1.0)
    lm = np.log10(M)

    t_end = 10.35 - 2.2 * lm + 0.25 * lm * lm + 0.05 * F      # log10 age at the last EEP
    age = 5.0 + (t_end - 5.0) * (1.0 - (1.0 - x) ** 3)
    dt_deep = (t_end - 5.0) * 3.0 * (1.0 - x) ** 2 / (MIST_N_EEP - 1.0)
    cur_mass = M * (1.0 - 0.08 * x ** 3)
    s = _surface(lm, F, x, cur_mass)

    shape = (fehs.size, masses.size, eeps.size)
    cols = {
        "eep": np.broadcast_to(E, shape),
        "mass": np.broadcast_to(cur_mass, shape),
        "initial_mass": np.broadcast_to(M, shape),
        "phase": np.broadcast_to(6.0 * x, shape),
        "interpolated": np.zeros(shape),
        "star_age": 10.0 ** age,
        "age": age,
        "dt_deep": np.broadcast_to(dt_deep, shape),
    }
    cols.update(s)
    grid = np.empty(shape + (len(columns),), dtype=np.float64)
    for j, name in enumerate(columns):
        grid[..., j] = cols[name]
    if ragged:
        last = track_max_eep(M[..., 0], F[..., 0])            # [n_feh, n_mass]
        dead = E > last[..., None]
        grid[np.broadcast_to(dead, shape)] = np.nan
    return grid, (fehs, masses, eeps), tuple(columns)


def synthetic_iso_grid(ages=None, fehs=None, eeps=None, ragged=True, columns=ISO_COLUMNS):
    """Isochrone table ``G[n_age, n_feh, n_eep, n_col]``; returns ``(grid, axes, columns)``."""
    ages = mist_log_ages() if ages is None else np.asarray(ages, float)
    fehs = MIST_FEHS if fehs is None else np.asarray(fehs, float)
    eeps = mist_eeps() if eeps is None else np.asarray(eeps, float)
    A = ages[:, None, None]
    F = fehs[None, :, None]
    E = eeps[None, None, :]
    x = (E - 1.0) / (MIST_N_EEP - 1.0)

    lm_to = (10.0 - A) / 2.4 + 0.02 * F                       # log10 turn-off mass
    span = lm_to + 0.06 + 1.0
    lm = -1.0 + span * (1.0 - (1.0 - x) ** 2.5)
    M = 10.0 ** lm
    dm_deep = M * np.log(10.0) * span * 2.5 * (1.0 - x) ** 1.5 / (MIST_N_EEP - 1.0)
    y = x ** 4
    cur_mass = M * (1.0 - 0.08 * y ** 3)
    s = _surface(lm, F, y, cur_mass)

    shape = (ages.size, fehs.size, eeps.size)
    cols = {
        "eep": np.broadcast_to(E, shape),
        "age": np.broadcast_to(A, shape),
        "mass": cur_mass,
        "initial_mass": M,
        "phase": np.broadcast_to(6.0 * y, shape),
        "dm_deep": dm_deep,
    }
    cols.update(s)
    grid = np.empty(shape + (len(columns),), dtype=np.float64)
    for j, name in enumerate(columns):
        grid[..., j] = np.broadcast_to(cols[name], shape)
    if ragged:
        first, last = iso_eep_range(ages)
        dead = (E < first[:, None, None]) | (E > last[:, None, None])
        grid[np.broadcast_to(dead, shape)] = np.nan
    return grid, (ages, fehs, eeps), tuple(columns)


def synthetic_bc_grid(bands=DEFAULT_BANDS, axes=None):
    """Bolometric-correction table ``G[nT, ng, nf, nA, n_band]``; returns ``(grid, axes, bands)``."""
    teff, logg, feh, av = bc_axes() if axes is None else [np.asarray(a, float) for a in axes]
    T = np.log10(teff / 5772.0)[:, None, None, None]
    g = logg[None, :, None, None]
    f = feh[None, None, :, None]
    a = av[None, None, None, :]
    shape = (teff.size, logg.size, feh.size, av.size)
    grid = np.empty(shape + (len(bands),), dtype=np.float64)
    for j, b in enumerate(bands):
        k = KNOWN_BANDS.index(b) if b in KNOWN_BANDS else (sum(map(ord, b)) % 23) + len(KNOWN_BANDS)
        lam = 0.35 + 0.21 * k                                   # pseudo-wavelength, micron
        c0 = 0.4 * np.cos(0.9 * k) - 0.1
        c1 = 2.2 * (lam - 0.55) - 3.5
        c2 = -6.0 + 0.3 * k
        ext = 1.0 / (0.3 + lam ** 1.3)                          # A_band / A_V
        bc = (c0 + c1 * T * (lam - 0.55) + c2 * T * T + 0.012 * g * (1.0 + T) + 0.03 * f * (1.0 - 0.1 * k)
              - a * ext * (1.0 + 0.02 * T))
        grid[..., j] = np.broadcast_to(bc, shape)
    return grid, (teff, logg, feh, av), tuple(bands)
