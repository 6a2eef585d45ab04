"""Small host helpers users of the reference import from ``isochrones.utils`` (utils.py:13-15, 43-95):
magnitude addition and the (separation, PA) distance used when observations are matched to sources.
On the device the same flux sum runs inside the fused kernels (``fast_addmags``, utils.py:67-75)."""
from __future__ import annotations

import numpy as np


def band_pairs(bands):
    """Every band paired with the last one (the reference's colour pairs)."""
    return [(bands[i], bands[-1]) for i in range(len(bands) - 1)]


def addmags(*mags):
    """Magnitude of the summed flux.  Entries are magnitudes or ``(mag, unc)`` pairs; with pairs the
    combined uncertainty is returned too (flux errors added in quadrature)."""
    tot, uncs = 0.0, []
    for mag in mags:
        if isinstance(mag, (tuple, list)) and len(mag) == 2:
            m, dm = mag
            f = 10 ** (-0.4 * m)
            tot = tot + f
            uncs.append(f * (1 - 10 ** (-0.4 * dm)))
        else:
            tot = tot + 10 ** (-0.4 * np.asarray(mag, dtype=float))
    totmag = -2.5 * np.log10(tot)
    if uncs:
        f_unc = np.sqrt(np.sum(np.square(uncs)))
        return totmag, -2.5 * np.log10(1 - f_unc / tot)
    return totmag


def fast_addmags(mags):
    """``-2.5 log10(sum 10^(-0.4 m))`` over a sequence of magnitudes."""
    tot = 0.0
    for m in mags:
        tot += 10 ** (-0.4 * m)
    return -2.5 * np.log10(tot)


def distance(pos0, pos1):
    """Angular distance between two positions given as (separation, position angle in degrees)."""
    r0, pa0 = pos0
    r1, pa1 = pos1
    dra = r1 * np.sin(pa1 * np.pi / 180) - r0 * np.sin(pa0 * np.pi / 180)
    ddec = r1 * np.cos(pa1 * np.pi / 180) - r0 * np.cos(pa0 * np.pi / 180)
    return np.sqrt(dra ** 2 + ddec ** 2)
