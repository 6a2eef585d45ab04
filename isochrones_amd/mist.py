"""Reference-style entry points for the MIST grids (isochrones/mist/__init__.py: ``MIST_Isochrone``,
``MIST_EvolutionTrack``; isochrones/isochrone.py:48-78 ``get_ichrone``).  The real MIST tables need the
network and HDF5 support this image lacks, so these build the MIST-shaped synthetic tables of
``isochrones_amd.grids`` (same axes, columns and ragged structure); with real tables, construct the
interpolators from ``DFInterpolator(df)`` / ``ingest.load_full_grid_npz`` instead."""
from __future__ import annotations

from .models import get_ichrone


def MIST_Isochrone(bands=None, **kwargs):
    """(eep, age, feh, distance, AV) interpolator over the [107, 15, 1710] isochrone grid."""
    return get_ichrone("mist", bands=bands, tracks=False, **kwargs)


def MIST_EvolutionTrack(bands=None, **kwargs):
    """(mass, eep, feh, distance, AV) interpolator over the [15, 196, 1710] evolution-track grid."""
    return get_ichrone("mist", bands=bands, tracks=True, **kwargs)


# the reference's "Basic" variants skip the companion grid; here both forms are the same object
MIST_BasicIsochrone = MIST_Isochrone
MIST_BasicEvolutionTrack = MIST_EvolutionTrack
