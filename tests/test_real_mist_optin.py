"""Opt-in known-answer checks that need the REAL MIST tables: they run only when $ISOCHRONES points at
the reference's data directory with its isochrone cache (``mist/full_grid_v1.2_vvcrit0.4_full_isos.npz``, read with numpy
alone: isochrones_amd/mist.py - no pytables).  Offline (this image) they are skipped; the recorded values are the
reference's own: isochrones/tests/test_basic.py:16-18."""
import os

import numpy as np
import pytest

KATS_LOGG = [((632, 7.55, -1.75), 2.4117770214014103, 0.0),       # exact grid point (test_basic.py:16)
             ((355, 9.653, 0.0), 4.4124675, 1e-6),                # test_basic.py:17
             ((700, 9.3, -0.03), 2.24831956, 1e-6)]               # test_basic.py:18


def _real_iso_table():
    from isochrones_amd import mist
    if not mist.available(tracks=False):
        pytest.skip("real MIST isochrone cache not available (%s)" % mist.mist_paths(tracks=False)["full_grid"])
    dfi = mist.load_model_table(tracks=False)
    if dfi.grid.shape[:3] != (107, 15, 1710):
        pytest.skip("the cache under $ISOCHRONES is not the full MIST isochrone grid (shape %s)" % (dfi.grid.shape,))
    return dfi


@pytest.mark.gpu
def test_reference_logg_kats_on_real_tables():
    dfi = _real_iso_table()
    for (eep, age, feh), want, rtol in KATS_LOGG:
        got = dfi([age, feh, float(eep)], ["logg"])[0]
        assert np.isclose(got, want, rtol=max(rtol, 1e-13)), (eep, age, feh, got, want)
