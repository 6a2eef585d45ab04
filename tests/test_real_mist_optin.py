"""Opt-in known-answer checks that need the REAL MIST tables: they run only when
$ISOCHRONES points at the reference's data directory *and* pandas can read its HDF5 stores
(pytables).  Offline (this image) they are skipped; the recorded values are the reference's own:
isochrones/tests/test_basic.py:16-18 and docs notebooks."""
import os

import numpy as np
import pytest

KATS_LOGG = [((632, 7.55, -1.75), 2.4117770214014103, 0.0),       # exact grid point (test_basic.py:16)
             ((355, 9.653, 0.0), 4.4124675, 1e-6),                # test_basic.py:17
             ((700, 9.3, -0.03), 2.24831956, 1e-6)]               # test_basic.py:18


def _real_iso_table():
    root = os.environ.get("ISOCHRONES")
    if not root or not os.path.isdir(os.path.join(root, "mist")):
        pytest.skip("real MIST tables not available ($ISOCHRONES)")
    try:
        import tables  # noqa: F401
        import pandas as pd
    except Exception:
        pytest.skip("pytables not installed: cannot read the reference's HDF5 stores")
    h5 = [f for f in os.listdir(os.path.join(root, "mist")) if f.endswith("full_isos.h5")]
    if not h5:
        pytest.skip("no isochrone HDF5 store found")
    return pd.read_hdf(os.path.join(root, "mist", h5[0]))


@pytest.mark.gpu
def test_reference_logg_kats_on_real_tables():
    import isochrones_amd as ia
    from isochrones_amd.interp import DFInterpolator
    df = _real_iso_table()
    dfi = DFInterpolator(df)
    ci = dfi.column_index["logg"]
    for (eep, age, feh), want, rtol in KATS_LOGG:
        got = dfi([age, feh, float(eep)], ["logg"])[0]
        assert np.isclose(got, want, rtol=max(rtol, 1e-13)), (eep, age, feh, got, want)
