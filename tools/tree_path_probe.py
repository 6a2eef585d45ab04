import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import isochrones_amd as ia
from isochrones_amd import _cabi
sys.path.insert(0, "/root/repo/tests")
from tests.test_gpu_dispatch_table import make_ic, tree_model
for nb in (8, 9, 12):
    ic, lo, hi = make_ic("iso", ia.grids.KNOWN_BANDS[:nb])
    mod = tree_model(ic, nb, 2)
    _cabi.trace_kernels(True)
    x = np.tile(np.array([340.0, 320.0, 9.6, 0.0, 330.0, 0.1]), (64, 1))
    print(nb, mod.lnpost(x)[:2], _cabi.traced_kernels(), _cabi.lib().iso_last_error())
