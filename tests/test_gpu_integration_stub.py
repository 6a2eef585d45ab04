"""INTEGRATION.md section 1 is the artefact a maintainer of the reference would copy: the ctypes stub that binds
libiso_hip.so from isochrones/.  This test extracts that code block and runs it verbatim (fresh interpreter,
libiso_hip.so found through LD_LIBRARY_PATH as the stub's bare CDLL("libiso_hip.so") expects) against objects that
carry the reference's attribute names (DFInterpolator: .grid / .index_columns / .column_index / .ndim,
isochrones/interp.py:571-588; ModelGridInterpolator: .model_grid.interp / .bc_grid.interp / .eep_replaces,
isochrones/models.py:253-445) and compares what `interp_mags` returns with the CPU oracle."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import sys, types
sys.path.insert(0, %(root)r)
import numpy as np
stub = types.ModuleType("isochrones_hip_stub")
exec(compile(open(%(stub)r).read(), "INTEGRATION.md#1", "exec"), stub.__dict__)       # the block, as it is written

import isochrones_amd as ia                       # only for the synthetic tables and the oracle's view of them
from tests import _fixtures as fx


class RefDFInterpolator:                          # what the stub reads of isochrones.interp.DFInterpolator
    def __init__(self, dfi):
        self.grid = np.ascontiguousarray(dfi.grid)
        self.index_columns = tuple(np.ascontiguousarray(a, dtype=float) for a in dfi.index_columns)
        self.ndim = len(self.index_columns)
        self.column_index = dict(dfi.column_index)


class RefGrid:
    def __init__(self, dfi):
        self.interp = RefDFInterpolator(dfi)


class RefIC:                                      # isochrones.models.ModelGridInterpolator, as far as the stub goes
    def __init__(self, ic):
        self.model_grid, self.bc_grid, self.eep_replaces = RefGrid(ic.model_grid.interp), RefGrid(ic.bc_grid.interp), ic.eep_replaces


rng = np.random.default_rng(11)
for tracks in (True, False):
    bands = ("J", "K", "G")
    if tracks:
        ic = ia.synthetic_track(bands=bands, fehs=np.array([-1.0, -0.5, 0.0, 0.5]), masses=np.array([0.7, 0.9, 1.0, 1.1, 1.3, 2.0]),
                                eeps=np.arange(300.0, 420.0))
        pars = np.array([rng.uniform(0.65, 2.1, 5000), rng.uniform(295, 425, 5000), rng.uniform(-1.1, 0.6, 5000),
                         rng.uniform(50, 150, 5000), rng.uniform(0, 1, 5000)])
    else:
        ic = ia.synthetic_isochrone(bands=bands, ages=np.array([8.5, 9.0, 9.5, 10.0]), fehs=np.array([-1.0, -0.5, 0.0, 0.5]),
                                    eeps=np.arange(200.0, 400.0))
        pars = np.array([rng.uniform(195, 405, 5000), rng.uniform(8.4, 10.1, 5000), rng.uniform(-1.1, 0.6, 5000),
                         rng.uniform(50, 150, 5000), rng.uniform(0, 1, 5000)])
    handle = stub.bind(RefIC(ic))
    i_bands = [ic.bc_grid.interp.column_index[b] for b in bands]
    T, g, f, m = stub.interp_mags(handle, pars, i_bands)
    oT, og, of, om = fx.make_oracle_ic(ic).interp_mag(pars, i_bands)
    for got, want, name in ((T, oT, "Teff"), (g, og, "logg"), (f, of, "feh"), (m, om, "mags")):
        fx.assert_close(got, want, 1e-12, what="INTEGRATION stub %%s (%%s)" %% (name, "tracks" if tracks else "isochrones"))
    assert np.isfinite(om).all(axis=1).sum() > 1000 and np.isnan(om).any()
    # the posterior route: the stub's own mirror of iso_model_desc filled with a model's descriptor, iso_model_create, iso_lnpost
    import ctypes
    mod = ia.SingleStarModel(ic, Teff=(5770, 100), logg=(4.4, 0.1), J=(9.5, 0.03), K=(9.1, 0.03), G=(10.6, 0.02), parallax=(10.0, 0.2))
    src = mod.model_desc()
    assert ctypes.sizeof(stub.ModelDesc) == ctypes.sizeof(src)          # the stub's structure is the header's
    desc = stub.ModelDesc.from_buffer_copy(bytes(src))
    assert desc.n_bands == 3 and desc.prior_feh.kind == src.prior_feh.kind and desc.bound_hi[3] == src.bound_hi[3]
    mh = stub.model(handle, desc)
    got = stub.lnpost(mh, pars.T)
    want = fx.make_oracle_ic(ic).lnpost(src, pars, parts=False)
    fx.assert_close(got, want, 1e-9, atol=1e-10, what="INTEGRATION stub lnpost (%%s)" %% ("tracks" if tracks else "isochrones"))
    assert np.isfinite(want).sum() > 200
print("STUB_OK")
'''


def test_integration_md_stub_runs_verbatim_and_matches_the_oracle(tmp_path):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec1 = text[text.index("## 1."):text.index("## 2.")]
    blocks = re.findall(r"```python\n(.*?)```", sec1, flags=re.S)
    assert len(blocks) == 1 and "iso_ic_create" in blocks[0] and "def interp_mags" in blocks[0] and "def lnpost" in blocks[0]
    stub = tmp_path / "_hip.py"
    stub.write_text(blocks[0])
    driver = tmp_path / "driver.py"
    driver.write_text(DRIVER % {"root": ROOT, "stub": str(stub)})
    libdir = os.path.join(ROOT, "isochrones_amd", "csrc")
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([sys.executable, str(driver)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "STUB_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
