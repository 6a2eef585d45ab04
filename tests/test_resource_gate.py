"""The build's resource gate (isochrones_amd/csrc/resources.py, enforced by csrc/build.py): no kernel of libiso_hip.so may
use accumulation registers, and scratch stays inside each family's budget.  Round 3 shipped 20 kernels that lived partly
in AGPRs - the regime in which one sampler instantiation made wrong accept / reject decisions that no test saw."""
import json
import os

import pytest

from isochrones_amd.csrc import build as B
from isochrones_amd.csrc import resources as R

SAMPLE = """
/x/iso_fast_mag.hip:12:1: remark: Function Name: _ZN3iso5fastk17k_interp_mag_fastILi0ELi1EEEvNS_8FastArgsENS_6MagOutE [-Rpass-analysis=kernel-resource-usage]
   12 | {
      | ^
/x/iso_fast_mag.hip:12:1: remark:     TotalSGPRs: 106 [-Rpass-analysis=kernel-resource-usage]
/x/iso_fast_mag.hip:12:1: remark:     VGPRs: 79 [-Rpass-analysis=kernel-resource-usage]
/x/iso_fast_mag.hip:12:1: remark:     AGPRs: 3 [-Rpass-analysis=kernel-resource-usage]
/x/iso_fast_mag.hip:12:1: remark:     ScratchSize [bytes/lane]: 24 [-Rpass-analysis=kernel-resource-usage]
/x/iso_fast_mag.hip:12:1: remark:     Dynamic Stack: False [-Rpass-analysis=kernel-resource-usage]
/x/iso_fast_mag.hip:12:1: remark:     Occupancy [waves/SIMD]: 6 [-Rpass-analysis=kernel-resource-usage]
/x/iso_fast_mag.hip:12:1: remark:     SGPRs Spill: 8 [-Rpass-analysis=kernel-resource-usage]
/x/iso_fast_mag.hip:12:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
/x/iso_fast_mag.hip:12:1: remark:     LDS Size [bytes/block]: 0 [-Rpass-analysis=kernel-resource-usage]
/x/iso_hip.hip:20:1: remark: Function Name: _ZN3iso12_GLOBAL__N_111k_unit_cubeEPKNS_8DevModelEPdlll [-Rpass-analysis=kernel-resource-usage]
/x/iso_hip.hip:20:1: remark:     VGPRs: 16 [-Rpass-analysis=kernel-resource-usage]
/x/iso_hip.hip:20:1: remark:     AGPRs: 0 [-Rpass-analysis=kernel-resource-usage]
/x/iso_hip.hip:20:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
"""


def test_remarks_are_parsed_per_kernel_with_demangled_names():
    t = R.parse(SAMPLE)
    assert set(t) == {"k_interp_mag_fast<0, 1>", "k_unit_cube"}
    assert t["k_interp_mag_fast<0, 1>"] == dict(sgpr=106, vgpr=79, agpr=3, scratch=24, waves=6, sgpr_spill=8, vgpr_spill=0, lds=0)
    assert R.family("k_interp_mag_fast<0, 1>") == "k_interp_mag_fast"


def test_agprs_and_scratch_over_budget_are_violations():
    t = R.parse(SAMPLE)
    bad = R.violations(t, scratch_budget={}, default_scratch=0, max_agpr=0)
    assert len(bad) == 2 and "3 AGPRs" in bad[0] and "24 B/lane" in bad[1]
    assert R.violations(t, scratch_budget={"k_interp_mag_fast": 24}, default_scratch=0, max_agpr=3) == []
    assert "k_interp_mag_fast" in R.render(t)


def test_the_library_as_built_is_inside_the_budget():
    """Every kernel hipcc compiled for libiso_hip.so: zero AGPRs, scratch within its family's budget; the table covers
    every translation unit's remarks (build() itself raises ResourceBudgetError otherwise - this pins the stored table)."""
    B.build()
    t = B.resource_table()
    assert len(t) > 300
    assert R.MAX_AGPR == 0 and all(r["agpr"] == 0 for r in t.values())
    assert R.violations(t) == []
    # every kernel's registers fit two waves per SIMD: that is what keeps the allocator out of the accumulation registers
    assert all(r["vgpr"] <= 256 and r["waves"] >= 2 for r in t.values()), [n for n, r in t.items() if r["waves"] < 2]
    n_remarks = 0
    for f in os.listdir(B.OBJDIR):
        if f.endswith(".res"):
            n_remarks += open(os.path.join(B.OBJDIR, f), errors="replace").read().count("Function Name:")
    if n_remarks:                                  # the objects' logs are not shipped to the GPU box
        assert n_remarks == len(t)


def test_gate_refuses_a_table_with_an_agpr_kernel(monkeypatch):
    t = dict(B.resource_table()) if os.path.exists(B.RESOURCES) else R.parse(SAMPLE)
    t = {k: dict(v) for k, v in t.items()}
    first = sorted(t)[0]
    t[first]["agpr"] = 2
    bad = R.violations(t)
    assert any(first in b and "AGPRs" in b for b in bad)
