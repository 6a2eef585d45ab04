"""The build's check of the generated code (isochrones_amd/csrc/isa_check.py, run by csrc/build.py on every translation
unit): no vector instruction may stand between the target of an `if`'s skip branch (`s_cbranch_execz`) and the `s_or_b64 exec`
that switches the skipped lanes back on.  That is the fault behind the wrong (isochrone, 3 stars, 9 bands) sampler kernel
of round 3 (profiles/r04/miscompile_hunt.md): the register allocator's copies of the primary's magnitude into AGPRs (or, in
the no-AGPR build, spill stores) were placed ahead of the restore and, the `then` side being skipped by every lane in
practice, wrote nothing.  tests/golden/isa_exec_restore_fault.txt is that block, cut from the wrong kernel's disassembly."""
import json
import os

from isochrones_amd.csrc import build as B
from isochrones_amd.csrc import isa_check as I

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "isa_exec_restore_fault.txt")

HEAD = "0000000000001000 <_Z1kv>:\n"


def listing(body):
    """llvm-objdump style lines at consecutive 4-byte addresses from 0x1000; `-> N` in a branch is instruction index N."""
    rows = [r.strip() for r in body.strip().split("\n")]
    out = [HEAD]
    for k, r in enumerate(rows):
        tail = ""
        if "->" in r:
            r, n = r.split("->")
            tail = " <_Z1kv+0x%x>" % (4 * int(n))
        out.append("\t%-58s // %012X: BF800000%s\n" % (r.strip(), 0x1000 + 4 * k, tail))
    return out


def test_the_wrong_kernels_block_is_found():
    found = I.scan_listing(open(GOLDEN))
    assert len(found) == 1
    sym, branch, target, text, n = found[0]
    assert "k_stretch_persistILi1ELi3ELi9E" in sym and (branch, target) == (0x82800, 0x829AC)
    assert text.startswith("v_accvgpr_write_b32 a89") and n == 0
    assert "k_stretch_persist<1, 3, 9, false, false, true, true>" in I.render(found)


def test_a_copy_ahead_of_the_restore_is_a_fault_and_the_restore_first_is_not():
    bad = listing("""
        s_and_saveexec_b64 s[28:29], s[2:3]
        s_cbranch_execz 2 -> 4
        v_mul_f64 v[0:1], v[0:1], v[0:1]
        v_mov_b32_e32 v9, v1
        s_mov_b64 s[10:11], s[30:31]
        scratch_store_dwordx2 off, v[14:15], off offset:264
        s_or_b64 exec, exec, s[28:29]
        s_endpgm""")
    f = I.scan_listing(bad)
    assert len(f) == 1 and f[0][3].startswith("scratch_store_dwordx2") and f[0][4] == 1
    good = listing("""
        s_and_saveexec_b64 s[28:29], s[2:3]
        s_cbranch_execz 2 -> 4
        v_mul_f64 v[0:1], v[0:1], v[0:1]
        v_mov_b32_e32 v9, v1
        s_mov_b64 s[10:11], s[30:31]
        s_or_b64 exec, exec, s[28:29]
        scratch_store_dwordx2 off, v[14:15], off offset:264
        s_endpgm""")
    assert I.scan_listing(good) == []


def test_lane_moves_and_all_lanes_done_exits_are_not_faults():
    spill_lane = listing("""
        s_and_saveexec_b64 s[18:19], vcc
        s_cbranch_execz 1 -> 3
        v_add_f64 v[0:1], v[0:1], v[2:3]
        v_writelane_b32 v255, s24, 16
        s_or_b64 exec, exec, s[18:19]
        s_endpgm""")
    assert I.scan_listing(spill_lane) == []              # v_writelane is not exec-masked
    done_exit = listing("""
        s_or_b64 exec, exec, s[0:1]
        s_cbranch_execz 2 -> 4
        s_branch 16
        s_branch 16
        v_mov_b32_e32 v42, 0
        s_and_saveexec_b64 s[0:1], s[16:17]
        s_endpgm""")
    assert I.scan_listing(done_exit) == []               # exec stays 0 on that path: dead code, not a lost write


def test_every_translation_unit_of_the_library_was_scanned_and_is_clean():
    B.build()
    objs = [os.path.join(B.OBJDIR, os.path.basename(s)[:-4]) for s in B.sources()]
    assert len(objs) >= 10
    n_device = 0
    for o in objs:
        with open(o + ".isa") as f:
            rec = json.load(f)
        assert rec["object"] == open(o + ".dig").read().strip(), "scan of another object"
        assert rec["found"] == [], I.render([tuple(r) for r in rec["found"]])
        n_device += os.path.getsize(o + ".o") > 1 << 19        # (objects that carry device code: hundreds of KB and up)
    assert n_device >= 8


def test_gate_refuses_a_faulty_scan(monkeypatch, tmp_path):
    """build() with one translation unit's record replaced by the golden finding: IsaFault, and no stamp that would call the
    library up to date; with ISOCHRONES_AMD_ISA_GATE=0 a warning."""
    import pytest
    B.build()
    o = os.path.join(B.OBJDIR, "iso_fast_iso3.isa")
    keep = open(o).read()
    rec = json.loads(keep)
    rec["found"] = [list(r) for r in I.scan_listing(open(GOLDEN))]
    stamp = open(B.STAMP).read()
    res = open(B.RESOURCES).read()
    try:
        with open(o, "w") as f:
            json.dump(rec, f)
        with pytest.raises(I.IsaFault, match="ahead of the exec restore"):
            B.build(verbose=True)
        assert not os.path.exists(B.STAMP)
        monkeypatch.setenv("ISOCHRONES_AMD_ISA_GATE", "0")
        B.build(verbose=True)
    finally:
        with open(o, "w") as f:
            f.write(keep)
        with open(B.STAMP, "w") as f:
            f.write(stamp)
        with open(B.RESOURCES, "w") as f:
            f.write(res)
    assert B.up_to_date()
