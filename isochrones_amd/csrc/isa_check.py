"""Build-time check of the generated gfx950 code for the one code-generation fault this project has met
(profiles/r04/miscompile_hunt.md): a vector instruction that runs in a control-flow JOIN block before the lanes the branch
had switched off are switched back on.

The shape in the ISA:

        s_and_saveexec_b64 s[28:29], s[2:3]     ; lanes that take the `then` side
        s_cbranch_execz  JOIN                   ; nobody does: skip it, exec == 0
        ...then side...
    JOIN:
        v_accvgpr_write_b32 a88, v12            ; <- register-allocator copy placed ahead of the restore:
        s_mov_b64 s[10:11], s[30:31]            ;    with exec == 0 it writes no lane at all
        s_or_b64 exec, exec, s[28:29]           ; lanes back on
        ... v_accvgpr_read_b32 v0, a88 ...      ; reads whatever a88 held before

Everything a wave executes at the target of an `s_cbranch_execz`, up to the first write of `exec`, runs with NO lane
enabled on the path that took the branch, so a vector ALU / memory / AGPR instruction there is either dead or - when it is
a copy whose destination is read later - a wrong value.  Correct structured-control-flow lowering puts the restore first.
The check: for every `s_cbranch_execz` that directly follows an instruction narrowing exec (the skip branch of an `if`; the
"all lanes finished" exits that follow an `s_or_b64 exec` are not of this kind - exec stays 0 there), the first write of
exec at the target being an `s_or_b64 exec, exec, sN`, no vector instruction may stand between the target and that
restore.  Scalar instructions (s_mov, s_load, s_waitcnt, s_nop, ...) are exec-independent and allowed.

Out of reach of a listing: an `if` whose `then` side is so short that the compiler emits no skip branch has no branch target
to anchor on, and a copy in front of its `s_or_b64 exec` cannot be told from the `then` side's own last instruction.  For
that form the net is tests/test_gpu_dispatch_table.py (every instantiation against the oracle, default priors).

    python -m isochrones_amd.csrc.isa_check libiso_hip.so          # exit 1 and a listing when anything is found
Used by build.py on every translation unit's object (ISOCHRONES_AMD_ISA_GATE=0 turns a finding into a warning) and by
tests/test_resource_gate.py.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor



def llvm_tool(name="llvm-objdump"):
    """The LLVM binary that belongs to the compiler the library is built with: $ISOCHRONES_AMD_LLVM_BIN, else next to the
    resolved hipcc (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), else $ROCM_PATH, /opt/rocm, PATH."""
    cands = []
    if os.environ.get("ISOCHRONES_AMD_LLVM_BIN"):
        cands.append(os.path.join(os.environ["ISOCHRONES_AMD_LLVM_BIN"], name))
    cc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if cc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(cc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", name), os.path.join(root, "llvm", "bin", name)]
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root:
            cands.append(os.path.join(root, "lib", "llvm", "bin", name))
    cands.append(shutil.which(name))
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError("%s not found (set ISOCHRONES_AMD_LLVM_BIN to the directory that holds it)" % name)


LANE_OPS = ("v_writelane_b32", "v_readlane_b32", "v_readfirstlane_b32")     # SGPR <-> one VGPR lane: not exec-masked
WINDOW = 64            # instructions looked at after a branch target before giving up (a restore is normally the first)

_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_SYM = re.compile(r"^[0-9A-Fa-f]+ <(\S+)>:")
_TGT = re.compile(r"<(\S+?)\+0x([0-9A-Fa-f]+)>")


class IsaFault(RuntimeError):
    pass


def writes_exec(op, args):
    if not op.startswith("s_"):
        return op.startswith("v_cmpx")
    dst = args.split(",")[0].strip()
    return dst in ("exec", "exec_lo", "exec_hi") or "saveexec" in op or "wrexec" in op


def narrows_exec(op, args):
    """The instruction ahead of an `if`'s skip branch: s_and_saveexec_b64 / s_mov_b64 exec, sN / s_and_b64 exec, ..."""
    return op.startswith("s_") and ("saveexec" in op or args.split(",")[0].strip() == "exec") and not op.startswith("s_or")


def is_scalar(op):
    return op.startswith("s_")


def scan_listing(lines):
    """`lines`: llvm-objdump -d output of one code object.  Returns [(kernel, branch address, target address, offending
    instruction text, instructions between target and restore)]."""
    found = []
    sym = None
    ins = []              # (addr, op, args) of the current symbol
    targets = []          # (branch addr, target offset in symbol)

    def finish():
        if not ins:
            return
        base = ins[0][0]
        index = {a: k for k, (a, _, _) in enumerate(ins)}
        for baddr, off in targets:
            k = index.get(base + off)
            if k is None:
                continue
            b = index[baddr]
            if not (b and narrows_exec(*ins[b - 1][1:])):
                continue                          # not the skip branch of an `if`: e.g. "every lane is done" exits, where
                                                  # exec stays 0 and what follows is dead for good
            first = None
            for n, (a, op, args) in enumerate(ins[k:k + WINDOW]):
                if writes_exec(op, args):
                    if first is not None and op == "s_or_b64" and args.replace(" ", "").startswith("exec,exec,"):
                        found.append((sym, baddr, base + off, first[1], first[0]))
                    break
                if op in ("s_endpgm", "s_branch", "s_setpc_b64") or op.startswith("s_cbranch"):
                    break                         # left the block without touching exec: nothing is restored here
                if first is None and not is_scalar(op) and op not in LANE_OPS:
                    first = (n, "%s %s" % (op, args))

    for ln in lines:
        m = _SYM.match(ln)
        if m:
            finish()
            sym, ins, targets = m.group(1), [], []
            continue
        m = _INS.match(ln)
        if not m:
            continue
        op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
        ins.append((addr, op, args))
        if op == "s_cbranch_execz":
            t = _TGT.search(ln)
            if t:
                targets.append((addr, int(t.group(2), 16)))
    finish()
    return found


def code_objects(lib, workdir):
    """The gfx950 code objects bundled in a host object / shared library, extracted into `workdir`."""
    local = os.path.join(workdir, os.path.basename(lib))
    shutil.copy(lib, local)
    subprocess.run([llvm_tool(), "--offloading", local], check=True, stdout=subprocess.DEVNULL, cwd=workdir)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn" in f and os.path.getsize(os.path.join(workdir, f)) > 0)


def scan_code_object(path):
    p = subprocess.Popen([llvm_tool(), "-d", path], stdout=subprocess.PIPE, text=True, errors="replace")
    out = scan_listing(p.stdout)
    p.wait()
    return out


def scan_library(lib, jobs=None):
    if lib.endswith((".hsaco", ".co")):
        return scan_code_object(lib)
    with tempfile.TemporaryDirectory(prefix="isa_check_") as wd:
        objs = code_objects(lib, wd)
        if not objs:
            if lib.endswith(".so"):
                raise IsaFault("no gfx950 code object found in %s" % lib)
            return []                     # a host-only translation unit
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            res = list(ex.map(scan_code_object, objs))
    return [f for r in res for f in r]


def render(found):
    try:
        from .resources import demangle
    except ImportError:                  # imported by build.py run as a script
        from resources import demangle
    names = dict(zip(sorted({f[0] for f in found}), demangle(sorted({f[0] for f in found}))))
    rows = ["%d vector instruction(s) at the target of an s_cbranch_execz ahead of the exec restore:" % len(found)]
    for sym, baddr, taddr, text, n in found:
        rows.append("  %s\n      branch at 0x%X -> 0x%X, instruction %d of the block: %s" % (names[sym], baddr, taddr, n + 1, text))
    return "\n".join(rows)


def check(lib, jobs=None):
    found = scan_library(lib, jobs)
    if found:
        msg = render(found)
        if os.environ.get("ISOCHRONES_AMD_ISA_GATE", "1") == "0":
            print("WARNING (ISOCHRONES_AMD_ISA_GATE=0): " + msg, file=sys.stderr)
        else:
            raise IsaFault(msg)
    return found


if __name__ == "__main__":
    bad = 0
    for lib in sys.argv[1:]:
        f = scan_library(lib)
        print("%s: %s" % (lib, render(f) if f else "clean (every s_cbranch_execz target restores exec before its first vector instruction)"))
        bad += len(f)
    sys.exit(1 if bad else 0)
