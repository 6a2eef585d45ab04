"""Build libiso_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).
The translation units are compiled in parallel and linked into one shared library."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libiso_hip.so")
OBJDIR = os.path.join(HERE, "build")
HEADERS = [os.path.join(HERE, "..", "..", "include", "isochrones_amd.h"),
           os.path.join(HERE, "iso_internal.h"), os.path.join(HERE, "iso_fast_kernel.h")] + \
    sorted(glob.glob(os.path.join(HERE, "kernels", "*.h"))) + sorted(glob.glob(os.path.join(HERE, "fast", "*.h")))
# -disable-machine-licm: MachineLICM hoists the fused evaluation's rematerialisable constants out of the persistent
# sampler's iteration loop and keeps them alive in registers (167 instead of 134 VGPRs, ~160 scalar registers parked in
# vector lanes); without it that kernel fits four waves per SIMD.  Measured neutral on every other kernel (A/B in one
# session: cfg 2 / cfg 3 batches, cfg 4 within 0.5 %).
# -Wno-bitwise-instead-of-logical: the bounds tests of the fused kernels use & and | on purpose (fast/brackets.h: a
# short-circuit turns every operand into a branch and the LDS reads behind them into a chain).
# -amdgpu-sched-strategy=max-ilp: the machine scheduler orders for instruction-level parallelism first instead of register
# pressure first.  The fused evaluation is a handful of independent dependent chains (logarithms, exponentials, brackets of
# several axes); a lone wave - a single star's fit - is bound by the latency of exactly those chains.  Measured (round 5,
# profiles/r05/ab_maxilp.jsonl): cfg 4 8.77 -> 8.51 us per step, cfg 2 73.7 -> 73.2 us, bit-identical chains.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wall",
         "-Wno-unused-function", "-Wno-bitwise-instead-of-logical", "-mllvm", "-disable-machine-licm"]
# ... for the translation units of the samplers and of the batch kernels whose registers are capped by __launch_bounds__.  A
# kernel WITHOUT a cap pays for the parallelism in registers: the batch kernels of observation trees went from 87 to 113
# registers (5 -> 4 waves per SIMD) and 134 -> 149 us per 10^6 rows, so their unit - and the generic / interpolation /
# summary kernels of iso_hip.hip, which were not measured - keep the default scheduler.
MAX_ILP_UNITS = {"iso_fast_track1", "iso_fast_iso1", "iso_fast_iso2", "iso_fast_iso3", "iso_fast_ast_track1", "iso_fast_ast_iso1",
                 "iso_fast_ast_iso2", "iso_fast_ast_iso3", "iso_fast_stretch_more", "iso_fast_stretch_tree", "iso_fast_mailbox",
                 "iso_fast_stretch_track1", "iso_fast_stretch_iso1", "iso_fast_stretch_iso2", "iso_fast_stretch_iso3"}


def unit_flags(src) -> list:
    """hipcc flags of one translation unit."""
    tu = os.path.basename(src)
    tu = tu[:-4] if tu.endswith(".hip") else tu
    return FLAGS + (["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if tu in MAX_ILP_UNITS else [])


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


STAMP = os.path.join(HERE, "libiso_hip.stamp")
#: registers / scratch of every kernel of the library (written by build(), read by tests/test_resource_gate.py)
RESOURCES = os.path.join(HERE, "libiso_hip.resources.json")


class ResourceBudgetError(RuntimeError):
    """A kernel of the library uses accumulation registers or more scratch than its family's budget (resources.py)."""


_CC_VERSION = None


def compiler_version() -> str:
    """`hipcc --version` (part of every digest: objects are not reused across toolchain upgrades)."""
    global _CC_VERSION
    if _CC_VERSION is None:
        try:
            _CC_VERSION = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, errors="replace").stdout.strip()
        except (OSError, RuntimeError):
            _CC_VERSION = "unknown"
    return _CC_VERSION


def source_digest() -> str:
    """sha256 over every source, header, the flags and this script: what the library was built from.  File times
    are not trusted for the up-to-date test (a checkout or a snapshot copy resets them); the digest stored next to
    the library is."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr(FLAGS).encode() + repr(sorted(MAX_ILP_UNITS)).encode())
    for path in sources() + HEADERS + [os.path.abspath(__file__), os.path.join(HERE, "resources.py")]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


_INCLUDE = None


def include_closure(src):
    """The source and the project headers it reaches through #include "..." lines (searched next to the including file,
    then under csrc/ and include/), in a fixed order.  System headers (<...>) belong to the toolchain, not to the digest."""
    global _INCLUDE
    import re
    if _INCLUDE is None:
        _INCLUDE = re.compile(r'^[ \t]*#[ \t]*include[ \t]*"([^"]+)"', re.M)
    roots = [HERE, os.path.join(HERE, "..", "..", "include")]
    seen, order, todo = set(), [], [os.path.abspath(src)]
    while todo:
        path = todo.pop()
        if path in seen:
            continue
        seen.add(path)
        order.append(path)
        with open(path, errors="replace") as f:
            text = f.read()
        for name in _INCLUDE.findall(text):
            for base in [os.path.dirname(path)] + roots:
                cand = os.path.abspath(os.path.join(base, name))
                if os.path.exists(cand):
                    todo.append(cand)
                    break
    return [order[0]] + sorted(order[1:])


def file_sha256(path) -> str:
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def read_stamp():
    """(digest of the sources the library was built from, sha256 of the library file written by that build), or
    (None, None).  Line 1 / line 2 of libiso_hip.stamp."""
    try:
        lines = open(STAMP).read().split()
    except OSError:
        return None, None
    return (lines[0] if lines else None), (lines[1] if len(lines) > 1 else None)


def built_digest():
    return read_stamp()[0]


def built_library_sha256():
    """sha256 of libiso_hip.so as the build that wrote the stamp produced it (compare with file_sha256(OUT): the binary
    next to the stamp is the one that build linked, not a file swapped in later)."""
    return read_stamp()[1]


def up_to_date() -> bool:
    src, so = read_stamp()
    return (os.path.exists(OUT) and os.path.exists(RESOURCES) and src == source_digest() and so is not None
            and so == file_sha256(OUT))


def resource_table() -> dict:
    """{kernel: {sgpr, vgpr, agpr, scratch, waves, sgpr_spill, ...}} of the library as built (hipcc's
    -Rpass-analysis=kernel-resource-usage remarks of every translation unit)."""
    import json
    with open(RESOURCES) as f:
        return json.load(f)


def build(force: bool = False, verbose: bool = False, gate: bool = True) -> str:
    """Compile what is out of date, link, record every kernel's register / scratch use and enforce the budget of
    resources.py on it (gate; ISOCHRONES_AMD_RESOURCE_GATE=0 turns a violation into a warning for experiments); scan every
    translation unit's generated code with isa_check.py (ISOCHRONES_AMD_ISA_GATE=0: warning only; =skip: no scan at all, for
    a toolchain without llvm-objdump)."""
    import json
    try:
        from . import resources as R
        from . import isa_check as I
    except ImportError:                  # run as a script
        sys.path.insert(0, HERE)
        import resources as R
        import isa_check as I
    digest = source_digest()
    if not force and not verbose and up_to_date():
        return OUT                       # this very file was built from exactly these sources: nothing to do
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    me = os.path.abspath(__file__)
    stale = not up_to_date()             # objects of another source state are only reused when their times say so
    jobs = []
    objs = []
    # an object is reused when the digest of what it was compiled from - its source, every header it includes (directly or
    # through another header), the flags, taken BEFORE the compiler started - is the digest of those files now (file times are
    # not trusted: an edit made while a build was running left objects that were newer than the header they had not seen).
    # Headers are followed per translation unit: an edit under kernels/ recompiles iso_hip.hip alone, not the fused families.
    import hashlib
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        h1 = hashlib.sha256((repr(unit_flags(src)) + compiler_version()).encode())
        for path in include_closure(src):
            with open(path, "rb") as f:
                h1.update(os.path.relpath(path, HERE).encode() + b"\0" + f.read())
        dig = h1.hexdigest()
        try:
            have = open(obj[:-2] + ".dig").read().strip()
        except OSError:
            have = None
        if force or have != dig or not os.path.exists(obj) or not os.path.exists(obj[:-2] + ".res"):
            jobs.append((([cc] + unit_flags(src) + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]), obj[:-2] + ".res", dig))

    def compile_one(job):
        cmd, log, dig = job
        try:
            os.remove(log[:-4] + ".dig")
        except OSError:
            pass
        p = subprocess.run(cmd, cwd=HERE, stderr=subprocess.PIPE, text=True, errors="replace")
        with open(log, "w") as f:
            f.write(p.stderr)
        if p.returncode == 0:
            with open(log[:-4] + ".dig", "w") as f:
                f.write(dig + "\n")
        # the remarks go to the log; anything else the compiler said (warnings, errors) is passed on
        rest = [ln for ln in p.stderr.splitlines() if "kernel-resource-usage" not in ln and not R.is_remark_context(ln)]
        if verbose:
            sys.stderr.write(p.stderr)
        elif rest and (p.returncode != 0 or any("warning:" in ln or "error:" in ln for ln in rest)):
            sys.stderr.write("\n".join(rest) + "\n")
        if p.returncode != 0:
            try:
                os.remove(log)
            except OSError:
                pass
        return p.returncode

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
            for rc in ex.map(compile_one, jobs):
                if rc != 0:
                    raise RuntimeError("hipcc failed")
    table = {}
    for obj in objs:
        with open(obj[:-2] + ".res", errors="replace") as f:
            table.update(R.parse(f.read()))
    bad = R.violations(table)
    if bad:
        msg = ("%d kernel(s) of libiso_hip.so outside the resource budget (isochrones_amd/csrc/resources.py):\n  " % len(bad)
               + "\n  ".join(bad[:40]) + ("\n  ..." if len(bad) > 40 else ""))
        if gate and os.environ.get("ISOCHRONES_AMD_RESOURCE_GATE", "1") != "0":
            for stale_file in (STAMP, RESOURCES):
                try:
                    os.remove(stale_file)      # never leave a stamp that calls such a library up to date
                except OSError:
                    pass
            raise ResourceBudgetError(msg)
        sys.stderr.write("WARNING: " + msg + "\n")
    # the generated code of every translation unit, checked for vector instructions ahead of an exec restore (isa_check.py:
    # the one code-generation fault this project has met); a result is kept with the digests of the object and the checker
    with open(I.__file__, "rb") as f:
        checker = hashlib.sha256(f.read()).hexdigest()

    def isa_of(obj):
        want = {"checker": checker, "object": open(obj[:-2] + ".dig").read().strip()}
        try:
            with open(obj[:-2] + ".isa") as f:
                have = json.load(f)
            if all(have.get(k) == v for k, v in want.items()):
                return have["found"]
        except (OSError, ValueError):
            pass
        want["found"] = I.scan_library(obj, jobs=1)
        with open(obj[:-2] + ".isa", "w") as f:
            json.dump(want, f)
        return want["found"]

    isa_gate = os.environ.get("ISOCHRONES_AMD_ISA_GATE", "1")
    if isa_gate == "skip":               # no llvm-objdump at hand: build without the scan (the GPU suite's closure test remains)
        sys.stderr.write("WARNING: ISOCHRONES_AMD_ISA_GATE=skip - the generated code was not scanned\n")
        faults = []
    else:
        with ThreadPoolExecutor(max_workers=os.cpu_count() or 2) as ex:
            faults = [tuple(r) for found in ex.map(isa_of, objs) for r in found]
    if faults:
        msg = I.render(faults)
        if gate and os.environ.get("ISOCHRONES_AMD_ISA_GATE", "1") != "0":
            for stale_file in (STAMP, RESOURCES):
                try:
                    os.remove(stale_file)
                except OSError:
                    pass
            raise I.IsaFault(msg)
        sys.stderr.write("WARNING: " + msg + "\n")
    if force or jobs or stale or _newer(OUT, objs):
        subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, cwd=HERE)
    with open(RESOURCES, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    with open(STAMP, "w") as f:
        f.write(digest + "\n" + file_sha256(OUT) + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
