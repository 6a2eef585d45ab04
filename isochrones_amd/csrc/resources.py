"""Register / scratch use of every kernel in libiso_hip.so, read from hipcc's -Rpass-analysis=kernel-resource-usage
remarks (csrc/build.py keeps them per translation unit), and the budget the build enforces on it.

Why a budget: a kernel that leaves the 256 architectural vector registers lives partly in accumulation registers
(AGPRs, on gfx950 the upper half of the unified file) and in spill lanes - in round 3 one sampler instantiation in that
regime made wrong accept / reject decisions that no test of the suite saw.  The build now fails
when a kernel drifts there instead of shipping it."""
from __future__ import annotations

import json
import re
import subprocess

_FIELDS = {
    "TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
    "Occupancy [waves/SIMD]": "waves", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
    "LDS Size [bytes/block]": "lds",
}
_RE = re.compile(r"remark:\s+(Function Name|" + "|".join(re.escape(k) for k in _FIELDS) + r"): (\S+)")


def is_remark_context(line: str) -> bool:
    """The source-echo lines clang prints under each remark ('   12 | {', '      | ^')."""
    return bool(re.match(r"^\s*\d*\s*\|", line))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    short = []
    for d in out[:len(names)]:
        d = d.replace("(anonymous namespace)::", "").replace("iso::fastk::", "").replace("iso::", "")
        d = re.sub(r"^void ", "", d)
        depth, cut = 0, len(d)
        for i, ch in enumerate(d):                    # cut the argument list: the first "(" outside template brackets
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        short.append(d[:cut].strip())
    return short


def parse(text: str) -> dict:
    """{demangled kernel name: {sgpr, vgpr, agpr, scratch, waves, sgpr_spill, vgpr_spill, lds}}"""
    raw, cur = {}, None
    for m in _RE.finditer(text):
        k, v = m.groups()
        if k == "Function Name":
            cur = raw.setdefault(v, {})
        elif cur is not None:
            cur[_FIELDS[k]] = int(v)
    names = list(raw)
    return dict(zip(demangle(names), (raw[n] for n in names)))


def family(name: str) -> str:
    return name.split("<", 1)[0]


#: bytes of scratch per lane a kernel family may use (default for families not listed: DEFAULT_SCRATCH).  Scratch is
#: ordinary per-lane memory - correct, only slow - so these are performance budgets; AGPRs are refused outright.
SCRATCH_BUDGET = {
    # family: bytes per lane.  A ratchet: set to what the family's worst instantiation needs today, lowered when a kernel
    # is reworked, never raised without a measurement that says the scratch is cheaper than the alternative.
    "k_lnpost_fast": 108,            # per-shape wave caps since round 4 (was 364 at the blanket 4-wave cap)
    "k_lnpost_wide": 20,
    "k_catalog_start": 64,
    "k_stretch_half": 252,           # (248 before the max-ilp scheduler of round 5: one instantiation, <1, 3, 7>, +4 B; the family is the fall-back of ensembles too large for the persistent kernels)
    "k_stretch_persist": 388,        # register-capped catalog form, triples with many bands
    "k_stretch_pair": 200,           # a single binary, one star per lane (uncapped registers)
    "k_lnpost": 96,                  # generic fallback kernel (one sample per lane since round 4: 384 -> 96)
    "k_lnpost_tree": 16,             # generic tree kernel: per-leaf values in LDS since round 5 (1 664 B of per-lane arrays before)
    "k_chain_quantiles_exact": 40,
    "k_stretch_isotrack": 24,        # 10-12 bands
    "k_mailbox_tree": 96,            # one resident wave per tree model, latency of ONE evaluation per request (four stars x seven / eight
                                     # bands: 44 / 92 B at 256 registers; every other shape none)
    "k_stretch_tree": 24,            # register-leaf forms: 20 B (five dwords of the evaluator's record, written once)
}
DEFAULT_SCRATCH = 0
MAX_AGPR = 0


def violations(table: dict, scratch_budget=None, default_scratch=None, max_agpr=None) -> list:
    sb = SCRATCH_BUDGET if scratch_budget is None else scratch_budget
    ds = DEFAULT_SCRATCH if default_scratch is None else default_scratch
    ma = MAX_AGPR if max_agpr is None else max_agpr
    bad = []
    for name, r in sorted(table.items()):
        if r.get("agpr", 0) > ma:
            bad.append("%s: %d AGPRs (limit %d)" % (name, r["agpr"], ma))
        lim = sb.get(family(name), ds)
        if r.get("scratch", 0) > lim:
            bad.append("%s: %d B/lane of scratch (budget of %s: %d)" % (name, r["scratch"], family(name), lim))
    return bad


def summary(table: dict) -> dict:
    fams = {}
    for name, r in table.items():
        f = fams.setdefault(family(name), {"kernels": 0, "max_vgpr": 0, "max_agpr": 0, "max_scratch": 0, "max_sgpr_spill": 0,
                                           "with_scratch": 0, "with_sgpr_spill": 0, "with_agpr": 0})
        f["kernels"] += 1
        f["max_vgpr"] = max(f["max_vgpr"], r.get("vgpr", 0))
        f["max_agpr"] = max(f["max_agpr"], r.get("agpr", 0))
        f["max_scratch"] = max(f["max_scratch"], r.get("scratch", 0))
        f["max_sgpr_spill"] = max(f["max_sgpr_spill"], r.get("sgpr_spill", 0))
        f["with_scratch"] += r.get("scratch", 0) > 0
        f["with_sgpr_spill"] += r.get("sgpr_spill", 0) > 0
        f["with_agpr"] += r.get("agpr", 0) > 0
    return fams


def render(table: dict) -> str:
    lines = ["%d kernels; %d with AGPRs, %d with scratch, %d with SGPR spills"
             % (len(table), sum(r.get("agpr", 0) > 0 for r in table.values()), sum(r.get("scratch", 0) > 0 for r in table.values()),
                sum(r.get("sgpr_spill", 0) > 0 for r in table.values())), "",
             "%-28s %7s %8s %8s %11s %14s %12s %14s" % ("family", "kernels", "max vgpr", "max agpr", "max scratch", "max sgpr spill",
                                                          "with scratch", "with sgpr spill")]
    for f, s in sorted(summary(table).items()):
        lines.append("%-28s %7d %8d %8d %11d %14d %12d %14d" % (f, s["kernels"], s["max_vgpr"], s["max_agpr"], s["max_scratch"],
                                                               s["max_sgpr_spill"], s["with_scratch"], s["with_sgpr_spill"]))
    lines += ["", "%-78s %5s %5s %5s %7s %5s %6s" % ("kernel", "sgpr", "vgpr", "agpr", "scratch", "waves", "sspill")]
    for name, r in sorted(table.items()):
        lines.append("%-78s %5d %5d %5d %7d %5d %6d" % (name[:78], r.get("sgpr", -1), r.get("vgpr", -1), r.get("agpr", -1),
                                                       r.get("scratch", -1), r.get("waves", -1), r.get("sgpr_spill", -1)))
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    import sys
    t = parse(open(sys.argv[1], errors="replace").read())
    sys.stdout.write(render(t))
    v = violations(t)
    if v:
        sys.stderr.write("\n".join(v) + "\n")
        sys.exit(1)
