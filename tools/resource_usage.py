"""Summarise `python isochrones_amd/csrc/build.py --force --verbose` (hipcc -Rpass-analysis=kernel-resource-usage):
one line per kernel (demangled): SGPRs, VGPRs, scratch bytes per lane, waves per SIMD.  With two logs: the kernels
whose numbers differ.   python tools/resource_usage.py new.log [old.log] [name filter]"""
import re
import subprocess
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = v
            out[cur] = {}
        elif cur:
            out[cur][k.split()[0]] = int(v)
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    short = lambda s: re.sub(r"\(.*$", "", s.replace("iso::fastk::", "").replace("void ", ""))
    return {short(d): out[n] for n, d in zip(names, dem)}


def fmt(r):
    return "sgpr %3d  vgpr %3d  scratch %4d  waves %d  sspill %3d" % (r.get("TotalSGPRs", -1), r.get("VGPRs", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("SGPRs", -1))


if __name__ == "__main__":
    args = sys.argv[1:]
    logs = [a for a in args if a.endswith(".log")]
    filt = [a for a in args if not a.endswith(".log")]
    new = parse(logs[0])
    old = parse(logs[1]) if len(logs) > 1 else None
    for name in sorted(new):
        if filt and not all(f in name for f in filt):
            continue
        if old is None:
            print("%-70s %s" % (name[:70], fmt(new[name])))
        elif name in old and old[name] != new[name]:
            print("%-70s\n    old %s\n    new %s" % (name[:100], fmt(old[name]), fmt(new[name])))
