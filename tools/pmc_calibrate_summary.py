#!/usr/bin/env python
"""Join the rocprofv3 --pmc passes of tools/pmc_calibrate.sh with the probe kernels' known byte counts.

    python tools/pmc_calibrate_summary.py <dir>   ->  table on stdout + <dir>/calibration.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

EXPECT = {  # kernel substring -> (kind, bytes per launch); write8: first 6 launches 8 MB, next 6 512 MiB
    "k_stream16": ("read", 2 << 30), "k_cells384": ("read", 1_000_000 * 384), "k_lines128": ("read", 4_000_000 * 128),
    "k_write16": ("write", (32 << 20) * 16)}


def main():
    root = sys.argv[1]
    per = defaultdict(lambda: defaultdict(list))       # counter -> kernel -> [value per dispatch, dispatch order]
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(float))
        names = {}
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                did = int(row["Dispatch_Id"])
                acc[did][row["Counter_Name"]] += float(row["Counter_Value"])
                names[did] = row["Kernel_Name"]
        for did in sorted(acc):
            for c, v in acc[did].items():
                per[c][names[did].split("(")[0]].append(v)
    out = {}
    print("%-28s %-12s %10s %16s %16s %10s" % ("counter", "kernel", "launches", "avg count", "expected bytes", "B / count"))
    for c in sorted(per):
        for k, vals in sorted(per[c].items()):
            groups = [(k, vals)]
            if "k_write8" in k:
                h = len(vals) // 2
                groups = [("k_write8 (8 MB)", vals[:h]), ("k_write8 (512 MiB)", vals[h:])]
            for name, vs in groups:
                vs = vs[1:] if len(vs) > 2 else vs               # first launch of a kind: cold TLB etc.
                avg = sum(vs) / max(len(vs), 1)
                if "8 MB" in name:
                    kind, exp = "write", 1_000_000 * 8
                elif "512 MiB" in name:
                    kind, exp = "write", (64 << 20) * 8
                else:
                    kind, exp = next((v for s, v in EXPECT.items() if s in name), (None, None))
                fac = (exp / avg) if (exp and avg) else None
                print("%-28s %-12s %10d %16.1f %16s %10s" % (c, name[:12] if len(name) > 12 and "write8" not in name else name, len(vs), avg, exp, ("%.2f" % fac) if fac else "-"))
                out.setdefault(c, {})[name] = dict(launches=len(vs), avg=avg, expected_bytes=exp, kind=kind, bytes_per_count=fac)
    json.dump(out, open(os.path.join(root, "calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
