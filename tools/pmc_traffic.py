#!/usr/bin/env python
"""profiles/pmc_traffic.json (the per-workload counter figures bench.py attaches to its lines) from a
tools/pmc_summarize.py summary, with the counter -> byte factors of a tools/pmc_calibrate.sh run.

    python tools/pmc_traffic.py profiles/r03/pmc_summary.json [profiles/r03/pmc_calibration.json] > profiles/pmc_traffic.json
"""
import json
import sys


def factors(cal):
    """bytes per FETCH_SIZE / WRITE_SIZE unit, measured on the probe kernel that reads whole 384-B cells with quads of
    lanes (the library kernels' gather) and on the coalesced 8-B-per-lane store (their result store)."""
    if not cal:
        return 2048.0, 1024.0, "FETCH_SIZE x 1024 x 2 (gfx950, MI355X_MICROARCH.md) + WRITE_SIZE x 1024 (uncalibrated)"
    f = next(v["bytes_per_count"] for k, v in cal["FETCH_SIZE"].items() if "cells384" in k)
    w = next(v["bytes_per_count"] for k, v in cal["WRITE_SIZE"].items() if "512 MiB" in k)
    return f, w, ("FETCH_SIZE x %.0f B (measured on 384-B-cell quad gathers with a known byte count), WRITE_SIZE x %.0f B "
                  "(measured on coalesced 8-B stores): tools/pmc_calibrate.sh" % (f, w))


def main():
    summary = json.load(open(sys.argv[1]))
    cal = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
    fb, wb, text = factors(cal)
    out = {"_comment": "per-launch PMC figures bench.py attaches to its lines (static: measured by tools/pmc_collect.sh on an "
                       "MI355X, rocprofv3 --pmc passes, one counter group per pass; the summary file has every counter). "
                       "fabric_bytes = L2-miss traffic (Infinity-Cache hits are counted by the fabric-side counters). busy "
                       "cycles: SQ_ACTIVE_INST_* x 4 / 1024 SIMDs; launch_cycles = GRBM_GUI_ACTIVE / 8 XCDs.",
           "source": sys.argv[1], "byte_factors": text}
    for e in summary:
        c, d = e.get("counters", {}), e.get("derived", {})
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        rec = {"fabric_read_bytes": c["FETCH_SIZE"]["avg"] * fb, "fabric_write_bytes": c["WRITE_SIZE"]["avg"] * wb}
        rec["fabric_bytes"] = rec["fabric_read_bytes"] + rec["fabric_write_bytes"]
        for k in ("l2_hit_rate", "launch_cycles", "effective_clock_GHz", "valu_busy_cycles_per_simd", "lds_busy_cycles_per_simd",
                  "salu_busy_cycles_per_simd", "valu_busy_fraction", "profiled_kernel_us", "valu_insts_per_wave",
                  "wave_wait_fraction"):
            if k in d:
                rec[k] = d[k]
        rec.update(kernel=e["kernel"], n=e["n"], algorithmic_bytes_per_launch=e["algorithmic_bytes_per_launch"],
                   distinct_batches=e.get("distinct_batches", 1))
        out[e["label"]] = rec
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
