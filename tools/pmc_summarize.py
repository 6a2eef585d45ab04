#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes over tools/pmc_workload.py into one JSON: per workload (manifest order) the mean of
every counter over its launches, plus derived figures.

    python tools/pmc_summarize.py <manifest.json> <dir-with-rocprof-output-dirs> <out.json>

Derived (MI355X_MICROARCH.md, HBM + rocprofv3 PMC sections): fabric read bytes = FETCH_SIZE [KiB... reported in units of
1 KiB] x 1024 x 2 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B; calibrated in round 1 on a known 724 MB stream), write
bytes = WRITE_SIZE x 1024; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count quad-cycles (x 4 = shader cycles);
effective clock = GRBM_GUI_ACTIVE / wall time of the launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

N_SIMD = 256 * 4
N_XCD = 8


def read_passes(root):
    """{counter: [(dispatch_id, kernel_name, value, start_ns, end_ns), ...] in dispatch order} over all passes."""
    out = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        per = defaultdict(lambda: defaultdict(float))
        meta = {}
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                did = int(row["Dispatch_Id"])
                per[did][row["Counter_Name"]] += float(row["Counter_Value"])
                meta[did] = (row["Kernel_Name"], int(row.get("Start_Timestamp", 0) or 0), int(row.get("End_Timestamp", 0) or 0))
        for did in sorted(per):
            for cname, val in per[did].items():
                out[cname].append((did, meta[did][0], val, meta[did][1], meta[did][2]))
    return out


def main():
    manifest = json.load(open(sys.argv[1]))
    data = read_passes(sys.argv[2])
    result = []
    cursor = defaultdict(int)                         # (counter, kernel substring) -> consumed dispatches
    for m in manifest:
        entry = dict(m)
        counters = {}
        for cname, rows in data.items():
            mine = [r for r in rows if m["kernel"] in r[1]]
            k0 = cursor[(cname, m["kernel"])]
            seg = mine[k0:k0 + m["launches"]]
            cursor[(cname, m["kernel"])] = k0 + m["launches"]
            seg = seg[m.get("skip", 0):]
            if not seg:
                continue
            dur = [(r[4] - r[3]) * 1e-3 for r in seg if r[4] > r[3]]
            counters[cname] = dict(launches=len(seg), avg=sum(r[2] for r in seg) / len(seg),
                                   avg_us=(sum(dur) / len(dur)) if dur else None, kernel_name=seg[0][1][:120])
        entry["counters"] = counters
        d = {}
        c = lambda k: counters[k]["avg"] if k in counters else None
        if c("FETCH_SIZE") is not None:
            d["fabric_read_bytes"] = c("FETCH_SIZE") * 1024 * 2
        if c("WRITE_SIZE") is not None:
            d["fabric_write_bytes"] = c("WRITE_SIZE") * 1024
        if "fabric_read_bytes" in d and "fabric_write_bytes" in d:
            d["fabric_bytes"] = d["fabric_read_bytes"] + d["fabric_write_bytes"]
            d["traffic_over_algorithmic"] = d["fabric_bytes"] / m["algorithmic_bytes_per_launch"]
        if c("TCC_HIT_sum") is not None and c("TCC_MISS_sum") is not None:
            d["l2_hit_rate"] = c("TCC_HIT_sum") / max(c("TCC_HIT_sum") + c("TCC_MISS_sum"), 1.0)
            d["l2_miss_bytes_128B_lines"] = c("TCC_MISS_sum") * 128
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = shader cycles of the launch; / wall time = effective clock
        cycles = None
        if c("GRBM_GUI_ACTIVE") is not None:
            cycles = c("GRBM_GUI_ACTIVE") / N_XCD
            d["launch_cycles"] = cycles
            if counters["GRBM_GUI_ACTIVE"]["avg_us"]:
                d["profiled_kernel_us"] = counters["GRBM_GUI_ACTIVE"]["avg_us"]
                d["effective_clock_GHz"] = cycles / (counters["GRBM_GUI_ACTIVE"]["avg_us"] * 1e3)
        # SQ_ACTIVE_INST_* count quad-cycles summed over all SIMDs: x 4 / 1024 SIMDs = issue-busy cycles per SIMD;
        # over the launch's cycles = the fraction of the launch during which an average SIMD issued that class
        for k, name in (("SQ_ACTIVE_INST_VALU", "valu"), ("SQ_ACTIVE_INST_LDS", "lds"), ("SQ_ACTIVE_INST_VMEM", "vmem"),
                        ("SQ_ACTIVE_INST_SCA", "salu"), ("SQ_ACTIVE_INST_ANY", "any")):
            if c(k) is not None and cycles:
                d[name + "_busy_cycles_per_simd"] = c(k) * 4 / N_SIMD
                d[name + "_busy_fraction"] = c(k) * 4 / N_SIMD / cycles
        if "fabric_bytes" in d and d.get("profiled_kernel_us"):
            d["fabric_GBs_profiled"] = d["fabric_bytes"] / d["profiled_kernel_us"] / 1e3
        if c("SQ_INSTS_VALU") is not None and c("SQ_WAVES"):
            d["valu_insts_per_wave"] = c("SQ_INSTS_VALU") / c("SQ_WAVES")
        if c("SQ_WAVE_CYCLES") is not None and c("SQ_WAIT_ANY") is not None:
            d["wave_wait_fraction"] = c("SQ_WAIT_ANY") / c("SQ_WAVE_CYCLES")
        if c("SQ_LDS_BANK_CONFLICT") is not None and c("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_fraction"] = c("SQ_LDS_BANK_CONFLICT") / c("SQ_LDS_IDX_ACTIVE")
        # memory-side request mix: write requests by size, and the mean time a read request spends outstanding at the
        # fabric interface (LEVEL = outstanding requests summed over TCC cycles; / requests = cycles per request): HBM
        # misses and Infinity-Cache hits differ in that latency, the byte counters do not separate them
        if c("TCC_EA0_WRREQ_sum") is not None and c("TCC_EA0_WRREQ_64B_sum") is not None:
            d["write_requests"] = c("TCC_EA0_WRREQ_sum")
            d["write_requests_64B"] = c("TCC_EA0_WRREQ_64B_sum")
            d["write_bytes_from_requests"] = 64.0 * c("TCC_EA0_WRREQ_64B_sum") + 32.0 * (c("TCC_EA0_WRREQ_sum") - c("TCC_EA0_WRREQ_64B_sum"))
        if c("TCC_EA0_RDREQ_sum") and c("TCC_EA0_RDREQ_LEVEL_sum") is not None:
            d["read_requests"] = c("TCC_EA0_RDREQ_sum")
            d["read_request_mean_outstanding_cycles"] = c("TCC_EA0_RDREQ_LEVEL_sum") / c("TCC_EA0_RDREQ_sum")
        entry["derived"] = d
        result.append(entry)
    json.dump(result, open(sys.argv[3], "w"), indent=1)
    for e in result:
        print(e["label"], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e["derived"].items()})


if __name__ == "__main__":
    main()
