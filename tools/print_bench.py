"""Compact view of a bench.py JSON line on stdin: kernel times and bounds per workload."""
import json
import sys

r = json.loads(sys.stdin.read().strip().splitlines()[-1])
rf = r["roofline"]
print("cfg2 %-12s %.4f ms  value %.3g  frac(alg) %.3f  bound %s %.3f" % (r["config"]["samples"], rf["kernel_ms"], r["value"], rf["frac"],
                                                                       rf["bounds"].get("bound"), rf["bounds"].get("frac") or 0))
for k, v in r.get("other_workloads", {}).items():
    b = v["roofline"]
    print("cfg2 %-12s %.4f ms  bound %s %.3f (hbm %.3f valu %.3f)" % (k, v["kernel_ms"], b.get("bound"), b.get("frac") or 0,
                                                                    (b.get("hbm") or {}).get("frac", 0), (b.get("valu") or {}).get("frac", 0)))
for k, v in r.get("cfg3_binary_6_bands", {}).items():
    if isinstance(v, dict):
        b = v["roofline"]
        print("cfg3 %-12s %.4f ms  bound %s %.3f (hbm %.3f valu %.3f)" % (k, v["kernel_ms"], b.get("bound"), b.get("frac") or 0,
                                                                        (b.get("hbm") or {}).get("frac", 0), (b.get("valu") or {}).get("frac", 0)))
if "host_array_path" in r:
    print("host array path %.3f ms" % r["host_array_path"]["ms"])
for k, v in (r.get("catalog") or {}).items():
    if isinstance(v, dict) and "stars_per_s" in v:
        print("catalog %s: %.4f s = %.3g stars/s (first call %.3f s)" % (k, v["wall_s"], v["stars_per_s"], v.get("first_call_wall_s") or 0))
if "cfg4_mcmc_256x5000" in r:
    print("cfg4", r["cfg4_mcmc_256x5000"])
if r.get("cpu_baseline"):
    print("cpu modes", {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in r["cpu_baseline"]["modes"].items()})
