#!/usr/bin/env python3
"""profiles/pmc_secondary.json from the r03_sec profile set (tools/bench_secondary.py under rocprofv3): VALU wave-instructions and
kernel time per call of the dominant kernels of configs 3, 4 and 5, as bench.py's secondary roofline blocks read them.
usage: make_secondary_pmc.py TAG"""
import csv
import json
import sys

tag = sys.argv[1]
summ = json.load(open("profiles/%s_pmc_summary.json" % tag))["kernels"]
stats = {r["Name"]: r for r in csv.DictReader(open("profiles/%s_kernel_stats.csv" % tag))}


def pick(sub):
    out = []
    for name, v in summ.items():
        if sub in name and "SQ_INSTS_VALU" in v:
            st = next((r for n, r in stats.items() if n.startswith(name.split("(")[0])), None)
            out.append({"symbol": name, "calls": v["SQ_INSTS_VALU"]["n"], "valu_insts_per_call": v["SQ_INSTS_VALU"]["mean"],
                        "profiled_ms_per_call": float(st["AverageNs"]) / 1e6 if st else None})
    return out


doc = {"source": "profiles/%s_pmc_summary.json, profiles/%s_kernel_stats.csv (tools/bench_secondary.py 1024 2000)" % (tag, tag),
       "c3": pick("dp_exact_tiled_kernel"), "c4": pick("enumerate_par_kernel"), "c5": pick("score_local")}
json.dump(doc, open("profiles/pmc_secondary.json", "w"), indent=1)
print(json.dumps(doc, indent=1))
