#!/usr/bin/env python3
"""profiles/pmc_traffic.json (what bench.py's roofline.traffic / roofline.valu read) from one profiled bench run.
usage: make_traffic.py profiles/TAG_pmc_summary.json BENCH.json [half_rate_share]
HBM bytes per launch = WRITE_SIZE (KiB units of rocprofv3; exact for 16-B-per-lane streaming stores) + 2 x FETCH_SIZE (gfx950
reports half of a wide coalesced read: MI355X_MICROARCH.md, HBM section); VALU = SQ_INSTS_VALU wave-instructions per launch."""
import json
import sys

summ = json.load(open(sys.argv[1]))
bench = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
half = float(sys.argv[3]) if len(sys.argv) > 3 else 0.57
name, k = max(((n, v) for n, v in summ["kernels"].items() if "dp_affine_tag_kernel" in n or "dp_affine_solo_kernel" in n),
              key=lambda nv: nv[1].get("WRITE_SIZE", {"mean": 0})["mean"])
w = k["WRITE_SIZE"]["mean"] * 1024.0
f = k["FETCH_SIZE"]["mean"] * 1024.0 * 2.0
out = {"hbm_bytes_per_launch": int(round(w + f)), "write_bytes": int(round(w)), "fetch_bytes_corrected_x2": int(round(f)),
       "valu_insts_per_launch": k["SQ_INSTS_VALU"]["mean"], "valu_half_rate_share": half,
       "launch_pairs": bench["config"]["launch_pairs"], "kernel": bench["config"]["kernel"], "profiled_kernel_symbol": name,
       "source": "%s (bench.py --streams %d --split %d)" % (sys.argv[1], bench["config"]["streams"], bench["config"]["launches_per_step"])}
json.dump(out, open("profiles/pmc_traffic.json", "w"))
print(out)
