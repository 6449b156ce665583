#!/usr/bin/env python3
"""profiles/pmc_traffic.json (what bench.py's roofline.traffic / roofline.valu read) from profiled bench runs, one entry per
launch shape and kernel build.
usage: make_traffic.py TAG:LAUNCH_PAIRS:KERNEL_NAME [...]      e.g.  r03_b42:512:"dp_affine_tag_kernel<NW=2,R=2,X=8,local,h16,key16,occ3>"
HBM bytes per launch = WRITE_SIZE (KiB units of rocprofv3; exact for 16-B-per-lane streaming stores) + 2 x FETCH_SIZE (gfx950
reports half of a wide coalesced read: MI355X_MICROARCH.md, HBM section); VALU = SQ_INSTS_VALU wave-instructions per launch."""
import json
import sys

shapes = []
for spec in sys.argv[1:]:
    tag, pairs, kernel = spec.split(":", 2)
    summ = json.load(open("profiles/%s_pmc_summary.json" % tag))
    occ = ", 3>" if "occ3" in kernel else ", 2>"          # the template's last parameter = waves per SIMD it was compiled for
    name, k = max(((n, v) for n, v in summ["kernels"].items() if "dp_affine_tag_kernel" in n and n.endswith(occ) and "WRITE_SIZE" in v),
                  key=lambda nv: nv[1]["WRITE_SIZE"]["mean"])
    w = k["WRITE_SIZE"]["mean"] * 1024.0
    f = k["FETCH_SIZE"]["mean"] * 1024.0 * 2.0
    shapes.append({"hbm_bytes_per_launch": int(round(w + f)), "write_bytes": int(round(w)), "fetch_bytes_corrected_x2": int(round(f)),
                   "valu_insts_per_launch": k["SQ_INSTS_VALU"]["mean"], "valu_half_rate_share": 0.57,
                   "launch_pairs": int(pairs), "kernel": kernel, "profiled_kernel_symbol": name,
                   "source": "profiles/%s_pmc_summary.json" % tag})
json.dump({"shapes": shapes}, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(shapes, indent=1))
