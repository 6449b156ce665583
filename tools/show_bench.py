#!/usr/bin/env python3
"""The figures of one bench.py JSON line, one per row.  usage: show_bench.py BENCH.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("config 2: %.3f ms/step, %.1f GCUPS on %d GPU(s); %s roof: %.0f of %.0f GB/s = %.3f (%.3f of this box's fill rate %.0f GB/s)" % (
    d["ms_per_step"], d["value"], d["n_gpus"], r["bound"], r["achieved"], r["peak"], r["frac"], r.get("frac_of_measured_fill") or 0, r.get("measured_fill_GBs_this_box") or 0))
cal = (d.get("config") or {}).get("calibration")
if cal:
    print("  launch pattern / kernel build chosen on this box: %s (calibration %.3f ms per step)" % (cal["chosen"], cal["ms_per_step"][cal["chosen"]]))
if d.get("kernel_only", {}).get("ms_per_step"):
    print("  kernel only: %.3f ms/step" % d["kernel_only"]["ms_per_step"])
if d.get("end_to_end"):
    print("  end to end: %.3f ms/step (%.1f GCUPS)" % (d["end_to_end"]["ms_per_step"], d["end_to_end"]["value"]))
s = d.get("secondary") or {}
if "c4" in s:
    c = s["c4"]
    print("config 4: %.3f s per call (first call %.3f s), search kernel %.1f ms, %d alignments created -> %.2f M/s" % (
        c["seconds"], c.get("first_call_seconds", 0), c["search_kernel_ms"], c["alignments_created"], c["value"] / 1e6))
if "c3" in s:
    print("config 3: %.3f s, %.2f GCUPS, DP kernel %.1f ms" % (s["c3"]["seconds"], s["c3"]["value"], s["c3"]["dp_kernel_ms"]))
if "c5" in s:
    print("config 5: %.3f s, %.0f GCUPS" % (s["c5"]["seconds"], s["c5"]["value"]))
c = d.get("cpu_baseline")
if c:
    print("reference on %d cores: %.5f GCUPS (%.1f s per pair); one core alone: %.1f s per pair" % (c["cores"], c["value"], c["seconds"], c.get("single_core_seconds", 0)))
