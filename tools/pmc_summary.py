#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into small JSON/CSV files for profiles/.
usage: pmc_summary.py OUT.json  DIR [DIR ...]      (each DIR holds *_counter_collection.csv from one --pmc pass)
       pmc_summary.py --stats OUT.csv DIR          (copies the *_kernel_stats.csv of a --kernel-trace --stats run)"""
import csv
import glob
import json
import os
import shutil
import sys


def main():
    if sys.argv[1] == "--stats":
        out, d = sys.argv[2], sys.argv[3]
        files = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)
        shutil.copyfile(files[0], out)
        return
    out = sys.argv[1]
    acc = {}
    for d in sys.argv[2:]:
        for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(fn) as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].split("(")[0]
                    if name.startswith("__amd_rocclr") or "at::native" in name:
                        continue
                    a = acc.setdefault(name, {}).setdefault(row["Counter_Name"], [0, 0.0])
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
                    acc[name].setdefault("_vgpr", int(row["VGPR_Count"]))
                    acc[name].setdefault("_lds", int(row["LDS_Block_Size"]))
    res = {}
    for k, v in acc.items():
        res[k] = {c: ({"n": a[0], "mean": a[1] / a[0]} if isinstance(a, list) else a) for c, a in v.items()}
    with open(out, "w") as fh:
        json.dump({"source": " ".join(sys.argv[2:]), "kernels": res}, fh, indent=1)


if __name__ == "__main__":
    main()
