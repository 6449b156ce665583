#!/usr/bin/env python3
"""bench.py's `secondary` block (configs 3, 4, 5 on one GPU) alone: the command the r03_sec profile set is taken on.
usage: bench_secondary.py [pairs] [length]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
length = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
alphabet, table = bench.load_blosum()
qs, ts = bench.make_workload(0, n, length)
ctx = aln_amd.Context(0)
print(json.dumps(bench.secondary_configs(aln_amd, ctx, alphabet, table, qs, ts, length)))
ctx.close()
