#!/usr/bin/env python3
"""Development aid: where an end-to-end step of config 2 spends its host time (one batch, one stream)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402
import bench  # noqa: E402

alphabet, table = bench.load_blosum()
qs, ts = bench.make_workload(0, 1024, 2000)
ctx = aln_amd.Context(0)
b = aln_amd.Batch(ctx, qs, ts)
b.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
b.optimal_strings(decode=False)
ctx.synchronize()
acc = {}


def tick(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t


N = 10
for _ in range(N):
    t = time.perf_counter()
    b.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
    t = tick("dp_submatrix returns (encode + uploads + launches)", t)
    ctx.synchronize()
    t = tick("  ... until the DP kernels are done", t)
    b.optimal_strings_enqueue()
    t = tick("strings_enqueue returns", t)
    ctx.synchronize()
    t = tick("  ... until traceback + string kernel are done", t)
    b.optimal_strings_collect(decode=False)
    t = tick("strings_collect (wait for the copy + fill the caller's buffers)", t)
for k, v in acc.items():
    print("%-70s %.3f ms" % (k, v / N * 1e3))
print("sum %.3f ms" % (sum(acc.values()) / N * 1e3))
