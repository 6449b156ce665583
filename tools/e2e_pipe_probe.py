#!/usr/bin/env python3
"""Development aid: the pipelined end-to-end loop of bench.py (units of 1024 pairs taking turns) under a few settings; prints wall
time per step and the DP kernel's own duration, to see whether the GPU idles or the kernels slow each other down.
usage: e2e_pipe_probe.py [units] [pairs_per_unit]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402
import bench  # noqa: E402

n_units = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
alphabet, table = bench.load_blosum()
qs, ts = bench.make_workload(0, 1024, 2000)
ctxs = [aln_amd.Context(0) for _ in range(n_units)]
units = []
for k in range(n_units):
    lo = (k * per) % 1024
    b = aln_amd.Batch(ctxs[k], qs[lo:lo + per], ts[lo:lo + per])
    b.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
    b.optimal_strings(decode=False)
    units.append(b)
for occ, ap in ((2, 1), (3, 1), (2, 0), (3, 0)):
    for c in ctxs:
        c.set_hint("tag_occupancy", occ)
        c.set_hint("tag_alt_prio", ap)
    pending = [False] * n_units
    N = 12 * n_units
    host = {"collect": 0.0, "dp": 0.0, "enq": 0.0}
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for j in range(N):
        k = j % n_units
        t = time.perf_counter()
        if pending[k]:
            units[k].optimal_strings_collect(decode=False)
        host["collect"] += time.perf_counter() - t
        t = time.perf_counter()
        units[k].dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
        host["dp"] += time.perf_counter() - t
        t = time.perf_counter()
        units[k].optimal_strings_enqueue()
        host["enq"] += time.perf_counter() - t
        pending[k] = True
    for k in range(n_units):
        if pending[k]:
            units[k].optimal_strings_collect(decode=False)
    wall = (time.perf_counter() - t0) / N * 1e3
    kms = np.mean([np.mean(u.dp_ms_history(8)) for u in units])
    print("units %d x %d pairs, occ%d altprio%d: %.3f ms per launch (%.3f per 1024 pairs); DP kernel %.3f ms; host per launch: collect %.3f dp %.3f enqueue %.3f"
          % (n_units, per, occ, ap, wall, wall * 1024 / per, kms, host["collect"] / N * 1e3, host["dp"] / N * 1e3, host["enq"] / N * 1e3))
