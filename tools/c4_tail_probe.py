#!/usr/bin/env python3
"""Development aid: does the heaviest pair bound config 4's launch?  Searches the 1024-pair bench batch (twice: index order, then
heaviest first), then the heaviest pairs alone, and prints the search kernel times."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402
import bench  # noqa: E402
from aln_amd.synth import make_subopt_regions  # noqa: E402

alphabet, table = bench.load_blosum()
qs, ts = bench.make_workload(0, 1024, 2000)
ctx = aln_amd.Context(0)
flags = make_subopt_regions(2002, 10)
b = aln_amd.Batch(ctx, qs, ts)
b.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
for rep in range(3):
    b.enumerate_all("cw", 256, 0.01, flags, K=258, node_cap=1 << 18, ali_cap=1 << 16, want_pairs=False, raise_on_overflow=False)
    sm, um = b.last_enum_ms()
    print("1024 pairs, call %d: search kernel %.2f ms" % (rep, sm))
created, nodes = b.last_enum_usage()
order = np.argsort(-nodes)
print("heaviest pairs:", [(int(p), int(created[p]), int(nodes[p])) for p in order[:6]], "total nodes %.1f M, created %d" % (nodes.sum() / 1e6, created.sum()))
b.close()
for p in order[:3]:
    p = int(p)
    b1 = aln_amd.Batch(ctx, [qs[p]], [ts[p]])
    b1.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
    with ctx.hints(enum_waves=16):
        for rep in range(2):
            b1.enumerate_all("cw", 256, 0.01, flags, K=258, node_cap=1 << 26, ali_cap=1 << 17, want_pairs=False, raise_on_overflow=False)
    sm, um = b1.last_enum_ms()
    c1, n1 = b1.last_enum_usage()
    print("pair %d alone (16 waves): search kernel %.2f ms, created %d, nodes %d" % (p, sm, c1[0], n1[0]))
    b1.close()
