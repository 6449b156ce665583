#!/usr/bin/env python3
"""Development aid: planes of a row-sweep variant (dp_variant_nw/r/x hints) against the default variant, ragged + full size."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402
import bench  # noqa: E402
from aln_amd.synth import homolog_pair, random_pair  # noqa: E402

nw, r, x = (int(v) for v in sys.argv[1].split(","))
alphabet, table = bench.load_blosum()
pairs = [random_pair(9000 + n, ql, tl) for n, (ql, tl) in enumerate([(5, 7), (300, 2000), (2000, 1300), (1, 2040), (700, 769), (64, 1537)])]
pairs += [homolog_pair(1001, 2000), homolog_pair(1003, 1990)]
ctx = aln_amd.Context(0)
bad = 0
for mode in (aln_amd.LOCAL, aln_amd.GLOBAL, aln_amd.SEMI_LOCAL):
    got = {}
    for name, hints in (("default", {}), ("variant", {"dp_variant_nw": nw, "dp_variant_r": r, "dp_variant_x": x})):
        with ctx.hints(**hints):
            b = aln_amd.Batch(ctx, [p[0] for p in pairs], [p[1] for p in pairs])
            b.dp_submatrix(alphabet, table, mode, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
            got[name] = ([b.get_cells(p) for p in range(len(pairs))], b.optimal(), b.kernel_name())
            b.close()
    print(mode, got["default"][2], "vs", got["variant"][2])
    for p in range(len(pairs)):
        for a, c in zip(got["default"][0][p], got["variant"][0][p]):
            if not np.array_equal(np.asarray(a).view(np.uint32), np.asarray(c).view(np.uint32)):
                bad += 1
    if not np.array_equal(got["default"][1][0].view(np.uint32), got["variant"][1][0].view(np.uint32)):
        bad += 1
print("VARIANT", sys.argv[1], "mismatches", bad)
