#!/usr/bin/env python3
"""Development aid: the long-row exact-kernel case of tests/test_gpu_dp.py with chunk skipping on and off, against the oracle;
prints where planes differ.  usage: exact_diff.py [mode gi ge]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("alignment-algos_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import aln_amd  # noqa: E402
import orc  # noqa: E402
from aln_amd.synth import homolog_pair  # noqa: E402

mode, gi, ge = (int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (1, 4.73, 0.34)
alpha, table = orc.load_blosum(os.path.join(ROOT, "tests", "golden", "BLOSUM62"))
shapes = [(70, 300), (40, 700), (350, 90), (45, 1100), (60, 2040), (33, 3000), (37, 4090)]
pairs = []
for n, (ql, tl) in enumerate(shapes):
    q, t = homolog_pair(81000 + n, max(ql, tl), sub_rate=0.3, indel=5)
    pairs.append((q[:ql], t[:tl]))
ctx = aln_amd.Context(0)
want = []
for q, t in pairs:
    S = orc.sim_submatrix(q, t, alpha, table)
    want.append(orc.dp_build(S, orc.Gap(mode, gi, ge), orc.FWD, bug_b4=True))
for prune in (0, 1):
    with ctx.hints(exact_prune=prune, exact_debug=1):
        b = aln_amd.Batch(ctx, [p[0] for p in pairs], [p[1] for p in pairs])
        b.dp_submatrix(alpha, table, mode, gi, ge, aln_amd.FWD, aln_amd.DP_AUTO if gi != int(gi) else aln_amd.DP_EXACT, bug_b4=True)
        print("prune", prune, b.kernel_name(), b.last_exact_stats())
        for p in range(len(pairs)):
            D, PQ, PT = b.get_cells(p)
            rc, D0, PQ0, PT0 = want[p]
            bad = np.argwhere(D.view(np.uint32) != D0.view(np.uint32))
            badp = np.argwhere((PQ != PQ0) | (PT != PT0))
            if len(bad) or len(badp):
                print("  pair", p, shapes[p], "score diffs", len(bad), "first", bad[:3].tolist(), "ptr diffs", len(badp), "first", badp[:3].tolist())
                for (i, j) in bad[:3]:
                    print("    cell", i, j, "got", D[i, j], "want", D0[i, j], "ptr got", PQ[i, j], PT[i, j], "want", PQ0[i, j], PT0[i, j])
        b.close()
