#!/usr/bin/env python3
"""BASELINE config 4 on ONE GPU: top-K near-optimal tracebacks per pair from the GPU-resident DP matrices.
n homolog pairs of L x L (config 2's generator), local 11/1 BLOSUM62 build with the tagged kernel, then
ConstrainedNearOptimal (cw) / UnconstrainedNearOptimal (ucw) with NUM_SUBOPT=K, DELTA_RATIO, flags from
make_subopt_regions(sf, 10) (gn2.cpp:268-283).
usage: bench_c4.py [n_pairs] [L] [K] [delta] [kind] [regions]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402
from aln_amd.synth import homolog_pair, make_subopt_regions  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    delta = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
    kind = sys.argv[5] if len(sys.argv) > 5 else "cw"
    regions = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    with open(os.path.join(ROOT, "tests", "golden", "BLOSUM62")) as fh:
        rows = [ln.split() for ln in fh if ln.strip() and not ln.startswith("#")]
    alpha = "".join(rows[0])
    table = np.array([[float(x) for x in r[1:]] for r in rows[1:]], dtype=np.float32)
    pairs = [homolog_pair(1000 + p, L) for p in range(n)]
    ctx = aln_amd.Context(0)
    b = aln_amd.Batch(ctx, [p[0] for p in pairs], [p[1] for p in pairs])
    b.dp_submatrix(alpha, table, aln_amd.LOCAL, 11, 1)
    ctx.synchronize()
    flags = make_subopt_regions(L + 2, regions)
    for rep in range(2):
        t0 = time.perf_counter()
        n_out, scores, lengths, lists, status = b.enumerate_all(kind, K, delta, flags, K=K + 2, node_cap=int(os.environ.get("NODE_CAP", 1 << 19)),
                                                               ali_cap=int(os.environ.get("ALI_CAP", 1 << 16)), want_pairs=(rep == 1),
                                                               raise_on_overflow=False)
        t1 = time.perf_counter()
        sm, um = b.last_enum_ms()
        emitted = int(lengths[lengths > 0].sum())
        print("config4 %s %d pairs %dx%d K=%d delta=%.3f regions=%d pairs_out=%d: wall %.3f s, search kernel %.2f ms, unroll kernel %.2f ms; "
              "sets %d..%d (sum %d), overflowed %d, %.0f alignments/s (search+unroll), %.2f M aligned pairs emitted"
              % (kind, n, L, L, K, delta, regions, rep, t1 - t0, sm, um, n_out.min(), n_out.max(), n_out.sum(), int((status != 0).sum()),
                 n_out.sum() / ((sm + um) * 1e-3), emitted / 1e6))
    b.close()


if __name__ == "__main__":
    main()
