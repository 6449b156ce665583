#!/usr/bin/env python3
"""Gn2Eval's gap model (table deletions + position-dependent insertions, ALN_GAP_DEL_TABLE_INS_TPOS) on a resident batch:
n pairs of L x L fractional similarity planes over 8 distinct templates, global; the tiled exact-order kernel against the literal
O(n^3) kernel (hint exact_literal) on a subset, then one reevaluate() round with new tables (gn2.cpp:146-185).
usage: bench_gn2.py [n_pairs] [L] [n_literal]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402


def tables(rng, T):
    d = rng.uniform(3, 30, (T, T)).astype(np.float32)
    gi = rng.choice([4.0, 9.5], (T, T)).astype(np.float32)
    ge = rng.choice([0.2, 0.7], (T, T)).astype(np.float32)
    cd = np.exp(rng.uniform(-6, 1, (T, T))).astype(np.float32)
    t1, t2 = np.meshgrid(np.arange(T), np.arange(T), indexing="ij")
    D = np.zeros((T, T), dtype=np.float32)
    m = t2 >= t1 + 2
    p1, p2 = t1[m], t2[m] - 2
    near = d[p2, p1] < np.float32(18.0)
    val = (gi[p2, p1] + ge[p2, p1] * (t2[m] - t1[m] - 2).astype(np.float32)).astype(np.float32) + cd[p2, p1]
    D[m] = np.where(near, val, np.float32(8100.0)).astype(np.float32)
    return D, rng.uniform(3, 9, T).astype(np.float32), rng.uniform(0.1, 0.9, T).astype(np.float32), rng.uniform(-0.5, 1.5, T).astype(np.float32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    nlit = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.RandomState(7)
    nt = 8
    T = L + 2
    tabs = [tables(rng, T) for _ in range(nt)]
    planes = []
    for p in range(n):
        S = rng.normal(0.1, 1.2, size=(T, T)).astype(np.float32)
        S[0, :] = 0; S[-1, :] = 0; S[:, 0] = 0; S[:, -1] = 0
        planes.append(S)
    ctx = aln_amd.Context(0)
    gap = dict(tgi=np.concatenate([t[1] for t in tabs]), tge=np.concatenate([t[2] for t in tabs]), tcn=np.concatenate([t[3] for t in tabs]),
               del_tables=[t[0] for t in tabs])
    res = {}
    for name, npairs, hints in (("tiled", n, {}), ("literal", nlit, {"exact_literal": 1})):
        t_idx = np.arange(npairs) % nt
        b = aln_amd.Batch(ctx, ["A" * L] * npairs, ["A" * L] * nt, np.arange(npairs), t_idx)
        with ctx.hints(**hints):
            b.dp_simmatrix(planes[:npairs], aln_amd.GLOBAL, 0, 0, aln_amd.FWD, **gap)
            ctx.synchronize()
            t0 = time.perf_counter()
            b.reevaluate()
            ctx.synchronize()
            dt = time.perf_counter() - t0
        sc, _, st = b.optimal(want_pairs=False)
        res[name] = (b.kernel_name(), b.last_dp_ms(), npairs, sc[:nlit].copy())
        print("%s: %s, %d pairs %dx%d: DP kernel %.2f ms (%.3f ms per pair), reevaluate wall %.3f s" % (name, b.kernel_name(), npairs, L, L, b.last_dp_ms(), b.last_dp_ms() / npairs, dt))
        if name == "tiled":
            # one refinement round: new tables on the resident batch (aln_batch_set_gap), reevaluate
            tabs2 = [tables(rng, T) for _ in range(nt)]
            t0 = time.perf_counter()
            b.set_gap(aln_amd.GLOBAL, tgi=gap["tgi"], tge=gap["tge"], tcn=gap["tcn"], del_tables=[t[0] for t in tabs2])
            b.reevaluate()
            ctx.synchronize()
            print("  round with new tables (set_gap + reevaluate): %.3f s wall, DP kernel %.2f ms" % (time.perf_counter() - t0, b.last_dp_ms()))
        b.close()
    same = np.array_equal(res["tiled"][3].view(np.uint32), res["literal"][3].view(np.uint32))
    print("speed-up per pair %.0fx; scores of the first %d pairs identical: %s" % ((res["literal"][1] / res["literal"][2]) / (res["tiled"][1] / res["tiled"][2]), nlit, same))


if __name__ == "__main__":
    main()
