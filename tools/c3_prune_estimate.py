#!/usr/bin/env python3
"""How many far candidates of config 3's exact-order DP could a bound-based skip remove?  (measure before
building.)  Builds ONE profile pair of bench.py's config-3 workload on the GPU, fetches its score plane H and similarity S, and
replays on the host the skip rule csrc/dp_exact_blocked.hip would apply per (row block of 16, wave of 64 target columns):

  far-left deletions: source chunks of 32 columns K, nearest first.  Skip K when for ALL 16 rows r and 64 columns b
        fl(fl(max_{k in K} H[r-1][k] - g_lb(K,b)) + S[r][b])  <  fl(m(r,b) + S[r][b])
  with g_lb = min-coefficients of the chunk at its smallest distance and m = the best far-left d of the chunks processed so far;
  far insertions: source chunks of 16 rows, nearest first, same test with the column maxima of H over the 16 rows.

Prints the surviving fraction of (chunk x row x column) candidate evaluations of both scans, for the whole matrix.
usage: c3_prune_estimate.py [L] [mode] [seed_offset]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402
from aln_amd.synth import random_profile  # noqa: E402

f32 = np.float32


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else aln_amd.GLOBAL
    off = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    qp, tp = random_profile(3000 + off, L), random_profile(4000 + 7 * off, L)
    ctx = aln_amd.Context(0)
    b = aln_amd.Batch(ctx, ["A" * L], ["A" * L])
    tgi, tge = b.dp_hmap2(qp, tp, mode, 4.73, 0.34, 0.5, 1.0, 0.12)
    H, _, _ = b.get_cells(0)
    S = b.get_sim(0)
    Q, T = H.shape
    local = mode == aln_amd.LOCAL
    ninf = f32(-np.inf)
    tot_d = kept_d = tot_i = kept_i = 0
    # ---- deletions -------------------------------------------------------------------------------------------------------------
    for a0 in range(1, Q - 1, 16):
        rows = np.arange(a0, min(a0 + 16, Q - 1))                     # target rows; sources are rows-1
        src = H[rows - 1]                                             # [nr, T]
        for wv0 in range(1, T - 1, 64):
            cols = np.arange(wv0, min(wv0 + 64, T - 1))
            kbase = ((wv0 - 1) // 256) * 256                          # first near column of the tile this wave belongs to
            nchunks = kbase // 32
            if nchunks == 0:
                continue
            Sb = S[np.ix_(rows, cols)]
            m = np.full((len(rows), len(cols)), ninf, f32)
            for c in range(nchunks - 1, -1, -1):                      # nearest first
                k0, k1 = 32 * c, 32 * c + 32
                ks = np.arange(max(k0, 1), k1)
                tot_d += len(ks) * m.size
                if c < nchunks - 1:
                    gi_lb = np.minimum(tgi[ks].min(), tgi[cols])
                    ge_lb = np.minimum(tge[ks].min(), tge[cols])
                    g_lb = (gi_lb + ge_lb * (cols - (k1 - 1) - 2).astype(f32)).astype(f32)
                    ub = (src[:, ks].max(axis=1)[:, None] - g_lb[None, :]).astype(f32)
                    t_ub = (ub + Sb).astype(f32)
                    floor = (m + Sb).astype(f32)
                    if local:
                        t_ub = np.maximum(t_ub, 0); floor = np.maximum(floor, 0)
                    if (t_ub < floor).all():
                        continue
                kept_d += len(ks) * m.size
                g = (np.minimum(tgi[ks][:, None], tgi[cols][None, :]) + np.minimum(tge[ks][:, None], tge[cols][None, :]) *
                     (cols[None, :] - ks[:, None] - 2).astype(f32)).astype(f32)                   # [nk, nc]
                d = (src[:, ks][:, :, None] - g[None, :, :]).astype(f32)                         # [nr, nk, nc]
                m = np.maximum(m, d.max(axis=1))
    # ---- insertions ------------------------------------------------------------------------------------------------------------
    nblk = (Q + 15) // 16
    colmax = np.full((nblk, T), ninf, f32)
    for kb in range(nblk):
        r0, r1 = max(16 * kb, 1), min(16 * kb + 16, Q - 1)
        if r1 > r0:
            colmax[kb] = H[r0:r1].max(axis=0)
    for a0 in range(1, Q - 1, 16):
        if a0 < 3:
            continue
        rows = np.arange(a0, min(a0 + 16, Q - 1))
        for wv0 in range(1, T - 1, 64):
            cols = np.arange(max(wv0, 2), min(wv0 + 64, T - 1))
            if len(cols) == 0:
                continue
            gi = np.minimum(tgi[cols - 1], tgi[cols]); ge = np.minimum(tge[cols - 1], tge[cols])
            Sb = S[np.ix_(rows, cols)]
            m = np.full((len(rows), len(cols)), ninf, f32)
            nchunks = (a0 - 2) // 16 + 1                              # chunks kc = 0, 16, ... <= a0-2
            for c in range(nchunks - 1, -1, -1):
                k0, k1 = 16 * c, min(16 * c + 16, a0 - 1)
                ks = np.arange(max(k0, 1), k1)
                if len(ks) == 0:
                    continue
                tot_i += len(ks) * m.size
                if c < nchunks - 1:
                    dist = (rows[:, None] - (k1 - 1) - 2).astype(f32)                              # smallest distance per target row
                    g_lb = (gi[None, :] + ge[None, :] * dist).astype(f32)
                    blk_max = H[ks][:, cols - 1].max(axis=0)                                       # this lane's column maximum over the chunk
                    t_ub = ((blk_max[None, :] - g_lb).astype(f32) + Sb).astype(f32)
                    floor = (m + Sb).astype(f32)
                    if local:
                        t_ub = np.maximum(t_ub, 0); floor = np.maximum(floor, 0)
                    if (t_ub < floor).all():
                        continue
                kept_i += len(ks) * m.size
                x = H[ks][:, cols - 1]                                                             # [nk, nc]
                g = (gi[None, None, :] + ge[None, None, :] * (rows[:, None, None] - ks[None, :, None] - 2).astype(f32)).astype(f32)
                d = (x[None, :, :] - g).astype(f32)
                m = np.maximum(m, d.max(axis=1))
    print("c3 prune estimate L=%d mode=%d: far-left deletions keep %.4f of %.3e candidate evaluations; far insertions keep %.4f of %.3e"
          % (L, mode, kept_d / max(tot_d, 1), tot_d, kept_i / max(tot_i, 1), tot_i))
    b.close()
    ctx.close()


if __name__ == "__main__":
    main()
