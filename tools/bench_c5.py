#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU (the driver's multi-GPU runs shard query rows): n x n synthetic proteins, length
U[400,600] (seed 5000+s), local 11/1 BLOSUM62, scores only.  Prints GCUPS = sum |q||t| / wall seconds of
aln_score_all_vs_all (upload of the residues and download of the score block included)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import aln_amd  # noqa: E402
from aln_amd.synth import MT19937, residues  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else n
    lines = open(os.path.join(ROOT, "tests", "golden", "BLOSUM62")).read().split("\n")
    k = 0
    while lines[k].startswith("#"):
        k += 1
    alphabet = "".join(lines[k].split())
    table = np.array([[float(x) for x in l.split()[1:]] for l in lines[k + 1:k + 1 + len(alphabet)]], dtype=np.float32)
    seqs = []
    for s in range(n):
        g = MT19937(5000 + s)
        ln = 400 + int(g.draw(1)[0] % 201)
        seqs.append(residues(g, ln))
    pool = aln_amd.SeqPool(seqs)
    ctx = aln_amd.Context(0)
    aln_amd.score_all_vs_all(ctx, pool, pool, alphabet, table, 11, 1, 0, min(64, rows))   # warm-up
    t0 = time.perf_counter()
    sc = aln_amd.score_all_vs_all(ctx, pool, pool, alphabet, table, 11, 1, 0, rows)
    dt = time.perf_counter() - t0
    lens = np.array([len(s) for s in seqs], dtype=np.float64)
    cells = lens[:rows].sum() * lens.sum()
    print("config5 %d x %d: %.3f s, %.1f GCUPS, checksum %.0f, self-scores ok=%s" % (
        rows, n, dt, cells / dt / 1e9, float(sc.sum()), bool((np.diag(sc[:, :rows]) >= sc[:, :rows].max(axis=1) - 1e-6).all())))


if __name__ == "__main__":
    main()
