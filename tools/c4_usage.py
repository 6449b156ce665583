#!/usr/bin/env python3
"""What config 4 needs of the per-pair pools: n pairs of the bench workload, cw K=256, make_subopt_regions(T,10), one DELTA_RATIO.
usage: c4_usage.py [n_pairs] [delta] [node_cap_log2] [ali_cap_log2]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
sys.path.insert(0, ROOT)
import aln_amd
from aln_amd.synth import make_subopt_regions
from bench import load_blosum, make_workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
ncap = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 26)
acap = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 17)
alphabet, table = load_blosum()
qs, ts = make_workload(0, n, 2000)
ctx = aln_amd.Context(0)
b = aln_amd.Batch(ctx, qs, ts)
b.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
flags = make_subopt_regions(2002, 10)
t0 = time.perf_counter()
n_out, scores, lengths, _, status = b.enumerate_all("cw", 256, delta, flags, K=258, node_cap=ncap, ali_cap=acap, want_pairs=False, raise_on_overflow=False)
dt = time.perf_counter() - t0
a, nd = b.last_enum_usage()
sm, um = b.last_enum_ms()
print("c4 usage: %d pairs delta %g: wall %.3f s, search %.1f ms, unroll %.1f ms; overflowed %d" % (n, delta, dt, sm, um, int((status != 0).sum())))
print("alignments created: max %d, mean %.0f, per homolog pair (odd p): %s" % (a.max(), a.mean(), a[1::2][:32].tolist()))
print("trie nodes used:    max %d, mean %.0f, per homolog pair (odd p): %s" % (nd.max(), nd.mean(), nd[1::2][:32].tolist()))
