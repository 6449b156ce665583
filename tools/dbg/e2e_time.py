"""Where the end-to-end step of bench.py spends its time (host clocks around the calls, synchronised)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd")); sys.path.insert(0, ROOT)
import aln_amd
import bench
alphabet, table = bench.load_blosum()
qs, ts = bench.make_workload(0, 1024, 2000)
ctx = aln_amd.Context(0)
b = aln_amd.Batch(ctx, qs, ts)
def T(f):
    t0 = time.perf_counter(); r = f(); ctx.synchronize(); return (time.perf_counter() - t0) * 1e3, r
for rep in range(3):
    a, _ = T(lambda: b.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST))
    a2, _ = T(lambda: b.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST))
    r1, _ = T(lambda: b.reevaluate())
    s1, _ = T(lambda: b.optimal_strings(decode=False))
    a3, _ = T(lambda: b.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST))
    o1, _ = T(lambda: b.optimal(want_pairs=False))
    a4, _ = T(lambda: b.dp_submatrix(alphabet, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST))
    print("rep %d: dp %.2f, dp again %.2f, reevaluate %.2f, strings %.2f, dp after strings %.2f, optimal %.2f, dp after optimal %.2f (kernel %.2f)" % (rep, a, a2, r1, s1, a3, o1, a4, b.last_dp_ms()), flush=True)
