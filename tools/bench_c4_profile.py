#!/usr/bin/env python3
"""ConstrainedNearOptimal on resident PROFILE-PROFILE matrices (what nalign2 does after its Hmap2Eval build, nalign2.cpp:131-145):
n pairs of L x L synthetic HMAP profiles, global, position-minimum gaps 4.73/0.34, NUM_SUBOPT=256, 10 flag regions.
usage: bench_c4_profile.py [n_pairs] [L] [delta]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402
from aln_amd.synth import make_subopt_regions, random_profile  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    delta = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
    n_prof = min(n, 16)
    qps = [random_profile(3000 + p, L) for p in range(n_prof)]
    tps = [random_profile(4000 + p, L) for p in range(n_prof)]
    qpool = {k: np.concatenate([p[k] for p in qps]) for k in ("aa", "sse", "conf")}
    tpool = {k: np.concatenate([p[k] for p in tps]) for k in ("aa", "sse", "conf")}
    q_idx = np.arange(n) % n_prof
    t_idx = (np.arange(n) * 7 + np.arange(n) // n_prof) % n_prof
    ctx = aln_amd.Context(0)
    b = aln_amd.Batch(ctx, ["A" * L] * n_prof, ["A" * L] * n_prof, q_idx, t_idx)
    b.dp_hmap2(qpool, tpool, aln_amd.GLOBAL, 4.73, 0.34, 0.5, 1.0, 0.12)
    ctx.synchronize()
    flags = make_subopt_regions(L + 2, 10)
    for waves in (0, 1):
        ctx.set_hint("enum_waves", waves)
        for rep in range(2):
            t0 = time.perf_counter()
            n_out, scores, lengths, _, status = b.enumerate_all("cw", 256, delta, flags, K=258, node_cap=1 << 20, ali_cap=1 << 15, want_pairs=False,
                                                                raise_on_overflow=False)
            dt = time.perf_counter() - t0
        sm, um = b.last_enum_ms()
        created, nodes = b.last_enum_usage()
        print("profile cw, %d pairs %dx%d delta %.3f, enum_waves %d: wall %.3f s, search kernel %.1f ms; created %d..%d (sum %d), kept %d, failed %d" % (
            n, L, L, delta, waves, dt, sm, created.min(), created.max(), created.sum(), n_out.sum(), int((status != 0).sum())))
    b.close()


if __name__ == "__main__":
    main()
