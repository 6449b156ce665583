"""CPU: the oracle's restatements of the PRUNED enumerators (KSConstrainedNearOptimal kscw.h:109-351, CRConstrainedNearOptimal
crcw.h:134-594).  Neither reference header compiles on LP64 (kscw.h:188, crcw.h:242: min(size_t, unsigned)), so no golden vector
can exist — parity UNPINNED.  What CAN be checked here without the reference are the properties every alignment either
enumerator emits must have by the source's own logic:

  * every list runs from (0,0) to (Q-1,T-1) with strictly increasing query and template positions, one of them by exactly 1 per
    step (a match state, a deletion or an insertion), never both by more than 1;
  * its score is what the path costs: the similarities of its pairs minus the evaluator's gap of every step (exact for integer
    gaps), and is above the enumerator's threshold, except for alignments that were forced along stored pointers;
  * with K_LIMIT large and nothing filtered (CRCW: MAX_OVERLAP 1, SORT_LIMIT large) the best alignment is the optimal score;
  * the sets are sorted "higher score first" and cut to NUM_SUBOPT;
  * CRCW: the reference's one out-of-bounds read (regions[-1], crcw.h:387) is reached on ordinary inputs — the oracle counts it.

The device kernels are compared with these restatements bit for bit in tests/test_gpu_enumerate.py.
"""
import numpy as np
import pytest

import orc
from aln_amd.synth import homolog_pair


def _rescore(pairs, S, gap, Q, T):
    s = np.float32(0)
    for k in range(1, len(pairs)):
        (pq, pt), (q, t) = pairs[k - 1], pairs[k]
        s = np.float32(s + S[q, t])
        if q - pq == 1:
            s = np.float32(s - orc.deletion(gap, Q, T, pq, q, pt, t))
        else:
            s = np.float32(s - orc.insertion(gap, Q, T, pq, q, pt, t))
    return s


@pytest.mark.parametrize("kind", ["kscw", "crcw"])
def test_pruned_enumerators_emit_consistent_alignments(kind, blosum62):
    alpha, table = blosum62
    oob_total = 0
    for n, ln in enumerate((24, 57, 90, 130)):
        q, t = homolog_pair(68000 + n, ln, sub_rate=0.2, indel=3)
        Q, T = len(q) + 2, len(t) + 2
        S = orc.sim_submatrix(q, t, alpha, table)
        for mode in (1, 4):
            gap = orc.Gap(mode, 11, 1)
            rc, D, PQ, PT = orc.dp_build(S, gap)
            rc2, sc, pl = orc.optimal(D, PQ, PT, False)
            flags = orc.make_subopt_regions(T, 2 + n)
            for nsub, delta, klim in ((300, 0.2, 16), (5, 0.5, 3), (300, 0.05, 64)):
                s = orc.AliSet()
                s.push(pl, sc)
                if kind == "kscw":
                    assert orc.enumerate_ks(D, PQ, PT, S, gap, flags, nsub, delta, klim, s) == 0
                else:
                    rc3, oob = orc.enumerate_cr(D, PQ, PT, S, gap, flags, nsub, delta, klim, s, sort_limit=100, max_overlap=1.0)
                    assert rc3 == 0
                    oob_total += oob
                assert 1 <= len(s) <= max(nsub, 1)
                scores = [float(s.get(k)["score"]) for k in range(len(s))]
                assert scores == sorted(scores, reverse=True)
                assert scores[0] == float(sc)                       # the optimal alignment leads the set
                for k in range(len(s)):
                    a = s.get(k)
                    p = a["pairs"]
                    assert tuple(p[0]) == (0, 0) and tuple(p[-1]) == (Q - 1, T - 1), (kind, n, mode, k)
                    dq, dt = np.diff(p[:, 0]), np.diff(p[:, 1])
                    assert (dq >= 1).all() and (dt >= 1).all() and ((dq == 1) | (dt == 1)).all(), (kind, n, mode, k)
                    assert _rescore(p, S, gap, Q, T) == a["score"], (kind, n, mode, k)
    if kind == "crcw":
        assert oob_total > 0
