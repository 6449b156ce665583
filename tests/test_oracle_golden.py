"""The oracle (oracle/aln_oracle.cpp) against the golden vectors produced by the real reference.

Pins a5-a9 (DP builders), a3/a12 (similarity, gaps), a19-a21 (tracebacks), a23/a24 (cw/ucw),
a26 (identity, sortSet) and a27 (gapped strings) of SURVEY.md §8(a).  CPU only.
"""
import numpy as np
import pytest

import goldens
import orc

DIR = {"fwd": orc.FWD, "rev": orc.REV}


def _dp(case, blosum62):
    alpha, table = blosum62
    S = orc.sim_submatrix(case["q"], case["t"], alpha, table)
    gap = orc.Gap(case["mode"], case["gi"], case["ge"])
    rc, D, PQ, PT = orc.dp_build(S, gap, DIR[case["dir"]], bug_b4=True)
    return S, gap, rc, D, PQ, PT


@pytest.mark.parametrize("prefix", ["known", "small", "enum", "mid", "c1", "c4", "aaa"])
def test_dp_optimal_enumeration(prefix, blosum62):
    cs = goldens.cases(prefix)
    assert cs
    for case in cs:
        S, gap, rc, D, PQ, PT = _dp(case, blosum62)
        if "sha" not in case:
            assert rc != 0, case["name"]
            continue
        assert rc == 0
        goldens.check_matrices(case, D, PQ, PT, S)
        islocal = case["mode"] == orc.LOCAL
        if "OPT" in case["sets"]:
            rc2, sc, pairs = orc.optimal(D, PQ, PT, islocal, kind=case["dir"])
            assert rc2 == 0
            s = orc.AliSet()
            s.push(pairs, sc)
            s.identity(case["q"], case["t"])
            got = [s.get(0)]
            tl = qls = ann = None
            if "tstr" in case["sets"]["OPT"]:
                tl, qls = s.strings(case["q"], case["t"])
                ann = [orc.annot(got[0]["score"], got[0]["identity"])]
            goldens.check_set(case, "OPT", got, tl, qls, ann)
        for key, kind in (("CW", "cw"), ("UCW", "ucw")):
            if key not in case["sets"]:
                continue
            rc2, sc, pairs = orc.optimal(D, PQ, PT, islocal)
            s = orc.AliSet()
            s.push(pairs, sc)
            flags = np.array([int(ch) for ch in case["flags"]], dtype=np.uint8) if "flags" in case else np.ones(len(case["t"]) + 2, np.uint8)
            nsub, delta = case.get("nsub", 10), case.get("delta", 0.3)
            assert orc.enumerate_noa(kind, D, PQ, PT, S, gap, flags, nsub, delta, s) == 0
            s.identity(case["q"], case["t"])
            got = [s.get(k) for k in range(len(s))]
            tl, qls = s.strings(case["q"], case["t"])
            ann = [orc.annot(g["score"], g["identity"]) for g in got]
            goldens.check_set(case, key, got, tl, qls, ann)


def test_submatrix_builds(blosum62):
    alpha, table = blosum62
    for c in goldens.subs():
        S = orc.sim_submatrix(c["q"], c["t"], alpha, table)
        gap = orc.Gap(c["mode"], c["gi"], c["ge"])
        q1, q2, t1, t2 = c["bounds"]
        rc, D, PQ, PT = orc.dp_build(S, gap, DIR[c["dir"]], bounds=(q1, q2, t1, t2), bug_b4=True)
        assert rc == 0
        goldens.check_matrices(c, D, PQ, PT)
        if "subali" in c:
            rc3, sc, pairs = orc.optimal(D, PQ, PT, False, sub=(q1, t1, q2, t2))
            assert rc3 == 0
            assert goldens.f32bits(sc) == c["subali"]["score"]
            assert pairs.reshape(-1).tolist() == c["subali"]["pairs"]


def test_known_answers(blosum62):
    """HEAGAWGHEE / PAWHEAE."""
    alpha, table = blosum62
    exp = {(3, 11, 1): (17.0, "^-HEAGAWGHEE$", "^p----AWHEAE$"),
           (1, 11, 1): (2.0, "^HEAGAWGHEE$", "^---PAWHEAE$"),
           (4, 4.73, 0.34): (21.54, "^HEAGAWGHE-E$", "^---PAW-HEAE$")}
    for (mode, gi, ge), (score, tl, ql) in exp.items():
        S = orc.sim_submatrix("PAWHEAE", "HEAGAWGHEE", alpha, table)
        rc, D, PQ, PT = orc.dp_build(S, orc.Gap(mode, gi, ge))
        rc2, sc, pairs = orc.optimal(D, PQ, PT, mode == 3)
        assert sc == np.float32(score)
        s = orc.AliSet()
        s.push(pairs, sc)
        a, b = s.strings("PAWHEAE", "HEAGAWGHEE")
        assert (a, b[0]) == (tl, ql)
    # find_max's seed cell (Q-2,T-2) wins the tie with (6,3) (optimal.h:111-121)
    S = orc.sim_submatrix("PAWHEAE", "HEAGAWGHEE", alpha, table)
    rc, D, PQ, PT = orc.dp_build(S, orc.Gap(3, 11, 1))
    assert D[6, 3] == 17 and D[7, 10] == 17
    assert orc.optimal(D, PQ, PT, True)[2][-2].tolist() == [7, 10]
