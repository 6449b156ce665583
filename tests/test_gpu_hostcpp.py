"""-m gpu: the C++11 host mirror (alignment-algos_amd/hostcpp) end to end: reference-style code — DPMatrix<S1,S2,E>,
Optimal / Optimal_Rev, UnconstrainedNearOptimal, AlignmentSet, HMAPSequence files + Hmap2Eval — compiled with g++
against the C ABI and run on the GPU (alignment-algos_amd/host_api_test), compared with the oracle bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import orc
import refrun
from aln_amd.synth import homolog_pair, random_profile

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "alignment-algos_amd", "host_api_test")
BLOSUM = os.path.join(ROOT, "tests", "golden", "BLOSUM62")


def run(args):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "alignment-algos_amd")])
    r = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "THROW" not in r.stdout, r.stdout
    return refrun.parse(r.stdout)


def write_hmap(path, name, prof):
    """HMAP text in the format hmapalib_seq.cpp:68-117,182-243 parses; returns the profile the parser will hold."""
    n = len(prof["conf"]) - 2
    pct = np.round(prof["aa"].astype(np.float64) * 100.0, 4).astype(np.float32)
    held = {"aa": (pct / np.float32(100.0)).astype(np.float32), "sse": prof["sse"].copy(), "conf": prof["conf"].copy()}
    # the parser's head/tail elements are value-initialised (zeros), not read from the file
    for k in ("aa", "sse", "conf"):
        held[k][0] = 0
        held[k][-1] = 0
    with open(path, "w") as f:
        f.write("ID : %s\nDE : synthetic\nSR : none\nEVD: 0 0\nLEN: %d\n" % (name, n))
        for i in range(1, n + 1):
            f.write("%d A %s\n" % (i, " ".join("%.9g" % float(v) for v in pct[i])))
            f.write(" - 4.73 0.34 0 0 0 0\n")
            f.write(" * %s %.9g 0 0\n" % (" ".join("%.9g" % float(v) for v in prof["sse"][i]), float(prof["conf"][i])))
        f.write("//\n")
    return held


@pytest.mark.parametrize("mode", [1, 4])
def test_hmap2eval_through_host_classes(mode, tmp_path):
    qp, tp = random_profile(91000, 37), random_profile(92000, 52)
    qh = write_hmap(str(tmp_path / "q.hmap"), "query", qp)
    th = write_hmap(str(tmp_path / "t.hmap"), "templ", tp)
    got = run(["profile", mode, tmp_path / "q.hmap", tmp_path / "t.hmap"])
    # HMAPaliParams defaults: alpha .5, beta 1, zero_shift .12, gaps 4.73 / 0.34 (hmap_eval.cpp:4-7, alib.cpp:17-18)
    S = orc.hmap2_sim(qh, th, 0.5, 0.12)
    tgi, tge = orc.hmap2_precalc(th, 4.73, 0.34, 1.0)
    rc, D, PQ, PT = orc.dp_build(S, orc.Gap(mode, tgi=tgi, tge=tge))
    assert np.array_equal(got["S"].view(np.uint32), S.view(np.uint32))
    assert np.array_equal(got["H"].view(np.uint32), D.view(np.uint32))
    rc2, sc, pairs = orc.optimal(D, PQ, PT, False)
    a = got["sets"]["OPT"]["alis"][0]
    assert a["score"].view(np.uint32) == sc.view(np.uint32) and np.array_equal(a["pairs"], pairs)


@pytest.mark.parametrize("mode", [1, 3, 4, 0])
def test_unmodified_plugin_evaluator(mode, blosum62):
    """A plugin written against evaluator.h only (host_api_test.cpp SqrtGapEval: BLOSUM similarities, square-root gap growth,
    free end gaps per align_t; no aln_describe_gaps) runs unchanged: aln_lowering.h tabulates its deletion()/insertion()
    (ALN_GAP_TABLES).  The oracle evaluates the same gap functions through callbacks, candidate by candidate."""
    alpha, table = blosum62
    q, t = homolog_pair(96000 + mode, 37, sub_rate=0.25, indel=3)
    gi, ge = np.float32(7.5), np.float32(1.25)
    got = run(["plain", mode, float(gi), float(ge), q, t, BLOSUM])
    Q, T = len(q) + 2, len(t) + 2
    free_del, free_ins = mode in (3, 4, 2), mode in (3, 4, 0)

    def dele(q1, q2, t1, t2):
        ln = t2 - t1 - 1
        if ln < 1 or (free_del and (t1 == 0 or t2 == T - 1)):
            return 0.0
        return float(np.float32(gi + np.float32(ge * np.sqrt(np.float32(ln)))))

    def ins(q1, q2, t1, t2):
        ln = q2 - q1 - 1
        if ln < 1 or (free_ins and (q1 == 0 or q2 == Q - 1)):
            return 0.0
        return float(np.float32(gi + np.float32(ge * np.sqrt(np.float32(ln)))))

    S = orc.sim_submatrix(q, t, alpha, table)
    rc, D, PQ, PT = orc.dp_build(S, orc.Gap(mode, callbacks=(dele, ins)), islocal=(mode == 3))
    assert np.array_equal(got["H"].view(np.uint32), D.view(np.uint32))
    assert np.array_equal(got["PQ"], PQ) and np.array_equal(got["PT"], PT)
    rc2, sc, pairs = orc.optimal(D, PQ, PT, mode == 3)
    a = got["sets"]["OPT"]["alis"][0]
    assert a["score"].view(np.uint32) == sc.view(np.uint32) and np.array_equal(a["pairs"], pairs)


@pytest.mark.parametrize("mode", [1, 4, 3])
def test_gn2eval_through_host_classes(mode, tmp_path):
    """Gn2Eval (hostcpp/gn2_eval.h) lowered to a host similarity plane + ALN_GAP_DEL_TABLE_INS_TPOS: the program prints the
    tables its pre_calculate built; the oracle's restatement of the gap functions (gn2_eval.h:100-165) on those tables must
    give the same matrices.  Parity UNPINNED against the reference itself (needs Troll)."""
    qp, tp = random_profile(94000, 29), random_profile(95000, 41)
    write_hmap(str(tmp_path / "q.hmap"), "query", qp)
    write_hmap(str(tmp_path / "t.hmap"), "templ", tp)
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "alignment-algos_amd")])
    r = subprocess.run([EXE, "gn2", str(mode), str(tmp_path / "q.hmap"), str(tmp_path / "t.hmap"), "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "THROW" not in r.stdout, r.stdout[:500] + r.stderr
    got = refrun.parse(r.stdout)
    Q, T = got["dim"]
    tabs = {}
    for line in r.stdout.split("\n"):
        if line.startswith("TAB "):
            tk = line.split()
            a = np.array([int(x, 16) for x in tk[2:]], dtype=np.uint32).view(np.float32)
            tabs[tk[1]] = a.reshape(T, T) if len(a) == T * T else a
    gap = orc.Gap(mode, gn2=tabs)
    rc, D, PQ, PT = orc.dp_build(got["S"], gap, islocal=(mode == 3))
    assert np.array_equal(got["H"].view(np.uint32), D.view(np.uint32))
    assert np.array_equal(got["PQ"], PQ) and np.array_equal(got["PT"], PT)
    rc2, sc, pairs = orc.optimal(D, PQ, PT, mode == 3)
    a = got["sets"]["OPT"]["alis"][0]
    assert a["score"].view(np.uint32) == sc.view(np.uint32) and np.array_equal(a["pairs"], pairs)
    assert np.isfinite(got["S"]).all() and (got["S"][1:-1, 1:-1] != 0).any()


@pytest.mark.parametrize("mode,gi,ge,direction", [(3, 11, 1, "fwd"), (4, 4.73, 0.34, "fwd"), (1, 11, 1, "rev"), (3, 4.73, 0.34, "rev")])
def test_aa_path_through_host_classes(mode, gi, ge, direction, blosum62):
    alpha, table = blosum62
    q, t = homolog_pair(93000 + mode, 48, sub_rate=0.2, indel=3)
    got = run(["aa", mode, gi, ge, direction, q, t, BLOSUM])
    S = orc.sim_submatrix(q, t, alpha, table)
    gap = orc.Gap(mode, gi, ge)
    rc, D, PQ, PT = orc.dp_build(S, gap, orc.FWD if direction == "fwd" else orc.REV)   # B4 off by default in the host mirror
    assert np.array_equal(got["H"].view(np.uint32), D.view(np.uint32))
    assert np.array_equal(got["PQ"], PQ) and np.array_equal(got["PT"], PT)
    rc2, sc, pairs = orc.optimal(D, PQ, PT, mode == 3, kind=direction)
    a = got["sets"]["OPT"]["alis"][0]
    assert a["score"].view(np.uint32) == sc.view(np.uint32) and np.array_equal(a["pairs"], pairs)
    if direction == "fwd":
        s = orc.AliSet()
        s.push(pairs, sc)
        orc.enumerate_noa("ucw", D, PQ, PT, S, gap, None, 20, 0.3, s)
        s.identity(q, t)
        u = got["sets"]["UCW"]
        assert u["n"] == len(s)
        for k in range(len(s)):
            r = s.get(k)
            assert u["alis"][k]["score"].view(np.uint32) == r["score"].view(np.uint32)
            assert u["alis"][k]["uid"] == r["uid"]
            assert np.array_equal(u["alis"][k]["pairs"], r["pairs"])
            assert u["alis"][k]["identity"].view(np.uint32) == r["identity"].view(np.uint32)


@pytest.mark.parametrize("mode,gi,ge", [(3, 11, 1), (1, 11, 1), (4, 4.73, 0.34)])
def test_dpmatrix_set_equals_one_dpmatrix_per_pair(mode, gi, ge):
    """hostcpp/dpmatrix_set.h (an extension: many DPMatrix builds in one launch) against the one-pair DPMatrix of the reference's
    surface, pair by pair: every cell, Optimal's alignment and score, ConstrainedNearOptimal's sorted set — with
    AASubstitutionEval (codes + table) and with a plugin that names its constant-affine gap model (a SimilarityMatrix plane per
    pair), and with a plugin whose similarity depends on a value its own pre_calculate stored for the pair (the hook must run right
    before each pair is lowered, dpmatrix.h:298); an evaluator whose gap functions must be tabulated per pair is refused."""
    from aln_amd.synth import homolog_pair, random_pair
    pairs = [homolog_pair(93000, 60, sub_rate=0.2, indel=3), random_pair(93001, 30, 45), homolog_pair(93002, 130, sub_rate=0.25, indel=4),
             ("ACDEFG", "ACDFG"), homolog_pair(93003, 300, sub_rate=0.15, indel=5)]
    args = ["set", mode, gi, ge, BLOSUM]
    for q, t in pairs:
        args += [q, t]
    r = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "SET OK" in r.stdout and "mismatches 0" in r.stdout and "SET tables refused" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("mismatches 0") == 3 and "SET pairstate" in r.stdout, r.stdout


@pytest.mark.parametrize("mode", [1, 4])
def test_dpmatrix_set_of_profile_pairs(mode, tmp_path):
    """DPMatrixSet with Hmap2Eval: three profile pairs of different sizes pooled into one resident batch (similarity, z-normalisation
    and the position-minimum gap DP on the device for all of them) against one DPMatrix per pair: every cell, a similarity probe,
    Optimal's alignment."""
    args = ["setprofile", mode]
    for k, (ql, tl) in enumerate([(37, 52), (80, 61), (23, 140)]):
        write_hmap(str(tmp_path / ("q%d.hmap" % k)), "query%d" % k, random_profile(94000 + k, ql))
        write_hmap(str(tmp_path / ("t%d.hmap" % k)), "templ%d" % k, random_profile(95000 + k, tl))
        args += [tmp_path / ("q%d.hmap" % k), tmp_path / ("t%d.hmap" % k)]
    r = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "SET OK" in r.stdout and "mismatches 0" in r.stdout, r.stdout + r.stderr
