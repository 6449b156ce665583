"""-m gpu parity tests of the DP build + Optimal traceback, through the C ABI, against
(a) the golden vectors produced by the real reference and (b) the oracle on seeded inputs.
Bit-exact: scores compared as uint32 bit patterns, pointers and pair lists as integers."""
import ctypes as C

import numpy as np
import pytest

import aln_amd
import goldens
import gpu_util
import orc
from aln_amd.synth import MT19937, homolog_pair, random_pair, residues

pytestmark = pytest.mark.gpu

DIRS = {"fwd": aln_amd.FWD, "rev": aln_amd.REV}


def run_group(cases, blosum62, algo=aln_amd.DP_AUTO):
    """All cases share mode/gaps/direction: one resident batch, one DP launch."""
    alpha, table = blosum62
    c0 = cases[0]
    b = aln_amd.Batch(gpu_util.ctx(), [c["q"] for c in cases], [c["t"] for c in cases])
    b.dp_submatrix(alpha, table, c0["mode"], c0["gi"], c0["ge"], DIRS[c0["dir"]], algo, bug_b4=True)
    return b


def group_key(c):
    return (c["mode"], c["gi"], c["ge"], c["dir"])


@pytest.mark.parametrize("prefix", ["known", "small", "enum", "mid", "c1", "c4"])
def test_golden_dp_and_optimal(prefix, blosum62):
    groups = {}
    for c in goldens.cases(prefix):
        groups.setdefault(group_key(c), []).append(c)
    assert groups
    for key, cases in sorted(groups.items()):
        b = run_group(cases, blosum62)
        scores, lists, status = b.optimal()
        for p, case in enumerate(cases):
            D, PQ, PT = b.get_cells(p)
            goldens.check_matrices(case, D, PQ, PT, b.get_sim(p))
            assert status[p] == 0
            if "tstr" in case["sets"]["OPT"]:
                tl, qls, idn = gpu_util.strings_for(case["q"], case["t"], [lists[p]])
                got = [{"score": scores[p], "pairs": lists[p], "identity": idn[0]}]
                ann = [orc.annot(scores[p], idn[0])]
                goldens.check_set(case, "OPT", got, tl, qls, ann)
            else:   # Optimal_Rev lists the reference itself cannot print (see oracle/ref_harness.cpp)
                goldens.check_set(case, "OPT", [{"score": scores[p], "pairs": lists[p]}])
        b.close()


def test_golden_submatrix_builds(blosum62):
    """7-argument DPMatrix ctor / build_subdpm + Optimal_Subali against the reference's matrices."""
    alpha, table = blosum62
    for c in goldens.subs():
        b = aln_amd.Batch(gpu_util.ctx(), [c["q"]], [c["t"]])
        q1, q2, t1, t2 = c["bounds"]
        b.dp_sub_submatrix(alpha, table, c["mode"], c["gi"], c["ge"], DIRS[c["dir"]], [(q1, t1, q2, t2)])
        D, PQ, PT = b.get_cells(0)
        goldens.check_matrices(c, D, PQ, PT)
        if "subali" in c:
            scores, lists, status = b.optimal(subali=True)
            assert status[0] == 0
            assert goldens.f32bits(scores[0]) == c["subali"]["score"]
            assert lists[0].reshape(-1).tolist() == c["subali"]["pairs"]
        b.close()
    # "Illegal bounds building DPM" (dpmatrix.h:360)
    b = aln_amd.Batch(gpu_util.ctx(), ["ACDEF"], ["ACDEF"])
    with pytest.raises(aln_amd.AlnError) as ei:
        b.dp_sub_submatrix(alpha, table, 1, 11, 1, aln_amd.FWD, [(3, 1, 3, 4)])
    assert ei.value.code == aln_amd.E_BOUNDS and "Illegal bounds building DPM" in str(ei.value)
    b.close()


@pytest.mark.parametrize("direction", ["fwd", "rev"])
def test_exact_kernel_vs_oracle(direction, blosum62):
    """Exact-order O(n^3) kernel: non-integer gaps (the reference's defaults 4.73/0.34), every align_t, both
    directions, ragged batch; DP_EXACT also forced on integer gaps where it must equal the fast kernel's planes."""
    alpha, table = blosum62
    rng = np.random.RandomState(7)
    lens = [(1, 1), (1, 40), (40, 1), (2, 2), (63, 65), (64, 64), (130, 171), (300, 257)]
    lens += [(int(rng.randint(1, 150)), int(rng.randint(1, 150))) for _ in range(5)]
    qs, ts = [], []
    for n, (ql, tl) in enumerate(lens):
        g = MT19937(51000 + n)
        qs.append(residues(g, ql))
        ts.append(residues(g, tl))
    d = DIRS[direction]
    od = orc.FWD if direction == "fwd" else orc.REV
    for mode in range(5):
        for (gi, ge, algo) in ((4.73, 0.34, aln_amd.DP_AUTO), (11, 1, aln_amd.DP_EXACT), (0.5, 0.25, aln_amd.DP_AUTO)):
            b = aln_amd.Batch(gpu_util.ctx(), qs, ts)
            b.dp_submatrix(alpha, table, mode, gi, ge, d, algo, bug_b4=True)
            assert "dp_exact" in b.kernel_name()
            scores, lists, status = b.optimal()
            for p, (q, t) in enumerate(zip(qs, ts)):
                S = orc.sim_submatrix(q, t, alpha, table)
                rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, gi, ge), od, bug_b4=True)
                D, PQ, PT = b.get_cells(p)
                assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (p, mode, gi, direction)
                assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (p, mode, gi, direction)
                rc2, sc, pairs = orc.optimal(D0, PQ0, PT0, mode == 3, kind=direction)
                assert status[p] == rc2 == 0
                assert np.float32(scores[p]).view(np.uint32) == sc.view(np.uint32)
                assert np.array_equal(lists[p], pairs), (p, mode, gi, direction)
            b.close()


def test_simmatrix_and_position_dependent_gaps(blosum62):
    """ALN_SIM_MATRIX (a caller-materialised SimilarityMatrix) with both gap models: fractional similarities and
    Hmap2Eval-style min(t[t1],t[t2]) gap coefficients (hmap2_eval.h:41-95) through the exact kernel, and an
    integer plane through the fast kernel."""
    rng = np.random.RandomState(3)
    dims = [(12, 9), (40, 77), (65, 64), (130, 100)]
    planes, qs, ts, tgis, tges = [], [], [], [], []
    for (Q, T) in dims:
        S = rng.uniform(-1.0, 1.5, size=(Q, T)).astype(np.float32)
        S[0, :] = 0; S[-1, :] = 0; S[:, 0] = 0; S[:, -1] = 0
        planes.append(S)
        qs.append("A" * (Q - 2)); ts.append("A" * (T - 2))
        tgis.append(rng.uniform(2.0, 6.0, size=T).astype(np.float32))
        tges.append(rng.uniform(0.1, 0.6, size=T).astype(np.float32))
    tgi_pool, tge_pool = np.concatenate(tgis), np.concatenate(tges)
    for mode in range(5):
        for model in ("const", "tpos"):
            for direction in ("fwd", "rev"):
                b = aln_amd.Batch(gpu_util.ctx(), qs, ts)
                if model == "tpos":
                    b.dp_simmatrix(planes, mode, 0, 0, DIRS[direction], tgi=tgi_pool, tge=tge_pool)
                else:
                    b.dp_simmatrix(planes, mode, 4.73, 0.34, DIRS[direction])
                scores, lists, status = b.optimal()
                for p, S in enumerate(planes):
                    gap = orc.Gap(mode, tgi=tgis[p], tge=tges[p]) if model == "tpos" else orc.Gap(mode, 4.73, 0.34)
                    rc, D0, PQ0, PT0 = orc.dp_build(S, gap, orc.FWD if direction == "fwd" else orc.REV)
                    D, PQ, PT = b.get_cells(p)
                    assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (p, mode, model, direction)
                    assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (p, mode, model, direction)
                    assert np.array_equal(b.get_sim(p).view(np.uint32), S.view(np.uint32))
                    rc2, sc, pairs = orc.optimal(D0, PQ0, PT0, mode == 3, kind=direction)
                    assert np.float32(scores[p]).view(np.uint32) == sc.view(np.uint32)
                    assert np.array_equal(lists[p], pairs)
                b.close()
    # integer-valued plane -> row-sweep kernel with the similarity read from the plane
    iplanes = [np.rint(S * 4).astype(np.float32) for S in planes]
    for mode in (1, 3):
        b = aln_amd.Batch(gpu_util.ctx(), qs, ts)
        b.dp_simmatrix(iplanes, mode, 5, 1, aln_amd.FWD, aln_amd.DP_FAST)
        assert "dp_affine_int" in b.kernel_name() and "simplane" in b.kernel_name()
        for p, S in enumerate(iplanes):
            rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, 5, 1))
            D, PQ, PT = b.get_cells(p)
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (p, mode)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (p, mode)
        b.close()


def test_error_behaviour(blosum62):
    """Error codes mirror the reference's throw sites / undefined behaviour."""
    alpha, table = blosum62
    b = aln_amd.Batch(gpu_util.ctx(), ["ACDJF"], ["ACDEF"])       # 'J' is not in the BLOSUM62 alphabet
    with pytest.raises(aln_amd.AlnError) as ei:
        b.dp_submatrix(alpha, table, 3, 11, 1)
    assert ei.value.code == aln_amd.E_RESIDUE
    with pytest.raises(aln_amd.AlnError) as ei:
        b.optimal()
    assert ei.value.code == aln_amd.E_STATE
    b.close()
    b = aln_amd.Batch(gpu_util.ctx(), ["ACD"], ["ACDEF"])
    with pytest.raises(aln_amd.AlnError) as ei:
        b.dp_submatrix(alpha, table, 7, 11, 1)                    # "Illegal gap style" (aasubalib.h:46)
    assert ei.value.code == aln_amd.E_GAPSTYLE
    with pytest.raises(aln_amd.AlnError) as ei:
        b.dp_submatrix(alpha, table, 3, 4.73, 0.34, algo=aln_amd.DP_FAST)
    assert ei.value.code == aln_amd.E_NOT_INTEGRAL
    b.close()


@pytest.mark.parametrize("tag_bits", [0, 12])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_random_batch_vs_oracle(mode, tag_bits, blosum62):
    """Ragged batch (lengths 1..520, crossing every wave/group boundary of the kernel variants) vs the oracle; with the 11-bit
    tag layout its lengths select, and with the 12-bit layout (pointer dialect 2, the instantiation of 2049..4094 residues)
    forced by the `tag_bits` hint — every align_t, every score-plane type."""
    alpha, table = blosum62
    rng = np.random.RandomState(100 + mode)
    lens = [(1, 1), (2, 300), (300, 2), (254, 254), (255, 257), (256, 256), (510, 130), (130, 511), (260, 519)]
    lens += [(int(rng.randint(1, 400)), int(rng.randint(1, 400))) for _ in range(6)]
    qs, ts = [], []
    for n, (ql, tl) in enumerate(lens):
        g = MT19937(31000 + 97 * mode + n)
        if n % 3 == 2 and ql > 20:
            q, t = homolog_pair(32000 + n, ql)
            t = (t * (tl // len(t) + 1))[:tl]
        else:
            q, t = residues(g, ql), residues(g, tl)
        qs.append(q)
        ts.append(t)
    # local builds have three plane layouts (uint16 scores + the 16-bit key layout, uint16 scores with 13-/14-bit keys, fp32 scores)
    layouts = ({}, {"key16": 0}, {"h16": 0, "key16": 0}) if mode == 3 else ({},)
    for (gi, ge) in ((11, 1), (3, 0), (0, 2)):
        ref = []
        for q, t in zip(qs, ts):
            S = orc.sim_submatrix(q, t, alpha, table)
            rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, gi, ge))
            ref.append((D0, PQ0, PT0) + tuple(orc.optimal(D0, PQ0, PT0, mode == 3)))
        for layout in layouts:
            with gpu_util.ctx().hints(tag_bits=tag_bits, **layout):
                b = aln_amd.Batch(gpu_util.ctx(), qs, ts)
                b.dp_submatrix(alpha, table, mode, gi, ge, aln_amd.FWD, aln_amd.DP_FAST)
            kn = b.kernel_name()
            assert "dp_affine_tag" in kn and ("tag12" in kn) == (tag_bits == 12), kn
            if mode == 3:
                assert ("key16" in kn) == (layout.get("key16", 1) == 1) and ("h16" in kn) == (layout.get("h16", 1) == 1), kn
            scores, lists, status = b.optimal()
            for p in range(len(qs)):
                D0, PQ0, PT0, rc2, sc, pairs = ref[p]
                D, PQ, PT = b.get_cells(p)
                assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (p, mode, gi, ge, kn)
                assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (p, mode, gi, ge, kn)
                assert status[p] == rc2 == 0
                assert np.float32(scores[p]).view(np.uint32) == sc.view(np.uint32)
                assert np.array_equal(lists[p], pairs)
            b.close()


@pytest.mark.parametrize("kernel", ["tag", "int"])
@pytest.mark.parametrize("variant", ["1,2", "1,8", "2,1", "2,4", "4,1", "4,2", "8,1", "1,4,8", "2,1,8", "2,2,8", "4,1,8"])
def test_kernel_variants_agree(variant, kernel, blosum62):
    """Every (waves per pair, groups per lane) instantiation of both row-sweep kernels (tagged keys, Q,T <= 2048;
    plain int32 with explicit arg-max, up to 8192) gives the oracle's planes."""
    alpha, table = blosum62
    nw, r, xc = ([int(x) for x in variant.split(",")] + [4])[:3]
    hints = {"dp_variant_nw": nw, "dp_variant_r": r, "dp_variant_x": xc, "tag_kernel": 0 if kernel == "int" else 1}
    if kernel == "int" and xc != 4:
        pytest.skip("8 columns per lane and group exist in the tagged kernel only")
    cap = 64 * xc * nw * r - 2
    qs, ts = [], []
    for n, tl in enumerate([cap, cap - 1, cap - 255, max(cap - 300, 5), 7]):
        if n % 2:
            q, t = homolog_pair(41000 + n, min(tl, 120))
            t = (t * (tl // len(t) + 1))[:tl]
        else:
            q, t = random_pair(41000 + n, 90, tl)
        qs.append(q)
        ts.append(t)
    for mode in (3, 1):
        b = aln_amd.Batch(gpu_util.ctx(), qs, ts)
        with gpu_util.ctx().hints(**hints):
            b.dp_submatrix(alpha, table, mode, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
        assert "NW=%d,R=%d" % (nw, r) in b.kernel_name() and ("dp_affine_%s" % kernel) in b.kernel_name()
        assert ("X=8" in b.kernel_name()) == (xc == 8)
        for p, (q, t) in enumerate(zip(qs, ts)):
            S = orc.sim_submatrix(q, t, alpha, table)
            rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, 11, 1))
            D, PQ, PT = b.get_cells(p)
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (variant, p, mode)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (variant, p, mode)
        b.close()


def test_full_size_properties(blosum62):
    """BASELINE.json config 2 sizes (2000 x 2000), where the O(n^3) oracle is too slow for a unit test:
    size-independent properties of the local DP — the best score is symmetric under swapping the two
    sequences, a self-alignment scores the sum of its diagonal, the reported path re-scores to the
    reported score, and the SURVEY App. C known answer (seed 12345, n=2000, local 11/1 -> 50, 20 pairs)."""
    alpha, table = blosum62
    idx = {ch: k for k, ch in enumerate(alpha)}
    pr = [random_pair(12345, 2000), homolog_pair(1001, 2000)]
    qs = [pr[0][0], pr[1][0], pr[0][1], pr[1][1], pr[0][0]]
    ts = [pr[0][1], pr[1][1], pr[0][0], pr[1][0], pr[0][0]]
    b = aln_amd.Batch(gpu_util.ctx(), qs, ts)
    b.dp_submatrix(alpha, table, aln_amd.LOCAL, 11, 1)
    assert "dp_affine_tag" in b.kernel_name()
    scores, lists, status = b.optimal()
    assert (status == 0).all()
    assert scores[0] == 50.0 and len(lists[0]) == 20
    assert scores[0] == scores[2] and scores[1] == scores[3]
    assert scores[4] == sum(table[idx[c], idx[c]] for c in qs[4])
    # the self-alignment reaches row 1, whose pointer is the origin: enumerate_local then stops with q_last == 0 and
    # does NOT prepend (0,0) (optimal.h:96-104) -> (1,1) .. (2000,2000) + the tail pair
    assert len(lists[4]) == 2001 and (lists[4][:, 0] == lists[4][:, 1]).all() and lists[4][0, 0] == 1
    # re-score every path: sum of similarities minus affine gap costs == reported score (local: interior pairs only)
    for p in range(4):
        pairs = lists[p][:-1]           # drop the tail pair and, when present, (0,0)
        if tuple(pairs[0]) == (0, 0):
            pairs = pairs[1:]
        q, t = "^" + qs[p] + "$", "^" + ts[p] + "$"
        s = 0.0
        for k, (i, j) in enumerate(pairs):
            s += table[idx[q[i]], idx[t[j]]]
            if k:
                di, dj = i - pairs[k - 1][0], j - pairs[k - 1][1]
                gap = max(di, dj) - 1
                assert min(di, dj) == 1
                if gap:
                    s -= 11 + (gap - 1)
        assert s == scores[p], (p, s, scores[p])
    b.close()


@pytest.mark.parametrize("direction", ["fwd", "rev"])
def test_blocked_exact_kernel_long_rows(direction, blosum62):
    """dp_exact_blocked (max-first scans, chunk re-walks, 16-row far-insertion window) on templates that need 2, 4 and 8
    column slots per thread and several row blocks: integer gaps forced through it (many exact ties -> the "first k"
    and rounding-tie paths), fractional gaps, and profile-style fractional planes with min(t[t1],t[t2]) gaps."""
    alpha, table = blosum62
    d = DIRS[direction]
    od = orc.FWD if direction == "fwd" else orc.REV
    shapes = [(70, 300), (40, 700), (350, 90), (45, 1100), (60, 2040), (33, 3000), (37, 4090)]
    pairs = []
    for n, (ql, tl) in enumerate(shapes):
        q, t = homolog_pair(81000 + n, max(ql, tl), sub_rate=0.3, indel=5)
        pairs.append((q[:ql], t[:tl]))
    for mode, gi, ge, algo in ((3, 11, 1, aln_amd.DP_EXACT), (1, 4.73, 0.34, aln_amd.DP_AUTO), (4, 1, 1, aln_amd.DP_EXACT)):
        b = aln_amd.Batch(gpu_util.ctx(), [p[0] for p in pairs], [p[1] for p in pairs])
        b.dp_submatrix(alpha, table, mode, gi, ge, d, algo, bug_b4=True)
        assert "dp_exact_blocked" in b.kernel_name() or "dp_exact_tiled" in b.kernel_name()
        for p, (q, t) in enumerate(pairs):
            S = orc.sim_submatrix(q, t, alpha, table)
            rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, gi, ge), od, bug_b4=True)
            D, PQ, PT = b.get_cells(p)
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (p, mode, gi, direction)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (p, mode, gi, direction)
        b.close()
    # fractional planes + position-dependent gaps
    rng = np.random.RandomState(11)
    dims = [(50, 600), (300, 1500), (37, 2050), (35, 3300)]
    planes, tgis, tges = [], [], []
    for (Q, T) in dims:
        S = rng.normal(-0.12, 1.0, size=(Q, T)).astype(np.float32)
        S[0, :] = 0; S[-1, :] = 0; S[:, 0] = 0; S[:, -1] = 0
        planes.append(S)
        pi = np.exp(rng.uniform(-0.25, 1.0, size=T)).astype(np.float32)
        tgis.append((np.float32(4.73) * pi).astype(np.float32))
        tges.append((np.float32(0.34) * pi).astype(np.float32))
    for mode in (1, 3):
        b = aln_amd.Batch(gpu_util.ctx(), ["A" * (Q - 2) for Q, T in dims], ["A" * (T - 2) for Q, T in dims])
        b.dp_simmatrix(planes, mode, 0, 0, d, tgi=np.concatenate(tgis), tge=np.concatenate(tges))
        assert "dp_exact_blocked" in b.kernel_name() or "dp_exact_tiled" in b.kernel_name()
        for p, S in enumerate(planes):
            rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, tgi=tgis[p], tge=tges[p]), od)
            D, PQ, PT = b.get_cells(p)
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (p, mode, direction)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (p, mode, direction)
        b.close()


@pytest.mark.parametrize("direction", ["fwd", "rev"])
def test_exact_chunk_skipping_is_invisible(direction, blosum62):
    """dp_exact_tiled_kernel skips far chunks of candidates that provably cannot matter (nearest chunk first, block maxima as
    bounds, a rounding margin — csrc/dp_exact_blocked.hip).  On pairs with dozens of row blocks and several column tiles: the
    planes with the skipping on equal the planes with it off AND the oracle's, bit for bit, pointers included — fractional
    constant gaps and integer gaps (ties everywhere) over BLOSUM scores, position-minimum gaps over fractional planes; global,
    local, semi-local; the counters show that chunks really were skipped.  Both forms of the kernel: the wavefront of 64-column tiles
    (no workgroup barrier, the default) and the 256-column tiles with a barrier per row."""
    alpha, table = blosum62
    d = DIRS[direction]
    od = orc.FWD if direction == "fwd" else orc.REV
    ctx = gpu_util.ctx()
    FORMS = ((1, 1), (0, 1), (1, 0))          # (skipping, wavefront form): the default, the same without skipping, the 256-column form
    shapes = [(620, 900), (1100, 530), (300, 1400)]
    pairs = []
    for n, (ql, tl) in enumerate(shapes):
        q, t = homolog_pair(83000 + n, max(ql, tl), sub_rate=0.3, indel=6)
        pairs.append((q[:ql], t[-tl:]))
    for mode, gi, ge, algo in ((1, 4.73, 0.34, aln_amd.DP_AUTO), (3, 11, 1, aln_amd.DP_EXACT), (4, 2.5, 0.25, aln_amd.DP_AUTO)):
        got = {}
        for prune, wf in FORMS:
            with ctx.hints(exact_prune=prune, exact_debug=1, exact_wavefront=wf):
                b = aln_amd.Batch(ctx, [p[0] for p in pairs], [p[1] for p in pairs])
                b.dp_submatrix(alpha, table, mode, gi, ge, d, algo)
                assert "dp_exact_tiled" in b.kernel_name() and ("wavefront" in b.kernel_name()) == bool(wf), b.kernel_name()
                got[(prune, wf)] = [b.get_cells(p) for p in range(len(pairs))]
                st = b.last_exact_stats()
                if prune:
                    assert st[1] > 0 and st[3] > 0 and st[1] <= st[0] and st[3] <= st[2], st      # chunks really were skipped
                b.close()
        got[1], got[0] = got[(1, 1)], got[(0, 1)]
        for p, (q, t) in enumerate(pairs):
            for other in FORMS[1:]:
                for x, y in zip(got[1][p], got[other][p]):
                    assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)), (mode, p, direction, other)
            S = orc.sim_submatrix(q, t, alpha, table)
            rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, gi, ge), od)
            D, PQ, PT = got[1][p]
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (mode, p, direction)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (mode, p, direction)
    # position-minimum gaps over fractional planes (the config-3 model)
    rng = np.random.RandomState(13)
    dims = [(700, 1000), (420, 1500)]
    planes, tgis, tges = [], [], []
    for (Q, T) in dims:
        S = rng.normal(-0.12, 1.0, size=(Q, T)).astype(np.float32)
        S[0, :] = 0; S[-1, :] = 0; S[:, 0] = 0; S[:, -1] = 0
        planes.append(S)
        pi = np.exp(rng.uniform(-0.25, 1.0, size=T)).astype(np.float32)
        tgis.append((np.float32(4.73) * pi).astype(np.float32))
        tges.append((np.float32(0.34) * pi).astype(np.float32))
    for mode in (1, 3, 4):
        got = {}
        for prune, wf in FORMS:
            with ctx.hints(exact_prune=prune, exact_debug=1, exact_wavefront=wf):
                b = aln_amd.Batch(ctx, ["A" * (Q - 2) for Q, T in dims], ["A" * (T - 2) for Q, T in dims])
                b.dp_simmatrix(planes, mode, 0, 0, d, tgi=np.concatenate(tgis), tge=np.concatenate(tges))
                assert "dp_exact_tiled" in b.kernel_name() and ("wavefront" in b.kernel_name()) == bool(wf), b.kernel_name()
                got[(prune, wf)] = [b.get_cells(p) for p in range(len(dims))]
                st = b.last_exact_stats()
                if prune:
                    assert st[1] > 0 and st[3] > 0 and st[1] <= st[0] and st[3] <= st[2], st
                else:
                    assert st[0] == 0 and st[2] == 0, st               # nothing is tested when the skipping is off
                b.close()
        got[1] = got[(1, 1)]
        for p, S in enumerate(planes):
            for other in FORMS[1:]:
                for x, y in zip(got[1][p], got[other][p]):
                    assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)), (mode, p, direction, other)
            rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, tgi=tgis[p], tge=tges[p]), od)
            D, PQ, PT = got[1][p]
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (mode, p, direction)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (mode, p, direction)


def test_full_size_three_kernels_agree(blosum62):
    """BASELINE config-2/3 sizes (2000 x 2000): the tagged O(n^2) kernel, the int O(n^2) kernel and the exact-order O(n^3)
    kernel are three independent programmes for the same recurrence; on integer gaps their score and pointer planes must
    be identical cell for cell (the O(n^3) oracle is too slow here; it pins each of them at smaller sizes)."""
    alpha, table = blosum62
    pr = [homolog_pair(1001, 2000), random_pair(1002, 2000), homolog_pair(1003, 1990)]
    qs, ts = [p[0] for p in pr], [p[1] for p in pr]
    for mode in (aln_amd.LOCAL, aln_amd.GLOBAL):
        planes = {}
        for name, algo, hints in (("tag", aln_amd.DP_FAST, {}), ("int", aln_amd.DP_FAST, {"tag_kernel": 0}), ("exact", aln_amd.DP_EXACT, {})):
            with gpu_util.ctx().hints(**hints):
                b = aln_amd.Batch(gpu_util.ctx(), qs, ts)
                b.dp_submatrix(alpha, table, mode, 11, 1, aln_amd.FWD, algo)
                kn = b.kernel_name()
                assert {"tag": "dp_affine_tag", "int": "dp_affine_int", "exact": "dp_exact_"}[name] in kn, kn
                planes[name] = [b.get_cells(p) for p in range(len(pr))]
                sc, lists, status = b.optimal()
                planes[name + "_opt"] = (sc, lists)
                b.close()
        for other in ("int", "exact"):
            for p in range(len(pr)):
                for a, c in zip(planes["tag"][p], planes[other][p]):
                    assert np.array_equal(np.asarray(a).view(np.uint32), np.asarray(c).view(np.uint32)), (mode, other, p)
            assert np.array_equal(planes["tag_opt"][0].view(np.uint32), planes[other + "_opt"][0].view(np.uint32))
            for p in range(len(pr)):
                assert np.array_equal(planes["tag_opt"][1][p], planes[other + "_opt"][1][p])


def gn2_tables(rng, T):
    """Synthetic Gn2Eval::pre_calculate outputs (gn2_eval.cpp:113-158) for a template of T positions: v_gi/v_ge/v_cn and the
    (p2,p1)-indexed distance / vv_gi / vv_ge / vv_cd tables.  Real inputs come from the Troll library (absent)."""
    t = {"v_gi": rng.uniform(3, 9, T), "v_ge": rng.uniform(0.1, 0.9, T), "v_cn": rng.uniform(-0.5, 1.5, T),
         "dist": rng.uniform(3, 30, (T, T)), "vv_gi": rng.choice([4.0, 9.5], (T, T)), "vv_ge": rng.choice([0.2, 0.7], (T, T)),
         "vv_cd": np.exp(rng.uniform(-6, 1, (T, T)))}
    return {k: v.astype(np.float32) for k, v in t.items()}


def gn2_deletion_table(tb, T, mode):
    """What a host lowering of Gn2Eval materialises: deletion(t1,t2) for every t1 < t2 (gn2_eval.h:100-130), fp32."""
    D = np.zeros((T, T), dtype=np.float32)
    for t1 in range(T):
        for t2 in range(t1 + 2, T):
            p1, p2 = t1, t2 - 2
            gp = np.float32(8100.0)
            if tb["dist"][p2, p1] < np.float32(18.0):
                gp = np.float32(np.float32(tb["vv_gi"][p2, p1] + np.float32(tb["vv_ge"][p2, p1] * np.float32(t2 - t1 - 2))) + tb["vv_cd"][p2, p1])
            if mode in (3, 4, 2) and (t1 == 0 or t2 == T - 1):
                gp = np.float32(0.0)
            D[t1, t2] = gp
    return D


@pytest.mark.parametrize("direction", ["fwd", "rev"])
def test_gn2_table_gap_model(direction):
    """ALN_GAP_DEL_TABLE_INS_TPOS (Gn2Eval's gap functions): table deletions, (gi[t1] + ge[t1] (di-2)) + cn[t1] insertions, all
    five end-gap styles, DP + Optimal + constrained enumeration vs the oracle's restatement.  Parity UNPINNED against the
    reference itself (Gn2Eval cannot be built without Troll); this pins kernel == oracle."""
    rng = np.random.RandomState(5)
    dims = [(9, 14), (40, 33), (66, 90), (120, 300), (300, 530), (3, 258), (20, 2)]      # up to three 256-column tiles, 19 row blocks
    planes, tabs = [], []
    for (Q, T) in dims:
        S = rng.normal(0.1, 1.2, size=(Q, T)).astype(np.float32)
        S[0, :] = 0; S[-1, :] = 0; S[:, 0] = 0; S[:, -1] = 0
        planes.append(S)
        tabs.append(gn2_tables(rng, T))
    pool = {k: np.concatenate([tb[k] for tb in tabs]) for k in ("v_gi", "v_ge", "v_cn")}
    for mode in range(5):
        b = aln_amd.Batch(gpu_util.ctx(), ["A" * (Q - 2) for Q, T in dims], ["A" * (T - 2) for Q, T in dims])
        dels = [gn2_deletion_table(tb, T, mode) for tb, (Q, T) in zip(tabs, dims)]
        b.dp_simmatrix(planes, mode, 0, 0, DIRS[direction], tgi=pool["v_gi"], tge=pool["v_ge"], tcn=pool["v_cn"], del_tables=dels)
        assert "dp_exact_tiled_kernel<gn2tab" in b.kernel_name(), b.kernel_name()
        scores, lists, status = b.optimal()
        for p, S in enumerate(planes):
            gap = orc.Gap(mode, gn2=tabs[p])
            rc, D0, PQ0, PT0 = orc.dp_build(S, gap, orc.FWD if direction == "fwd" else orc.REV)
            D, PQ, PT = b.get_cells(p)
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (p, mode, direction)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (p, mode, direction)
            rc2, sc, pairs = orc.optimal(D0, PQ0, PT0, mode == 3, kind=direction)
            assert np.float32(scores[p]).view(np.uint32) == sc.view(np.uint32)
            assert np.array_equal(lists[p], pairs)
            if direction == "fwd" and mode in (1, 4):
                T = dims[p][1]
                flags = orc.make_subopt_regions(T, 3)
                s = orc.AliSet()
                s.push(pairs, sc)
                orc.enumerate_noa("cw", D0, PQ0, PT0, S, gap, flags, 12, 0.1, s)
                got = b.enumerate(p, "cw", 12, 0.1, flags, max_alignments=max(12, len(s)) + 2)
                assert len(got) == len(s)
                for k, g in enumerate(got):
                    r = s.get(k)
                    assert np.float32(g["score"]).view(np.uint32) == r["score"].view(np.uint32), (p, mode, k)
                    assert np.array_equal(g["pairs"], r["pairs"])
        b.close()


def test_edge_sizes(blosum62):
    """Empty batch, empty sequences (only '^$'), 1-residue sequences against the longest a kernel takes, and the exact
    maximum sizes of the tagged kernel (2048 x 2048 including sentinels) — against the oracle where it is fast enough,
    otherwise against the other kernels."""
    alpha, table = blosum62
    # empty batch: every call is a no-op
    b = aln_amd.Batch(gpu_util.ctx(), [], [])
    b.dp_submatrix(alpha, table, 3, 11, 1)
    sc, lists, status = b.optimal()
    assert len(sc) == 0 and len(lists) == 0
    b.close()
    # empty and tiny sequences in one ragged batch, every align_t, both gap kinds
    g = MT19937(424242)
    long_t = residues(g, 2046)
    qs = ["", "", "A", "W", long_t[:5], ""]
    ts = ["", "ACD", "", long_t, "W", long_t]
    for mode in range(5):
        for gi, ge in ((11, 1), (4.73, 0.34)):
            b = aln_amd.Batch(gpu_util.ctx(), qs, ts)
            b.dp_submatrix(alpha, table, mode, gi, ge)
            scores, lists, status = b.optimal()
            for p, (q, t) in enumerate(zip(qs, ts)):
                S = orc.sim_submatrix(q, t, alpha, table)
                rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, gi, ge))
                D, PQ, PT = b.get_cells(p)
                assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (mode, gi, p)
                assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (mode, gi, p)
                rc2, sc, pairs = orc.optimal(D0, PQ0, PT0, mode == 3)
                assert status[p] == rc2, (mode, gi, p, status[p], rc2)
                if rc2 == 0:
                    assert np.float32(scores[p]).view(np.uint32) == sc.view(np.uint32) and np.array_equal(lists[p], pairs)
            b.close()
    # the tagged kernel's largest matrix: 2046 residues each
    q, t = homolog_pair(77001, 2046)
    planes = {}
    for name, env, algo in (("tag", None, aln_amd.DP_FAST), ("exact", None, aln_amd.DP_EXACT)):
        b = aln_amd.Batch(gpu_util.ctx(), [q], [t])
        b.dp_submatrix(alpha, table, 3, 11, 1, aln_amd.FWD, algo)
        planes[name] = (b.kernel_name(), b.get_cells(0), b.optimal())
        b.close()
    assert "dp_affine_tag" in planes["tag"][0] and "dp_exact_tiled" in planes["exact"][0]
    for a, c in zip(planes["tag"][1], planes["exact"][1]):
        assert np.array_equal(np.asarray(a).view(np.uint32), np.asarray(c).view(np.uint32))
    assert np.array_equal(planes["tag"][2][1][0], planes["exact"][2][1][0])
    # one residue more needs 12 tag bits (pointer dialect 2, four waves of 1024 columns): same planes as the exact kernel
    q2, t2 = q + "A", t + "C"
    res = []
    for algo in (aln_amd.DP_FAST, aln_amd.DP_EXACT):
        b = aln_amd.Batch(gpu_util.ctx(), [q2], [t2])
        b.dp_submatrix(alpha, table, 3, 11, 1, aln_amd.FWD, algo)
        res.append((b.kernel_name(), b.get_cells(0)))
        b.close()
    assert "dp_affine_tag" in res[0][0] and "tag12" in res[0][0], res[0][0]
    for a, c in zip(res[0][1], res[1][1]):
        assert np.array_equal(np.asarray(a).view(np.uint32), np.asarray(c).view(np.uint32))
    # 4094 residues is the tagged kernel's largest matrix (4096 x 4096 with the sentinels); one more -> the int kernel, global too
    q3, t3 = homolog_pair(77002, 4095)
    for mode in (3, 1):
        kn = []
        cells = []
        for qq, tt in ((q3[:4094], t3[:4094]), (q3, t3[:4000])):
            b = aln_amd.Batch(gpu_util.ctx(), [qq], [tt])
            b.dp_submatrix(alpha, table, mode, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
            kn.append(b.kernel_name())
            cells.append(b.get_cells(0))
            b.close()
        assert "tag12" in kn[0] and "dp_affine_int" in kn[1], kn
        # the common part of the two matrices (same residues, same recurrence) is identical wherever no path can reach beyond it:
        # row by row the first 4001 columns of the 4094-row build equal the int kernel's 4095-row build up to its last interior row
        D0, PQ0, PT0 = cells[0]
        D1, PQ1, PT1 = cells[1]
        assert np.array_equal(D0[:4094, :4000].view(np.uint32), D1[:4094, :4000].view(np.uint32)), mode
        assert np.array_equal(PQ0[:4094, :4000], PQ1[:4094, :4000]) and np.array_equal(PT0[:4094, :4000], PT1[:4094, :4000]), mode


def tabulate_gaps(gap, Q, T):
    """What aln_lowering.h does for an arbitrary evaluator, here with the oracle's gap functions as "the evaluator":
    deletion(t1,t2) for every t1 < t2, insertion for interior / head / tail query positions."""
    L = orc.lib()
    err = C.c_int(0)
    D = np.zeros((T, T), dtype=np.float32)
    for t1 in range(T):
        for t2 in range(t1 + 1, T):
            D[t1, t2] = L.orc_deletion(gap.ref, Q, T, 1, 2, t1, t2, C.byref(err))
    I = np.zeros((3, T, Q), dtype=np.float32)
    for t1 in range(T - 1):
        for d in range(1, Q - 2):
            I[0, t1, d] = L.orc_insertion(gap.ref, Q, T, 1, 1 + d, t1, t1 + 1, C.byref(err))
        for q2 in range(1, Q):
            I[1, t1, q2] = L.orc_insertion(gap.ref, Q, T, 0, q2, t1, t1 + 1, C.byref(err))
        for q1 in range(0, Q - 1):
            I[2, t1, q1] = L.orc_insertion(gap.ref, Q, T, q1, Q - 1, t1, t1 + 1, C.byref(err))
    assert err.value == 0
    return D, I


@pytest.mark.parametrize("direction", ["fwd", "rev"])
def test_tabulated_gap_model(direction):
    """ALN_GAP_TABLES: the evaluator's deletion()/insertion() fully materialised (what the host mirror does for a plugin that
    does not name a closed-form model).  Tables filled from the oracle's min(t[t1],t[t2]) functions must reproduce the
    oracle's DP for that model bit for bit, in every end-gap style, plus Optimal and a constrained enumeration."""
    rng = np.random.RandomState(21)
    dims = [(7, 11), (30, 41), (50, 36)]
    planes, tgis, tges = [], [], []
    for (Q, T) in dims:
        S = rng.normal(0.0, 1.3, size=(Q, T)).astype(np.float32)
        S[0, :] = 0; S[-1, :] = 0; S[:, 0] = 0; S[:, -1] = 0
        planes.append(S)
        tgis.append(rng.uniform(2.0, 6.0, size=T).astype(np.float32))
        tges.append(rng.uniform(0.1, 0.6, size=T).astype(np.float32))
    for mode in range(5):
        gaps = [orc.Gap(mode, tgi=tgis[p], tge=tges[p]) for p in range(len(dims))]
        tabs = [tabulate_gaps(gaps[p], *dims[p]) for p in range(len(dims))]
        b = aln_amd.Batch(gpu_util.ctx(), ["A" * (Q - 2) for Q, T in dims], ["A" * (T - 2) for Q, T in dims])
        b.dp_simmatrix(planes, mode, 0, 0, DIRS[direction], del_tables=[t[0] for t in tabs], ins_tables=[t[1] for t in tabs])
        assert "dp_exact_kernel" in b.kernel_name()
        scores, lists, status = b.optimal()
        for p, S in enumerate(planes):
            rc, D0, PQ0, PT0 = orc.dp_build(S, gaps[p], orc.FWD if direction == "fwd" else orc.REV)
            D, PQ, PT = b.get_cells(p)
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (p, mode, direction)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (p, mode, direction)
            rc2, sc, pairs = orc.optimal(D0, PQ0, PT0, mode == 3, kind=direction)
            assert np.float32(scores[p]).view(np.uint32) == sc.view(np.uint32) and np.array_equal(lists[p], pairs)
            if direction == "fwd" and mode in (1, 3):
                T = dims[p][1]
                flags = orc.make_subopt_regions(T, 4)
                s = orc.AliSet()
                s.push(pairs, sc)
                orc.enumerate_noa("cw", D0, PQ0, PT0, S, gaps[p], flags, 10, 0.1, s)
                got = b.enumerate(p, "cw", 10, 0.1, flags, max_alignments=max(10, len(s)) + 2)
                assert len(got) == len(s)
                for k, g in enumerate(got):
                    r = s.get(k)
                    assert np.float32(g["score"]).view(np.uint32) == r["score"].view(np.uint32), (p, mode, k)
                    assert np.array_equal(g["pairs"], r["pairs"])
        b.close()


def test_batched_subrectangle_fills(blosum62):
    """Many small sub-matrix builds in ONE launch (what the reference's SSSS loop fill does one DPMatrix at a time,
    ssss.h:621-631): 40 rectangles over a few sequence pairs, both directions, + Optimal_Subali, against the oracle."""
    alpha, table = blosum62
    rng = np.random.RandomState(77)
    base = [homolog_pair(88000 + n, 120, sub_rate=0.25, indel=4) for n in range(4)]
    qs, ts, q_idx, t_idx, bounds = [p[0] for p in base], [p[1] for p in base], [], [], []
    for k in range(40):
        p = k % 4
        Q, T = len(qs[p]) + 2, len(ts[p]) + 2
        q1 = int(rng.randint(0, Q - 3)); q2 = int(rng.randint(q1 + 1, min(Q - 1, q1 + 25) + 1))
        t1 = int(rng.randint(0, T - 3)); t2 = int(rng.randint(t1 + 1, min(T - 1, t1 + 25) + 1))
        q_idx.append(p); t_idx.append(p); bounds.append((q1, t1, q2, t2))
    for direction in ("fwd", "rev"):
        for mode, gi, ge in ((1, 11, 1), (1, 4.73, 0.34), (4, 11, 1)):
            b = aln_amd.Batch(gpu_util.ctx(), qs, ts, q_idx, t_idx)
            b.dp_sub_submatrix(alpha, table, mode, gi, ge, DIRS[direction], bounds)
            if direction == "fwd":
                scores, lists, status = b.optimal(subali=True)
            for k, (q1, t1, q2, t2) in enumerate(bounds):
                p = q_idx[k]
                S = orc.sim_submatrix(qs[p], ts[p], alpha, table)
                rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, gi, ge), orc.FWD if direction == "fwd" else orc.REV,
                                                bounds=(q1, q2, t1, t2))
                D, PQ, PT = b.get_cells(k)
                assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (direction, mode, gi, k)
                assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (direction, mode, gi, k)
                if direction == "fwd":
                    rc2, sc, pairs = orc.optimal(D0, PQ0, PT0, False, sub=(q1, t1, q2, t2))
                    assert status[k] == rc2
                    if rc2 == 0:
                        assert np.float32(scores[k]).view(np.uint32) == sc.view(np.uint32) and np.array_equal(lists[k], pairs)
            b.close()


def test_optimal_enqueue_collect(blosum62):
    """aln_batch_optimal_enqueue / _collect (what bench.py pipelines) give the scores, list lengths and status of
    aln_batch_optimal; two slots, FIFO, a third enqueue is a state error; aln_batch_dp_ms_history reports every build."""
    alpha, table = blosum62
    pairs = [homolog_pair(99000 + n, ln) for n, ln in enumerate((30, 200, 513))] + [("", "ACD")]
    for mode in (3, 1):
        b = aln_amd.Batch(gpu_util.ctx(), [p[0] for p in pairs], [p[1] for p in pairs])
        b.dp_submatrix(alpha, table, mode, 11, 1)
        sc0, lists0, st0 = b.optimal()
        b.optimal_enqueue()
        b.reevaluate()
        b.optimal_enqueue()
        with pytest.raises(aln_amd.AlnError) as ei:
            b.optimal_enqueue()
        assert ei.value.code == aln_amd.E_STATE
        for _ in range(2):
            sc, cnt, st = b.optimal_collect()
            assert np.array_equal(sc.view(np.uint32), sc0.view(np.uint32)) and np.array_equal(st, st0)
            assert cnt.tolist() == [len(x) for x in lists0]
        with pytest.raises(aln_amd.AlnError):
            b.optimal_collect()
        ms = b.dp_ms_history(8)
        assert len(ms) == 2 and (ms > 0).all()
        b.close()


def test_row_alternating_priority_is_invisible(blosum62):
    """The context hint "tag_alt_prio" (s_setprio alternating per row, a scheduling hint of the tagged kernel) changes no cell."""
    alpha, table = blosum62
    pairs = [homolog_pair(77000 + n, ln) for n, ln in enumerate((1500, 1100, 700))] + [random_pair(77100, 1990, 1800)]
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    planes = []
    for flag in (0, 1):
        b = aln_amd.Batch(gpu_util.ctx(), qs, ts)
        with gpu_util.ctx().hints(tag_alt_prio=flag):
            b.dp_submatrix(alpha, table, 3, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
        assert "dp_affine_tag" in b.kernel_name() and "NW=2" in b.kernel_name()
        sc, lists, st = b.optimal()
        planes.append([b.get_cells(p) for p in range(len(qs))] + [sc, lists])
        b.close()
    for p in range(len(qs)):
        for x, y in zip(planes[0][p], planes[1][p]):
            assert np.array_equal(x, y)
    assert np.array_equal(planes[0][-2], planes[1][-2])
    assert all(np.array_equal(a, b) for a, b in zip(planes[0][-1], planes[1][-1]))
    # and against the oracle for one of them
    S = orc.sim_submatrix(qs[2], ts[2], alpha, table)
    rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(3, 11, 1))
    D, PQ, PT = planes[1][2]
    assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)) and np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0)


def test_two_contexts_on_two_streams_overlap(blosum62):
    """bench.py alternates its steps over two resident batches, each with its own context and HIP stream, so that their
    launches overlap on the GPU: both give the results of a lone batch, step after step."""
    alpha, table = blosum62
    pairs = [homolog_pair(78000 + n, 1200 + 100 * (n % 5)) for n in range(24)]
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    lone = aln_amd.Batch(gpu_util.ctx(), qs, ts)
    lone.dp_submatrix(alpha, table, 3, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
    sc0, lists0, st0 = lone.optimal()
    cells0 = lone.get_cells(5)
    lone.close()
    ctxs = [aln_amd.Context(0), aln_amd.Context(0)]        # each context creates its own non-blocking HIP stream
    bs = [aln_amd.Batch(c, qs, ts) for c in ctxs]
    for b in bs:
        b.dp_submatrix(alpha, table, 3, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
    for step in range(6):
        b = bs[step % 2]
        if step >= 2:
            sc, cnt, st = b.optimal_collect()
            assert np.array_equal(sc.view(np.uint32), sc0.view(np.uint32)) and np.array_equal(st, st0)
            assert cnt.tolist() == [len(x) for x in lists0]
        b.reevaluate()
        b.optimal_enqueue()
    for b in bs:
        sc, cnt, st = b.optimal_collect()
        assert np.array_equal(sc.view(np.uint32), sc0.view(np.uint32))
        for x, y in zip(b.get_cells(5), cells0):
            assert np.array_equal(x, y)
        b.close()
    for c in ctxs:
        c.close()


def test_optimal_strings_equal_the_reference(blosum62):
    """aln_batch_optimal_strings (traceback + pair lists to the host + SequenceGaps + calcIdentity for every pair of a batch)
    against the reference's OPT sets: template line, query line, identity, score."""
    alpha, table = blosum62
    for mode, gi, ge in ((3, 11, 1), (1, 11, 1), (4, 4.73, 0.34)):
        cs = [c for c in goldens.cases() if c["dir"] == "fwd" and c["mode"] == mode and c["gi"] == gi and "OPT" in c.get("sets", {})
              and "tstr" in c["sets"]["OPT"] and "throw" not in c and c["name"].split("_")[0] not in ("aaa",)][:40]
        assert len(cs) >= 5
        b = aln_amd.Batch(gpu_util.ctx(), [c["q"] for c in cs], [c["t"] for c in cs])
        b.dp_submatrix(alpha, table, mode, gi, ge)
        scores, ident, status, tl, ql = b.optimal_strings()
        for k, c in enumerate(cs):
            ref = c["sets"]["OPT"]
            assert status[k] == 0, c["name"]
            assert goldens.f32bits(scores[k]) == ref["alis"][0]["score"], c["name"]
            assert goldens.f32bits(ident[k]) == ref["alis"][0]["identity"], c["name"]
            assert tl[k] == ref["tstr"] and ql[k] == ref["alis"][0]["qstr"], c["name"]
        b.close()


@pytest.mark.parametrize("direction", ["fwd", "rev"])
def test_device_strings_equal_the_host_renderer(direction, blosum62):
    """csrc/gapped_strings.hip (one wave per pair lays out both lines and counts identities on the device) against the host
    renderer aln_gapped_strings / aln_identity (pinned by every golden set) fed with the same batch's pair lists: all five align
    types, both build directions (reverse local lists may be unprintable: empty lines on both routes), ragged sizes from empty
    sequences to pairs with end jumps of hundreds of columns (written wave-wide), and the enqueue / collect form with both slots
    in flight."""
    alpha, table = blosum62
    pairs = [random_pair(61000 + n, ql, tl) for n, (ql, tl) in enumerate([(0, 0), (1, 1), (1, 40), (33, 2), (64, 64), (150, 700), (700, 130), (257, 255)])]
    pairs += [homolog_pair(61100 + n, ln, sub_rate=0.2, indel=6) for n, ln in enumerate((40, 130, 420, 900))]
    pairs += [("ACDEFGHIKL", "ACDEFGHIKL"), ("AAAAAAAAAAAAAAAAAAAAWWWWWWWWWWWW", "WWWWWWWWWWWWCCCCCCCCCCCCCCCCCCCCCCCCCCCCCC")]
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    ctx = gpu_util.ctx()
    for mode, gi, ge in ((3, 11, 1), (1, 11, 1), (0, 11, 1), (2, 11, 1), (4, 4.73, 0.34)):
        b = aln_amd.Batch(ctx, qs, ts)
        b.dp_submatrix(alpha, table, mode, gi, ge, DIRS[direction])
        sc0, lists, st0 = b.optimal()
        scores, ident, status, tl, ql = b.optimal_strings()
        b.optimal_strings_enqueue()                                    # both slots in flight, collected oldest first
        b.optimal_strings_enqueue()
        again = [b.optimal_strings_collect(), b.optimal_strings_collect()]
        printed = 0
        for k, (q, t) in enumerate(pairs):
            assert status[k] == st0[k] and np.float32(scores[k]).view(np.uint32) == np.float32(sc0[k]).view(np.uint32), (mode, k)
            if st0[k] != 0:
                assert tl[k] == "" and ql[k] == "" and ident[k] == 0, (mode, k)
                continue
            pl = lists[k]
            Q, T = len(q) + 2, len(t) + 2
            printable = len(pl) > 0 and tuple(pl[-1]) == (Q - 1, T - 1) and all(tuple(pl[i]) != tuple(pl[i - 1]) for i in range(1, len(pl)))
            assert np.float32(ident[k]).view(np.uint32) == gpu_util.identity_for(q, t, pl).view(np.uint32), (mode, k)
            if not printable:
                assert tl[k] == "" and ql[k] == "", (mode, k)
                continue
            want_t, want_q, _ = gpu_util.strings_for(q, t, [pl])
            assert tl[k] == want_t and ql[k] == want_q[0], (mode, direction, k, tl[k], want_t)
            printed += 1
        assert printed >= (len(pairs) - 2 if direction == "fwd" else 1)
        for got in again:
            assert np.array_equal(got[0].view(np.uint32), scores.view(np.uint32)) and np.array_equal(got[1].view(np.uint32), ident.view(np.uint32))
            assert got[3] == tl and got[4] == ql
        # call order: nothing to collect, a third enqueue while two slots wait, the one-call form while a slot waits
        with pytest.raises(aln_amd.AlnError) as ei:
            b.optimal_strings_collect()
        assert ei.value.code == aln_amd.E_STATE
        b.optimal_strings_enqueue()
        b.optimal_strings_enqueue()
        for call in (b.optimal_strings_enqueue, b.optimal_strings):
            with pytest.raises(aln_amd.AlnError) as ei:
                call()
            assert ei.value.code == aln_amd.E_STATE
        b.optimal_strings_collect(); b.optimal_strings_collect()
        b.close()


@pytest.mark.parametrize("variant", [(2, 2, 8), (4, 1, 4), (8, 1, 4), (2, 1, 8)])
def test_skewed_exchange_lags_agree(variant, blosum62):
    """The context hint "tag_lag" (how many rows wave w of a pair runs behind wave w-1; 0 = the synchronous per-row exchange)
    changes no cell: every lag against the oracle on a ragged batch that includes pairs shorter than the lag and pairs whose
    last columns leave the later waves empty."""
    alpha, table = blosum62
    nw, r, x = variant
    cap = 64 * x * nw * r - 2
    qs, ts = [], []
    for n, (ql, tl) in enumerate([(90, cap), (3, cap - 1), (1, 40), (37, cap - 255), (64, 64 * x * r + 1), (5, 7), (0, 0), (130, cap // 2)]):
        q, t = random_pair(43000 + n, ql, tl)
        if n == 0:
            t = (q * (tl // max(len(q), 1) + 1))[:tl]         # repeats of the query: long diagonals crossing the wave boundaries
        qs.append(q)
        ts.append(t)
    want = {}
    for mode in (3, 4):
        for p, (q, t) in enumerate(zip(qs, ts)):
            S = orc.sim_submatrix(q, t, alpha, table)
            want[(mode, p)] = orc.dp_build(S, orc.Gap(mode, 11, 1))
    ctx = gpu_util.ctx()
    for lag in (0, 1, 2, 4):
        for mode in (3, 4):
            b = aln_amd.Batch(ctx, qs, ts)
            with ctx.hints(dp_variant_nw=nw, dp_variant_r=r, dp_variant_x=x, tag_lag=lag):
                b.dp_submatrix(alpha, table, mode, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
            assert "dp_affine_tag" in b.kernel_name() and "NW=%d,R=%d" % (nw, r) in b.kernel_name()
            for p in range(len(qs)):
                rc, D0, PQ0, PT0 = want[(mode, p)]
                D, PQ, PT = b.get_cells(p)
                assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (variant, lag, mode, p)
                assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (variant, lag, mode, p)
            b.close()


def test_segment_queue_is_invisible(blosum62):
    """The segment queue of the tagged kernel (context hint "tag_segments": a pair's rows are built by up to six workgroups that
    take (pair, segment) items from a device queue and hand the row state on through HBM): ragged batch with pairs below and
    above the 512-row threshold, local and global, 1-, 2- and 4-wave instantiations, repeated launches on the same batch (the
    queue is reset by every launch) — planes, scores and paths equal the one-workgroup-per-pair launch, which the oracle and
    the reference goldens pin; one mid-size pair is compared with the oracle directly."""
    alpha, table = blosum62
    lens = [(2000, 1990), (511, 700), (512, 640), (513, 100), (1300, 2046), (3, 3), (700, 700), (1999, 64), (640, 1500)]
    pairs = [homolog_pair(79000 + n, max(q, t))for n, (q, t) in enumerate(lens)]
    qs = [p[0][:q] for p, (q, t) in zip(pairs, lens)]
    ts = [p[1][:t] for p, (q, t) in zip(pairs, lens)]
    ctx = gpu_util.ctx()
    S = orc.sim_submatrix(qs[6], ts[6], alpha, table)
    for mode in (3, 1):
        rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, 11, 1))
        for variant in ({}, {"dp_variant_nw": 1, "dp_variant_r": 4, "dp_variant_x": 8}, {"dp_variant_nw": 4, "dp_variant_r": 2, "dp_variant_x": 4}):
            res = {}
            for segq in (0, -6, -3):
                with ctx.hints(tag_segments=segq, **variant):
                    b = aln_amd.Batch(ctx, qs, ts)
                    b.dp_submatrix(alpha, table, mode, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
                    assert b.kernel_name().endswith("+segq") == (segq != 0), b.kernel_name()
                    for rep in range(3):
                        b.reevaluate()
                    sc, lists, st = b.optimal()
                    res[segq] = ([b.get_cells(p) for p in range(len(qs))], sc, lists, st)
                    b.close()
            a = res[0]
            for ks in (-6, -3):
                c = res[ks]
                assert (a[3] == 0).all() and (c[3] == 0).all()
                assert np.array_equal(a[1].view(np.uint32), c[1].view(np.uint32))
                for p in range(len(qs)):
                    for x, y in zip(a[0][p], c[0][p]):
                        assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)), (mode, variant, ks, p)
                    assert np.array_equal(a[2][p], c[2][p])
            D, PQ, PT = c[0][6]
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)) and np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0)


def test_gn2_rounds_set_gap_then_reevaluate():
    """The refinement rounds of gn2.cpp:146-185 on a resident batch: aln_batch_set_gap swaps the gap tables (similarity planes stay
    in HBM), aln_batch_reevaluate rebuilds.  Three rounds, forward and reverse, global and local: every round equals a fresh batch
    built with that round's tables; the tiled exact kernel equals the literal O(n^3) kernel on 900 x 1100 pairs (where the
    oracle is too slow), which in turn is oracle-checked by test_gn2_table_gap_model; constant and position gaps can be
    swapped in the same way."""
    rng = np.random.RandomState(11)
    dims = [(900, 1100), (310, 700), (64, 64)]
    planes = []
    for (Q, T) in dims:
        S = rng.normal(0.1, 1.2, size=(Q, T)).astype(np.float32)
        S[0, :] = 0; S[-1, :] = 0; S[:, 0] = 0; S[:, -1] = 0
        planes.append(S)
    rounds = [[gn2_tables(rng, T) for (Q, T) in dims] for _ in range(3)]
    ctx = gpu_util.ctx()

    def args(tabs, mode):
        pool = {k: np.concatenate([tb[k] for tb in tabs]) for k in ("v_gi", "v_ge", "v_cn")}
        dels = [gn2_deletion_table(tb, T, mode) for tb, (Q, T) in zip(tabs, dims)]
        return dict(tgi=pool["v_gi"], tge=pool["v_ge"], tcn=pool["v_cn"], del_tables=dels)

    for direction in ("fwd", "rev"):
        for mode in (1, 3):
            b = aln_amd.Batch(ctx, ["A" * (Q - 2) for Q, T in dims], ["A" * (T - 2) for Q, T in dims])
            b.dp_simmatrix(planes, mode, 0, 0, DIRS[direction], **args(rounds[0], mode))
            for r in range(3):
                if r > 0:
                    b.set_gap(mode, 0, 0, **args(rounds[r], mode))
                    b.reevaluate()
                assert "dp_exact_tiled_kernel<gn2tab" in b.kernel_name()
                with ctx.hints(exact_literal=1):
                    f = aln_amd.Batch(ctx, ["A" * (Q - 2) for Q, T in dims], ["A" * (T - 2) for Q, T in dims])
                    f.dp_simmatrix(planes, mode, 0, 0, DIRS[direction], **args(rounds[r], mode))
                    assert f.kernel_name().startswith("dp_exact_kernel"), f.kernel_name()
                for p in range(len(dims)):
                    for x, y in zip(b.get_cells(p), f.get_cells(p)):
                        assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)), (direction, mode, r, p)
                sb, lb, stb = b.optimal()
                sf, lf, stf = f.optimal()
                assert np.array_equal(sb.view(np.uint32), sf.view(np.uint32)) and all(np.array_equal(x, y) for x, y in zip(lb, lf))
                f.close()
            # the same call swaps in any other gap model: constant gaps on the resident planes == a fresh build
            b.set_gap(mode, 4.73, 0.34)
            b.reevaluate()
            f = aln_amd.Batch(ctx, ["A" * (Q - 2) for Q, T in dims], ["A" * (T - 2) for Q, T in dims])
            f.dp_simmatrix(planes, mode, 4.73, 0.34, DIRS[direction])
            for p in range(len(dims)):
                for x, y in zip(b.get_cells(p), f.get_cells(p)):
                    assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)), (direction, mode, "const", p)
            f.close()
            b.close()


def test_solo_kernel_vs_oracle_and_tagged(blosum62):
    """dp_affine_solo (context hint "tag_solo": one wave per pair, strips of 512 columns visited in blocks of 16 rows, no barrier):
    ragged batch around every strip boundary (511..513, 1023..1025, 1535..1537, 2046 columns), short and long queries around the
    16-row blocks, empty sequences, local (16-bit key layout and 13-bit layout) / global / semi-local, free and priced end
    gaps — against the oracle where it is fast enough, else against the tagged multi-wave kernel (itself pinned by the
    reference goldens)."""
    alpha, table = blosum62
    lens = [(30, 510), (17, 511), (16, 512), (15, 513), (33, 1022), (2, 1023), (1, 1024), (48, 1025), (31, 1534), (32, 1535), (64, 1536),
            (5, 2046), (0, 0), (0, 600), (700, 0), (2046, 2046), (1300, 700), (100, 90)]
    pairs = [homolog_pair(80000 + n, max(q, t, 1)) for n, (q, t) in enumerate(lens)]
    qs = [p[0][:q] for p, (q, t) in zip(pairs, lens)]
    ts = [p[1][:t] for p, (q, t) in zip(pairs, lens)]
    ctx = gpu_util.ctx()
    for mode, gi, ge, h16 in ((3, 11, 1, 1), (3, 11, 1, 0), (3, 2, 0, 1), (1, 11, 1, 1), (4, 11, 1, 1), (0, 7, 2, 1), (2, 7, 2, 1)):
        res = {}
        for solo in (0, 1):
            with ctx.hints(tag_solo=solo, h16=h16, key16=(0 if gi == 2 else 1)):
                b = aln_amd.Batch(ctx, qs, ts)
                b.dp_submatrix(alpha, table, mode, gi, ge, aln_amd.FWD, aln_amd.DP_FAST)
                assert b.kernel_name().startswith("dp_affine_solo" if solo else "dp_affine_tag"), b.kernel_name()
                b.reevaluate()
                sc, lists, st = b.optimal()
                res[solo] = ([b.get_cells(p) for p in range(len(qs))], sc, lists, st)
                b.close()
        a, c = res[0], res[1]
        assert np.array_equal(a[3], c[3]) and np.array_equal(a[1].view(np.uint32), c[1].view(np.uint32)), (mode, gi)
        for p in range(len(qs)):
            for x, y in zip(a[0][p], c[0][p]):
                assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)), (mode, gi, h16, p, lens[p])
            assert np.array_equal(a[2][p], c[2][p]), (mode, gi, p)
        for p in (0, 3, 5, 7, 11, 13, 17):                    # narrow or short enough for the O(n^3) oracle
            S = orc.sim_submatrix(qs[p], ts[p], alpha, table)
            rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, gi, ge))
            D, PQ, PT = c[0][p]
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (mode, gi, p)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (mode, gi, p)


def test_templates_beyond_the_row_sweep_kernels(blosum62):
    """The reference has no length limit (dpmatrix.h:250-259).  The O(n^2) row-sweep kernels hold a row in registers (tagged keys:
    4096 columns, plain int32: 8192); wider templates run in the exact-order kernel under DP_AUTO (same results, its O(n^3)
    scans), DP_FAST says ALN_E_TOO_LONG.  Checked against the oracle with a short query."""
    alpha, table = blosum62
    q, t = homolog_pair(99001, 8400, sub_rate=0.2, indel=4)
    q = q[:28]
    for tl, algo, want in ((8190, aln_amd.DP_FAST, "dp_affine_int"), (8400, aln_amd.DP_AUTO, "dp_exact")):
        tt = t[:tl]
        b = aln_amd.Batch(gpu_util.ctx(), [q], [tt])
        b.dp_submatrix(alpha, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, algo)
        assert want in b.kernel_name(), b.kernel_name()
        S = orc.sim_submatrix(q, tt, alpha, table)
        rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(aln_amd.LOCAL, 11, 1))
        D, PQ, PT = b.get_cells(0)
        assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), tl
        assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), tl
        scores, lists, status = b.optimal()
        rc2, sc, pl = orc.optimal(D0, PQ0, PT0, True)
        assert status[0] == 0 and bits_of(scores[0]) == bits_of(sc) and np.array_equal(lists[0], pl)
        b.close()
    b = aln_amd.Batch(gpu_util.ctx(), [q], [t])
    with pytest.raises(aln_amd.AlnError) as ei:
        b.dp_submatrix(alpha, table, aln_amd.LOCAL, 11, 1, aln_amd.FWD, aln_amd.DP_FAST)
    assert ei.value.code == aln_amd.E_TOO_LONG
    b.close()


def bits_of(x):
    return int(np.float32(x).view(np.uint32))
