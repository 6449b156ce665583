"""-m gpu parity tests of near-optimal enumeration (ConstrainedNearOptimal cw.h:68-284, UnconstrainedNearOptimal
ucw.h:64-236) on the device-resident planes, through the C ABI: set size, discovery/sort order, scores (bit
patterns), uids, pair lists, gapped strings, identities and FASTA annotations against the golden vectors of
the real reference and against the oracle on seeded homolog pairs."""
import numpy as np
import pytest

import aln_amd
import goldens
import gpu_util
import orc
from aln_amd.synth import homolog_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 0, 3], ids=["one_wave", "waves_auto", "waves3"])
def enum_waves(request):
    """cw / ucw / kscw searches run in the one-wave kernels (enumerate.hip, enumerate_ks.hip) and in the several-waves-per-pair kernel (enumerate_par.hip:
    tasks, slot tree, set order rebuilt on the host) — same sets, same order."""
    with gpu_util.ctx().hints(enum_waves=request.param):
        yield request.param


def _flags(case):
    return np.array([int(ch) for ch in case["flags"]], dtype=np.uint8) if "flags" in case else np.ones(len(case["t"]) + 2, np.uint8)


@pytest.mark.parametrize("prefix", ["known", "enum", "c1", "c4", "aaa"])
def test_golden_enumeration(prefix, blosum62, enum_waves):
    alpha, table = blosum62
    n_sets = 0
    for case in goldens.cases(prefix):
        if case["dir"] != "fwd" or not ("CW" in case["sets"] or "UCW" in case["sets"]):
            continue
        b = aln_amd.Batch(gpu_util.ctx(), [case["q"]], [case["t"]])
        b.dp_submatrix(alpha, table, case["mode"], case["gi"], case["ge"])
        nsub, delta = case.get("nsub", 10), case.get("delta", 0.3)
        for key, kind in (("CW", "cw"), ("UCW", "ucw")):
            if key not in case["sets"]:
                continue
            got = b.enumerate(0, kind, nsub, delta, _flags(case))
            tl, qls, idn = gpu_util.strings_for(case["q"], case["t"], [g["pairs"] for g in got])
            for g, i in zip(got, idn):
                assert goldens.f32bits(g["identity"]) == goldens.f32bits(i)
            ann = [orc.annot(g["score"], g["identity"]) for g in got]
            goldens.check_set(case, key, got, tl, qls, ann)
            n_sets += 1
        b.close()
    assert n_sets > 0


@pytest.mark.parametrize("kind", ["cw", "ucw"])
def test_enumeration_vs_oracle(kind, blosum62, enum_waves):
    """Seeded mutated homologs, integer (fast DP kernel) and fractional (exact kernel) gaps, several thresholds, region
    counts and set limits; one resident batch, every pair enumerated from its own planes."""
    if enum_waves == 3 and kind == "ucw":
        pytest.skip("the unconstrained searches of this test run to user_limit (minutes): one several-wave variant is enough")
    alpha, table = blosum62
    rng = np.random.RandomState(17)
    lens = [9, 24, 57, 64, 90, 130]
    pairs = [homolog_pair(61000 + n, ln, sub_rate=0.2, indel=3) for n, ln in enumerate(lens)]
    for mode in (1, 3, 4):
        for (gi, ge) in ((11, 1), (4.73, 0.34)):
            b = aln_amd.Batch(gpu_util.ctx(), [p[0] for p in pairs], [p[1] for p in pairs])
            b.dp_submatrix(alpha, table, mode, gi, ge)
            for p, (q, t) in enumerate(pairs):
                T = len(t) + 2
                flags = orc.make_subopt_regions(T, int(rng.randint(1, 9)))
                delta = float(rng.choice([0.01, 0.05, 0.1, 0.3]))
                nsub = int(rng.choice([3, 20, 256]))
                if kind == "ucw" and len(q) > 60:
                    delta = min(delta, 0.05)        # keep the unconstrained search small
                S = orc.sim_submatrix(q, t, alpha, table)
                gap = orc.Gap(mode, gi, ge)
                rc, D0, PQ0, PT0 = orc.dp_build(S, gap)
                rc2, sc, pl = orc.optimal(D0, PQ0, PT0, mode == 3)
                s = orc.AliSet()
                s.push(pl, sc)
                orc.enumerate_noa(kind, D0, PQ0, PT0, S, gap, flags, nsub, delta, s)
                s.identity(q, t)
                got = b.enumerate(p, kind, nsub, delta, flags, max_alignments=max(nsub, len(s)) + 2)
                assert len(got) == len(s), (kind, mode, gi, p, len(got), len(s))
                for k, g in enumerate(got):
                    r = s.get(k)
                    assert np.float32(g["score"]).view(np.uint32) == r["score"].view(np.uint32), (kind, mode, gi, p, k)
                    assert g["uid"] == r["uid"]
                    assert np.array_equal(g["pairs"], r["pairs"]), (kind, mode, gi, p, k)
                    assert np.float32(g["identity"]).view(np.uint32) == r["identity"].view(np.uint32)
            b.close()


def test_enumeration_user_limit_and_overflow(blosum62, enum_waves):
    """user_limit forces the optimal path once the set is larger (cw.h:127-140); a too-small output buffer is an error."""
    alpha, table = blosum62
    q, t = homolog_pair(62001, 80, sub_rate=0.25, indel=3)
    b = aln_amd.Batch(gpu_util.ctx(), [q], [t])
    b.dp_submatrix(alpha, table, 1, 11, 1)
    S = orc.sim_submatrix(q, t, alpha, table)
    gap = orc.Gap(1, 11, 1)
    rc, D0, PQ0, PT0 = orc.dp_build(S, gap)
    rc2, sc, pl = orc.optimal(D0, PQ0, PT0, False)
    for lim in (5, 40):
        s = orc.AliSet()
        s.push(pl, sc)
        orc.enumerate_noa("ucw", D0, PQ0, PT0, S, gap, None, 100000, 0.2, s, user_limit=lim)
        got = b.enumerate(0, "ucw", 100000, 0.2, user_limit=lim, max_alignments=len(s) + 2)
        assert len(got) == len(s)
        for k, g in enumerate(got):
            r = s.get(k)
            assert np.float32(g["score"]).view(np.uint32) == r["score"].view(np.uint32)
            assert np.array_equal(g["pairs"], r["pairs"])
    with pytest.raises(aln_amd.AlnError) as ei:
        b.enumerate(0, "ucw", 100000, 0.2, max_alignments=2)
    assert ei.value.code == aln_amd.E_OVERFLOW
    b.close()


@pytest.mark.parametrize("kind", ["cw", "ucw"])
def test_batched_enumeration_grows_to_user_limit(kind, blosum62, enum_waves):
    """aln_batch_enumerate_all with alignment pools far too small and no pool retries left: a pair whose set outgrows its slots is
    searched again until the pool has the size user_limit bounds, and comes back with the set the reference's brake defines
    (cw.h:127-140: beyond user_limit every branch is forced down the optimal path) — the oracle's set, not ALN_E_OVERFLOW."""
    alpha, table = blosum62
    lens = [80, 33, 110, 64]
    pairs = [homolog_pair(64500 + n, ln, sub_rate=0.25, indel=3) for n, ln in enumerate(lens)]
    maxT = max(len(t) for _, t in pairs) + 2
    ctx = gpu_util.ctx()
    b = aln_amd.Batch(ctx, [p[0] for p in pairs], [p[1] for p in pairs])
    b.dp_submatrix(alpha, table, 1, 11, 1)
    flags = np.zeros((len(pairs), maxT), dtype=np.uint8)
    for p, (q, t) in enumerate(pairs):
        flags[p, :len(t) + 2] = orc.make_subopt_regions(len(t) + 2, 3)
    lim, nsub, delta = 40, 100000, 0.3
    want = []
    for p, (q, t) in enumerate(pairs):
        S = orc.sim_submatrix(q, t, alpha, table)
        gap = orc.Gap(1, 11, 1)
        rc, D0, PQ0, PT0 = orc.dp_build(S, gap)
        rc2, sc, pl = orc.optimal(D0, PQ0, PT0, False)
        s = orc.AliSet()
        s.push(pl, sc)
        assert orc.enumerate_noa(kind, D0, PQ0, PT0, S, gap, flags[p, :len(t) + 2] if kind == "cw" else None, nsub, delta, s, user_limit=lim) == 0
        want.append(s)
    assert max(len(s) for s in want) > lim                       # the brake really acted
    K = max(len(s) for s in want) + 2
    with ctx.hints(enum_pool_retries=0):
        n_out, scores, lengths, lists, status = b.enumerate_all(kind, nsub, delta, flags if kind == "cw" else None, K=K, node_cap=1 << 16,
                                                                 ali_cap=16, user_limit=lim)
    assert (status == 0).all(), status
    for p, s in enumerate(want):
        assert n_out[p] == len(s), (p, n_out[p], len(s))
        for k in range(len(s)):
            r = s.get(k)
            assert scores[p, k].view(np.uint32) == r["score"].view(np.uint32), (p, k)
            assert np.array_equal(lists[p, k, :lengths[p, k]], r["pairs"]), (p, k)
    b.close()


@pytest.mark.parametrize("kind", ["cw", "ucw"])
def test_batched_enumeration_matches_oracle(kind, blosum62, enum_waves):
    """aln_batch_enumerate_all (BASELINE config 4 form): every pair of a ragged resident batch in ONE launch, per-pair
    SuboptFlags rows; set size, order, score bits and pair lists against the oracle, and against the one-pair entry."""
    alpha, table = blosum62
    lens = [9, 24, 57, 64, 90, 130, 33, 71]
    pairs = [homolog_pair(63000 + n, ln, sub_rate=0.2, indel=3) for n, ln in enumerate(lens)]
    maxT = max(len(t) for _, t in pairs) + 2
    nsub, delta = 20, (0.05 if kind == "cw" else 0.02)
    for mode, (gi, ge) in ((3, (11, 1)), (1, (11, 1)), (4, (4.73, 0.34))):
        b = aln_amd.Batch(gpu_util.ctx(), [p[0] for p in pairs], [p[1] for p in pairs])
        b.dp_submatrix(alpha, table, mode, gi, ge)
        flags = np.zeros((len(pairs), maxT), dtype=np.uint8)
        for p, (q, t) in enumerate(pairs):
            flags[p, :len(t) + 2] = orc.make_subopt_regions(len(t) + 2, 1 + p % 5)
        n_out, scores, lengths, lists, status = b.enumerate_all(kind, nsub, delta, flags, K=nsub + 2, node_cap=1 << 22, ali_cap=1 << 17)
        assert (status == 0).all()
        for p, (q, t) in enumerate(pairs):
            S = orc.sim_submatrix(q, t, alpha, table)
            gap = orc.Gap(mode, gi, ge)
            rc, D0, PQ0, PT0 = orc.dp_build(S, gap)
            rc2, sc, pl = orc.optimal(D0, PQ0, PT0, mode == 3)
            s = orc.AliSet()
            s.push(pl, sc)
            orc.enumerate_noa(kind, D0, PQ0, PT0, S, gap, flags[p, :len(t) + 2], nsub, delta, s)
            assert n_out[p] == len(s), (kind, mode, p, n_out[p], len(s))
            one = b.enumerate(p, kind, nsub, delta, flags[p, :len(t) + 2], max_alignments=nsub + 2)
            for k in range(len(s)):
                r = s.get(k)
                assert scores[p, k].view(np.uint32) == r["score"].view(np.uint32), (kind, mode, p, k)
                assert np.array_equal(lists[p, k, :lengths[p, k]], r["pairs"]), (kind, mode, p, k)
                assert np.array_equal(one[k]["pairs"], r["pairs"])
        b.close()


def test_batched_enumeration_overflow_is_per_pair(blosum62, enum_waves):
    """A pool that is too small for one pair flags only that pair (status ALN_E_OVERFLOW); the others are complete."""
    alpha, table = blosum62
    pairs = [homolog_pair(64000, 12), homolog_pair(64001, 120, sub_rate=0.25, indel=3)]
    b = aln_amd.Batch(gpu_util.ctx(), [p[0] for p in pairs], [p[1] for p in pairs])
    b.dp_submatrix(alpha, table, 1, 11, 1)
    n_out, scores, lengths, lists, status = b.enumerate_all("ucw", 50, 0.2, None, K=52, node_cap=200, raise_on_overflow=False)
    assert status[0] == 0 and status[1] == aln_amd.E_OVERFLOW
    one = b.enumerate(0, "ucw", 50, 0.2, max_alignments=52)
    assert n_out[0] == len(one)
    for k, g in enumerate(one):
        assert np.array_equal(lists[0, k, :lengths[0, k]], g["pairs"])
    b.close()


def test_kscw_vs_oracle(blosum62, enum_waves):
    """KSConstrainedNearOptimal (kscw.h:109-351) on the device: per-node candidate collection, libstdc++-ordered sort /
    partial_sort of the operations, halving limits, forced optimal paths — against the oracle's restatement (which sorts with
    the host's std::sort / std::partial_sort).  Parity UNPINNED against the reference (its header does not build on LP64)."""
    alpha, table = blosum62
    rng = np.random.RandomState(29)
    lens = [9, 24, 57, 64, 90, 130, 200]
    pairs = [homolog_pair(65000 + n, ln, sub_rate=0.2, indel=3) for n, ln in enumerate(lens)]
    for mode, (gi, ge) in ((1, (11, 1)), (4, (4.73, 0.34)), (3, (11, 1))):
        b = aln_amd.Batch(gpu_util.ctx(), [p[0] for p in pairs], [p[1] for p in pairs])
        b.dp_submatrix(alpha, table, mode, gi, ge)
        for p, (q, t) in enumerate(pairs):
            T = len(t) + 2
            flags = orc.make_subopt_regions(T, int(rng.randint(2, 9)))
            delta = float(rng.choice([0.05, 0.1, 0.3]))
            nsub = int(rng.choice([5, 40, 300]))
            klim = int(rng.choice([1, 2, 3, 4, 8, 16, 33]))
            ulim = int(rng.choice([100000, 7]))
            S = orc.sim_submatrix(q, t, alpha, table)
            gap = orc.Gap(mode, gi, ge)
            rc, D0, PQ0, PT0 = orc.dp_build(S, gap)
            rc2, sc, pl = orc.optimal(D0, PQ0, PT0, mode == 3)
            s = orc.AliSet()
            s.push(pl, sc)
            if p == len(pairs) - 1:
                delta, klim, ulim, nsub = 0.6, (33 if mode == 1 else 5), 100000, 300      # hundreds of operations per node: the
            assert orc.enumerate_ks(D0, PQ0, PT0, S, gap, flags, nsub, delta, klim, s, user_limit=ulim) == 0   # partition / heap paths
            s.identity(q, t)
            got = b.enumerate(p, "kscw", nsub, delta, flags, user_limit=ulim, k_limit=klim, max_alignments=max(nsub, len(s)) + 2)
            assert len(got) == len(s), (mode, gi, p, klim, len(got), len(s))
            for k, g in enumerate(got):
                r = s.get(k)
                assert np.float32(g["score"]).view(np.uint32) == r["score"].view(np.uint32), (mode, gi, p, klim, k)
                assert g["uid"] == r["uid"], (mode, gi, p, klim, k, g["uid"], r["uid"])
                assert np.array_equal(g["pairs"], r["pairs"]), (mode, gi, p, klim, k)
        b.close()


def test_batched_kscw_matches_single_pair_and_oracle(blosum62, enum_waves):
    """aln_batch_enumerate_all with ALN_ENUM_KSCW: every pair of a ragged batch in one launch (one workgroup per pair, own
    pool slices); set size, order, score bits and pair lists against the one-pair entry point and the oracle's restatement."""
    alpha, table = blosum62
    lens = [9, 24, 57, 64, 90, 130, 33, 200]
    pairs = [homolog_pair(66000 + n, ln, sub_rate=0.2, indel=3) for n, ln in enumerate(lens)]
    maxT = max(len(t) for _, t in pairs) + 2
    for mode, (gi, ge), nsub, delta, klim in ((1, (11, 1), 40, 0.1, 4), (3, (11, 1), 20, 0.3, 16), (4, (4.73, 0.34), 300, 0.05, 2)):
        b = aln_amd.Batch(gpu_util.ctx(), [p[0] for p in pairs], [p[1] for p in pairs])
        b.dp_submatrix(alpha, table, mode, gi, ge)
        flags = np.zeros((len(pairs), maxT), dtype=np.uint8)
        for p, (q, t) in enumerate(pairs):
            flags[p, :len(t) + 2] = orc.make_subopt_regions(len(t) + 2, 2 + p % 5)
        n_out, scores, lengths, lists, status = b.enumerate_all("kscw", nsub, delta, flags, K=nsub + 2, node_cap=1 << 20, ali_cap=1 << 16,
                                                                 k_limit=klim)
        assert (status == 0).all()
        for p, (q, t) in enumerate(pairs):
            S = orc.sim_submatrix(q, t, alpha, table)
            gap = orc.Gap(mode, gi, ge)
            rc, D0, PQ0, PT0 = orc.dp_build(S, gap)
            rc2, sc, pl = orc.optimal(D0, PQ0, PT0, mode == 3)
            s = orc.AliSet()
            s.push(pl, sc)
            assert orc.enumerate_ks(D0, PQ0, PT0, S, gap, flags[p, :len(t) + 2], nsub, delta, klim, s) == 0
            assert n_out[p] == len(s), (mode, p, n_out[p], len(s))
            one = b.enumerate(p, "kscw", nsub, delta, flags[p, :len(t) + 2], k_limit=klim, max_alignments=nsub + 2)
            assert len(one) == len(s)
            for k in range(len(s)):
                r = s.get(k)
                assert scores[p, k].view(np.uint32) == r["score"].view(np.uint32), (mode, p, k)
                assert np.array_equal(lists[p, k, :lengths[p, k]], r["pairs"]), (mode, p, k)
                assert np.array_equal(one[k]["pairs"], r["pairs"])
        b.close()


def test_crcw_vs_oracle(blosum62):
    """CRConstrainedNearOptimal (crcw.h:134-594) on the device: sorted operations, sub-paths followed to the end of the flag
    region, the overlap filter, limits, forced optimal paths — one pair at a time and the whole batch in one launch, against
    the oracle's restatement.  Parity UNPINNED against the reference (crcw.h does not build on LP64, :242); the one
    out-of-bounds read of the source (regions[-1], :387) is modelled as a region of its own in both (the oracle counts how
    often it is reached: it is, on these inputs)."""
    alpha, table = blosum62
    rng = np.random.RandomState(31)
    lens = [9, 24, 57, 64, 90, 130, 200, 3, 1]
    pairs = [homolog_pair(67000 + n, ln, sub_rate=0.2, indel=3) for n, ln in enumerate(lens)]
    maxT = max(len(t) for _, t in pairs) + 2
    oob_total = 0
    for mode, (gi, ge) in ((1, (11, 1)), (4, (4.73, 0.34)), (3, (11, 1))):
        b = aln_amd.Batch(gpu_util.ctx(), [p[0] for p in pairs], [p[1] for p in pairs])
        b.dp_submatrix(alpha, table, mode, gi, ge)
        for rep in range(2):
            nsub = int(rng.choice([5, 40, 300]))
            delta = float(rng.choice([0.05, 0.1, 0.3, 0.6]))
            klim = int(rng.choice([1, 2, 3, 4, 8, 16, 33]))
            slim = int(rng.choice([1, 3, 10, 100]))
            ulim = int(rng.choice([100000, 7]))
            movl = float(rng.choice([0.0, 0.3, 0.75, 1.0]))
            flags = np.zeros((len(pairs), maxT), dtype=np.uint8)
            for p, (q, t) in enumerate(pairs):
                flags[p, :len(t) + 2] = orc.make_subopt_regions(len(t) + 2, 1 + (p + rep) % 7)
            n_out, scores, lengths, lists, status = b.enumerate_all("crcw", nsub, delta, flags, K=nsub + 2, node_cap=1 << 20, ali_cap=1 << 16,
                                                                     k_limit=klim, sort_limit=slim, max_overlap=movl, user_limit=ulim)
            assert (status == 0).all(), (mode, rep, status)
            for p, (q, t) in enumerate(pairs):
                S = orc.sim_submatrix(q, t, alpha, table)
                gap = orc.Gap(mode, gi, ge)
                rc, D0, PQ0, PT0 = orc.dp_build(S, gap)
                rc2, sc, pl = orc.optimal(D0, PQ0, PT0, mode == 3)
                s = orc.AliSet()
                s.push(pl, sc)
                rc3, oob = orc.enumerate_cr(D0, PQ0, PT0, S, gap, flags[p, :len(t) + 2], nsub, delta, klim, s, sort_limit=slim,
                                            user_limit=ulim, max_overlap=movl)
                assert rc3 == 0
                oob_total += oob
                s.identity(q, t)
                what = (mode, rep, p, nsub, delta, klim, slim, ulim, movl)
                assert n_out[p] == len(s), what + (n_out[p], len(s))
                one = b.enumerate(p, "crcw", nsub, delta, flags[p, :len(t) + 2], user_limit=ulim, k_limit=klim, sort_limit=slim,
                                  max_overlap=movl, max_alignments=max(nsub, len(s)) + 2)
                assert len(one) == len(s), what
                for k in range(len(s)):
                    r = s.get(k)
                    assert scores[p, k].view(np.uint32) == r["score"].view(np.uint32), what + (k,)
                    assert np.array_equal(lists[p, k, :lengths[p, k]], r["pairs"]), what + (k,)
                    assert np.float32(one[k]["score"]).view(np.uint32) == r["score"].view(np.uint32), what + (k,)
                    assert one[k]["uid"] == r["uid"], what + (k, one[k]["uid"], r["uid"])
                    assert np.array_equal(one[k]["pairs"], r["pairs"]), what + (k,)
                    assert np.float32(one[k]["identity"]).view(np.uint32) == r["identity"].view(np.uint32)
        b.close()
    assert oob_total > 0
