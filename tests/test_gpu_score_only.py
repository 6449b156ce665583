"""-m gpu: all-vs-all score-only path (BASELINE config 5 shape): aln_score_all_vs_all against (a) the resident-plane
path's Optimal(local) scores for every pair and (b) the oracle on a sample; ragged lengths around every kernel
variant boundary (256/512/1024 columns), row-block calls as a rank of a multi-GPU job would issue them."""
import numpy as np
import pytest

import aln_amd
import gpu_util
import orc
from aln_amd.synth import MT19937, homolog_pair, residues

pytestmark = pytest.mark.gpu


def make_set(seed, lens):
    out = []
    for n, ln in enumerate(lens):
        g = MT19937(seed + n)
        out.append(residues(g, ln))
    return out


@pytest.mark.parametrize("tmax", [120, 254, 255, 500, 1022, 1500])
def test_scores_equal_plane_path_and_oracle(tmax, blosum62):
    alpha, table = blosum62
    rng = np.random.RandomState(tmax)
    qlens = [1, 7, 64, 200, 333] + [int(rng.randint(2, 400)) for _ in range(4)]
    tlens = [tmax, tmax - 1, 1, 5, max(tmax // 2, 2)] + [int(rng.randint(2, tmax)) for _ in range(4)]
    qs, ts = make_set(81000 + tmax, qlens), make_set(82000 + tmax, tlens)
    # plant homologs so that some scores are large
    h1, h2 = homolog_pair(83000 + tmax, min(tmax, 200))
    qs[3], ts[4] = h1, h2[:tlens[4]] if tlens[4] < len(h2) else h2
    ctx = gpu_util.ctx()
    got = aln_amd.score_all_vs_all(ctx, qs, ts, alpha, table, 11, 1)
    assert got.shape == (len(qs), len(ts))
    # (a) every pair through the plane path (tagged DP kernel + find_max)
    qi, ti = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
    b = aln_amd.Batch(ctx, qs, ts, qi.reshape(-1), ti.reshape(-1))
    b.dp_submatrix(alpha, table, aln_amd.LOCAL, 11, 1)
    scores, _, status = b.optimal(want_pairs=False)
    b.close()
    assert np.array_equal(got.reshape(-1).view(np.uint32), scores.view(np.uint32))
    # (b) the oracle on a sample
    for (i, j) in [(0, 0), (3, 4), (4, 0), (8, 8), (2, 1)]:
        S = orc.sim_submatrix(qs[i], ts[j], alpha, table)
        rc, D, PQ, PT = orc.dp_build(S, orc.Gap(orc.LOCAL, 11, 1))
        assert got[i, j] == orc.optimal(D, PQ, PT, True)[1]
    # row blocks (what one rank of a sharded job computes)
    blk = aln_amd.score_all_vs_all(ctx, qs, ts, alpha, table, 11, 1, 2, 7)
    assert np.array_equal(blk, got[2:7])
    # the default is the packed two-queries-per-wave kernel; the one-query 32-bit kernel must agree
    with ctx.hints(score_packed=0):
        assert np.array_equal(aln_amd.score_all_vs_all(ctx, qs, ts, alpha, table, 11, 1), got)


@pytest.mark.parametrize("mode", [0, 1, 2, 4])
def test_non_local_scores_equal_plane_path_and_oracle(mode, blosum62):
    """The four non-local align types: aln_score_all_vs_all reports the final cell's score (what Optimal returns for them) —
    against the resident-plane path for every pair of ragged sets around the 256-column class boundaries (empty and 1-residue
    sequences included), gaps 11/1 and 3/0, and against the oracle on a sample."""
    alpha, table = blosum62
    rng = np.random.RandomState(100 + mode)
    qlens = [0, 1, 2, 7, 64, 200, 333] + [int(rng.randint(2, 400)) for _ in range(3)]
    tlens = [0, 1, 2, 254, 255, 256, 257, 600, 1022] + [int(rng.randint(2, 1500)) for _ in range(3)]
    qs, ts = make_set(83000 + mode, qlens), make_set(84000 + mode, tlens)
    qs[5] = ts[7][100:300]                                   # plant homologs: long diagonals and positive scores
    qs[6] = ts[8][500:833]
    ctx = gpu_util.ctx()
    for gi, ge in ((11, 1), (3, 0)):
        got = aln_amd.score_all_vs_all(ctx, qs, ts, alpha, table, gi, ge, align_type=mode)
        qi, ti = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
        b = aln_amd.Batch(ctx, qs, ts, qi.reshape(-1), ti.reshape(-1))
        b.dp_submatrix(alpha, table, mode, gi, ge)
        want = b.corner_scores().reshape(len(qs), len(ts))
        b.close()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.argwhere(got != want)[:5]
        for (i, j) in ((0, 0), (1, 1), (3, 4), (5, 7), (6, 8), (2, 3), (9, 11)):
            S = orc.sim_submatrix(qs[i], ts[j], alpha, table)
            rc, D, PQ, PT = orc.dp_build(S, orc.Gap(mode, gi, ge))
            assert got[i, j] == D[-1, -1], (mode, gi, i, j)
    blk = aln_amd.score_all_vs_all(ctx, qs, ts, alpha, table, 11, 1, 3, 8, align_type=mode)
    full = aln_amd.score_all_vs_all(ctx, qs, ts, alpha, table, 11, 1, align_type=mode)
    assert np.array_equal(blk, full[3:8])


def test_score_only_rejects_what_it_cannot_do(blosum62):
    alpha, table = blosum62
    ctx = gpu_util.ctx()
    with pytest.raises(aln_amd.AlnError) as ei:
        aln_amd.score_all_vs_all(ctx, ["ACJ"], ["ACD"], alpha, table, 11, 1)
    assert ei.value.code == aln_amd.E_RESIDUE


@pytest.mark.parametrize("mode", [3, 1, 4])
def test_long_templates_and_fractional_gaps_go_through_full_builds(mode, blosum62):
    """The reference scores any pair with any evaluator values (N x DPMatrix + Optimal).  What the register-resident kernels do
    not take — templates beyond 2048 columns, fractional gaps — aln_score_all_vs_all computes through resident batches of full
    builds (tagged / int32 / exact-order kernels) and Optimal's score, in the same call and the same output matrix."""
    alpha, table = blosum62
    qlens = [5, 120, 333, 90]
    tlens = [300, 2047, 2048, 2600, 40, 3100]                 # two templates beyond the kernels' 2048 columns, among short ones
    qs, ts = make_set(85000 + mode, qlens), make_set(86000 + mode, tlens)
    qs[2] = ts[3][1000:1333]                                  # a homolog inside a long template
    ctx = gpu_util.ctx()
    for gi, ge in ((11, 1), (4.73, 0.34)):
        got = aln_amd.score_all_vs_all(ctx, qs, ts, alpha, table, gi, ge, align_type=mode)
        qi, ti = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
        b = aln_amd.Batch(ctx, qs, ts, qi.reshape(-1), ti.reshape(-1))
        b.dp_submatrix(alpha, table, mode, gi, ge)
        want = (b.optimal(want_pairs=False)[0] if mode == 3 else b.corner_scores()).reshape(len(qs), len(ts))
        b.close()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (mode, gi, np.argwhere(got != want)[:5])
        for (i, j) in ((0, 0), (2, 3), (1, 5)):
            S = orc.sim_submatrix(qs[i], ts[j], alpha, table)
            rc, D, PQ, PT = orc.dp_build(S, orc.Gap(mode, gi, ge))
            ref = orc.optimal(D, PQ, PT, True)[1] if mode == 3 else D[-1, -1]
            assert np.float32(got[i, j]).view(np.uint32) == np.float32(ref).view(np.uint32), (mode, gi, i, j)


def test_more_than_one_slab_of_query_rows(blosum62):
    """> 32768 query rows: the packed kernel walks the rows in slabs of 32768 (blockIdx.y limit), each with its own length
    order on the device; both kernels must agree on every row and equal the oracle on samples from both slabs."""
    alpha, table = blosum62
    ctx = gpu_util.ctx()
    n = 32768 + 700
    g = MT19937(99)
    lens = (g.draw(n) % 23).astype(int)                    # 0 .. 22 residues, ragged, empties included
    blob = residues(g, int(lens.sum()))
    off = np.concatenate([[0], np.cumsum(lens)])
    qs = [blob[off[k]:off[k + 1]] for k in range(n)]
    ts = [residues(g, 30), residues(g, 7), "", residues(g, 300)]
    got = aln_amd.score_all_vs_all(ctx, qs, ts, alpha, table, 11, 1)
    with ctx.hints(score_packed=0):
        plain = aln_amd.score_all_vs_all(ctx, qs, ts, alpha, table, 11, 1)
    assert np.array_equal(got.view(np.uint32), plain.view(np.uint32))
    for i in (0, 1, 32767, 32768, 32769, n - 1, 20000, 33000):
        for j in range(len(ts)):
            S = orc.sim_submatrix(qs[i], ts[j], alpha, table)
            rc, D, PQ, PT = orc.dp_build(S, orc.Gap(orc.LOCAL, 11, 1))
            assert got[i, j] == orc.optimal(D, PQ, PT, True)[1], (i, j)
