"""-m gpu parity tests of the profile path (BASELINE config 3 shape): Hmap2Eval similarity + z-normalisation on the
device, pre_calculate gap arrays, exact DP with min(t[t1],t[t2]) gaps and Optimal — against golden vectors from the
real hmath.h / DPMatrix (oracle/ref_profile.cpp) and against the oracle on a ragged batch.  Everything is compared
bit for bit (uint32 patterns); the device exp() is the host libm's algorithm, see csrc/sim_hmap2.hip."""
import numpy as np
import pytest

import aln_amd
import gpu_util
import orc
from aln_amd.synth import random_profile
from test_oracle_profile import inputs, load_profile_cases

pytestmark = pytest.mark.gpu


def test_golden_profile_cases():
    meta, z = load_profile_cases()
    for m in meta:
        qp, tp = inputs(z, m["inputs"])
        name = m["name"]
        Q, T = len(qp["conf"]), len(tp["conf"])
        b = aln_amd.Batch(gpu_util.ctx(), ["A" * (Q - 2)], ["A" * (T - 2)])
        tgi, tge = b.dp_hmap2(qp, tp, m["mode"], m["gi"], m["ge"], m["alpha"], m["beta"], m["zero_shift"], True, m["dir"])
        assert "dp_exact" in b.kernel_name()
        assert np.array_equal(tgi.view(np.uint32), z[name + "/TGI"].view(np.uint32)), name
        assert np.array_equal(tge.view(np.uint32), z[name + "/TGE"].view(np.uint32)), name
        assert np.array_equal(b.get_sim(0).view(np.uint32), z[name + "/S"].view(np.uint32)), name
        D, PQ, PT = b.get_cells(0)
        assert np.array_equal(D.view(np.uint32), z[name + "/H"].view(np.uint32)), name
        assert np.array_equal(PQ, z[name + "/PQ"]) and np.array_equal(PT, z[name + "/PT"]), name
        if "opt" in m:
            scores, lists, status = b.optimal()
            assert status[0] == 0
            assert int(np.float32(scores[0]).view(np.uint32)) == m["opt"]["score"]
            assert lists[0].reshape(-1).tolist() == m["opt"]["pairs"], name
        b.close()


def test_profile_batch_vs_oracle():
    """A ragged batch sharing one query pool and one template pool (pairs pick from the pools), global and semi-local
    as nalign2.cpp:84-85 / SURVEY 8d C3, sizes crossing the 64-element chunks of the sequential statistics kernel."""
    qlens, tlens = [5, 63, 64, 130], [66, 7, 129, 200]
    qps = [random_profile(71000 + n, ln) for n, ln in enumerate(qlens)]
    tps = [random_profile(72000 + n, ln) for n, ln in enumerate(tlens)]
    qpool = {k: np.concatenate([p[k] for p in qps]) for k in ("aa", "sse", "conf")}
    tpool = {k: np.concatenate([p[k] for p in tps]) for k in ("aa", "sse", "conf")}
    q_idx = [0, 1, 2, 3, 3, 0]
    t_idx = [0, 1, 2, 3, 0, 3]
    for mode in (1, 4, 3):
        b = aln_amd.Batch(gpu_util.ctx(), ["A" * n for n in qlens], ["A" * n for n in tlens], q_idx, t_idx)
        b.dp_hmap2(qpool, tpool, mode, 4.73, 0.34, 0.5, 1.0, 0.12)
        scores, lists, status = b.optimal()
        for p, (qi, ti) in enumerate(zip(q_idx, t_idx)):
            S = orc.hmap2_sim(qps[qi], tps[ti], 0.5, 0.12)
            tgi, tge = orc.hmap2_precalc(tps[ti], 4.73, 0.34, 1.0)
            assert np.array_equal(b.get_sim(p).view(np.uint32), S.view(np.uint32)), (mode, p)
            rc, D0, PQ0, PT0 = orc.dp_build(S, orc.Gap(mode, tgi=tgi, tge=tge))
            D, PQ, PT = b.get_cells(p)
            assert np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (mode, p)
            assert np.array_equal(PQ, PQ0) and np.array_equal(PT, PT0), (mode, p)
            rc2, sc, pairs = orc.optimal(D0, PQ0, PT0, mode == 3)
            assert status[p] == 0 and np.float32(scores[p]).view(np.uint32) == sc.view(np.uint32)
            assert np.array_equal(lists[p], pairs)
        b.close()


def test_full_size_exact_kernels_agree():
    """BASELINE config-3 size (2000 x 2000 profiles, Hmap2Eval on the device, min(t[t1],t[t2]) gaps): the tiled kernel (shared
    far-left deletion scans), the slot kernel (per-row scans) and — on a 700-column pair, where it finishes in seconds — the
    literal O(n^3) kernel are independent programmes for the same arithmetic; planes, scores and paths must be identical."""
    def run(qps, tps, hints):
        with gpu_util.ctx().hints(**hints):
            qpool = {k: np.concatenate([p[k] for p in qps]) for k in ("aa", "sse", "conf")}
            tpool = {k: np.concatenate([p[k] for p in tps]) for k in ("aa", "sse", "conf")}
            b = aln_amd.Batch(gpu_util.ctx(), ["A" * (len(p["conf"]) - 2) for p in qps], ["A" * (len(p["conf"]) - 2) for p in tps])
            b.dp_hmap2(qpool, tpool, 1, 4.73, 0.34, 0.5, 1.0, 0.12)
            name = b.kernel_name()
            cells = [b.get_cells(p) for p in range(len(qps))]
            sc, lists, status = b.optimal()
            b.close()
            return name, cells, sc, lists

    def same(a, c):
        for p in range(len(a[1])):
            for x, y in zip(a[1][p], c[1][p]):
                assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)), (a[0], c[0], p)
            assert np.array_equal(a[3][p], c[3][p])
        assert np.array_equal(a[2].view(np.uint32), c[2].view(np.uint32))

    qps = [random_profile(73000, 2000), random_profile(73001, 1990)]
    tps = [random_profile(74000, 2000), random_profile(74001, 2000)]
    tiled = run(qps, tps, {})
    slots = run(qps, tps, {"exact_tiles": 0})
    assert "dp_exact_tiled" in tiled[0] and "dp_exact_blocked" in slots[0]
    same(tiled, slots)
    qps, tps = [random_profile(73002, 300)], [random_profile(74002, 700)]
    tiled = run(qps, tps, {})
    literal = run(qps, tps, {"exact_literal": 1})
    assert "dp_exact_tiled" in tiled[0] and literal[0].startswith("dp_exact_kernel")
    same(tiled, literal)
