"""The oracle's profile path (Hmap2Eval arithmetic: hmath.h dot/pearson, exp, z-normalisation, shift,
pre_calculate gap arrays, DP with min(t[t1],t[t2]) gaps, Optimal) against golden vectors produced by the REAL
hmath.h / SimilarityMatrix / DPMatrix / Optimal through oracle/ref_profile.cpp.  CPU only."""
import json
import os

import numpy as np

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_profile_cases():
    meta = json.load(open(os.path.join(GOLD, "profile_cases.json")))["cases"]
    z = np.load(os.path.join(GOLD, "profile_cases.npz"), allow_pickle=False)
    return meta, z


def inputs(z, key):
    qp = {k: z["%s/q_%s" % (key, k)] for k in ("aa", "sse", "conf")}
    tp = {k: z["%s/t_%s" % (key, k)] for k in ("aa", "sse", "conf")}
    return qp, tp


def test_profile_path_matches_reference_primitives():
    meta, z = load_profile_cases()
    assert len(meta) >= 30
    for m in meta:
        qp, tp = inputs(z, m["inputs"])
        name = m["name"]
        S = orc.hmap2_sim(qp, tp, m["alpha"], m["zero_shift"])
        assert np.array_equal(S.view(np.uint32), z[name + "/S"].view(np.uint32)), name
        tgi, tge = orc.hmap2_precalc(tp, m["gi"], m["ge"], m["beta"])
        assert np.array_equal(tgi.view(np.uint32), z[name + "/TGI"].view(np.uint32)), name
        assert np.array_equal(tge.view(np.uint32), z[name + "/TGE"].view(np.uint32)), name
        rc, D, PQ, PT = orc.dp_build(S, orc.Gap(m["mode"], tgi=tgi, tge=tge), m["dir"], bug_b4=True)
        assert np.array_equal(D.view(np.uint32), z[name + "/H"].view(np.uint32)), name
        assert np.array_equal(PQ, z[name + "/PQ"]) and np.array_equal(PT, z[name + "/PT"]), name
        if "opt" in m:
            rc2, sc, pairs = orc.optimal(D, PQ, PT, m["mode"] == 3)
            assert int(sc.view(np.uint32)) == m["opt"]["score"] and pairs.reshape(-1).tolist() == m["opt"]["pairs"], name
        # dot_product / pearson_corr probes (hmath.h:18-26, :94-103)
        prim = z[name + "/PRIM"]
        k = 0
        Q, T = len(qp["conf"]), len(tp["conf"])
        for i in range(1, min(Q - 1, 6)):
            for j in range(1, min(T - 1, 6)):
                a = np.ascontiguousarray(qp["aa"][i])
                b = np.ascontiguousarray(tp["aa"][j])
                d = np.float32(orc.lib().orc_dot(orc._fp(a), orc._fp(b), 20))
                sa = np.ascontiguousarray(qp["sse"][i])
                sb = np.ascontiguousarray(tp["sse"][j])
                pc = np.float32(orc.lib().orc_pearson(orc._fp(sa), orc._fp(sb), 3))
                assert d.view(np.uint32) == prim[k:k + 1].view(np.uint32)[0]
                assert pc.view(np.uint32) == prim[k + 1:k + 2].view(np.uint32)[0]
                k += 2
