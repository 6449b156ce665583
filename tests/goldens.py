"""Loader for tests/golden/aa_cases.{json,npz} (outputs of the real reference, see oracle/gen_golden.py)."""
import hashlib
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_J = None
_Z = None


def load():
    global _J, _Z
    if _J is None:
        with open(os.path.join(GOLD, "aa_cases.json")) as f:
            _J = json.load(f)
        _Z = np.load(os.path.join(GOLD, "aa_cases.npz"), allow_pickle=False)
    return _J, _Z


def cases(prefix=None):
    j, _ = load()
    return [c for c in j["cases"] if prefix is None or c["name"].startswith(prefix)]


def subs():
    return load()[0]["subs"]


def arr(name, key):
    z = load()[1]
    k = "%s/%s" % (name, key)
    return z[k] if k in z.files else None


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def f32bits(x):
    return int(np.float32(x).view(np.uint32))


def check_matrices(case, D, PQ, PT, S=None):
    """Bit-exact comparison with the reference's score / pointer matrices (full arrays if stored, else sha256)."""
    name = case["name"]
    H = arr(name, "H")
    if H is not None:
        assert np.array_equal(np.asarray(D, np.float32).view(np.uint32), H.view(np.uint32)), name + " H"
        assert np.array_equal(PQ, arr(name, "PQ")), name + " PQ"
        assert np.array_equal(PT, arr(name, "PT")), name + " PT"
        if S is not None and arr(name, "S") is not None:
            assert np.array_equal(np.asarray(S, np.float32).view(np.uint32), arr(name, "S").view(np.uint32)), name + " S"
    else:
        assert sha(np.asarray(D, np.float32)) == case["sha"]["H"], name + " H sha"
        assert sha(np.asarray(PQ, np.int32)) == case["sha"]["PQ"], name + " PQ sha"
        assert sha(np.asarray(PT, np.int32)) == case["sha"]["PT"], name + " PT sha"


def check_set(case, key, got, tstr=None, qstrs=None, annots=None):
    """got: list of dicts(score, uid, pairs[n,2]) in set order; compares with the golden set `key`."""
    ref = case["sets"][key]
    name = case["name"] + "/" + key
    assert len(got) == ref["n"], name + " size %d vs %d" % (len(got), ref["n"])
    if tstr is not None and "tstr" in ref:
        assert tstr == ref["tstr"], name + " tstr"
    for k, (g, r) in enumerate(zip(got, ref["alis"])):
        assert f32bits(g["score"]) == r["score"], "%s[%d] score" % (name, k)
        if "uid" in g:
            assert g["uid"] == r["uid"], "%s[%d] uid" % (name, k)
        p = np.asarray(g["pairs"], np.int32).reshape(-1, 2)
        if "pairs" in r:
            assert p.reshape(-1).tolist() == r["pairs"], "%s[%d] pairs" % (name, k)
        else:
            assert sha(p) == r["pairs_sha"], "%s[%d] pairs sha" % (name, k)
        if "identity" in g:
            assert f32bits(g["identity"]) == r["identity"], "%s[%d] identity" % (name, k)
        if qstrs is not None and "tstr" in ref:
            if "qstr" in r:
                assert qstrs[k] == r["qstr"], "%s[%d] qstr" % (name, k)
            elif "qstr_sha" in r:
                assert hashlib.sha256(qstrs[k].encode()).hexdigest() == r["qstr_sha"], "%s[%d] qstr sha" % (name, k)
        if annots is not None and "annot" in r:
            assert annots[k] == r["annot"], "%s[%d] annot" % (name, k)
