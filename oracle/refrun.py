"""TEST INFRASTRUCTURE — run oracle/_ref/ref_harness (the real reference, compiled in
place by oracle/Makefile) and parse its stdout."""
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HARNESS = os.path.join(HERE, "_ref", "ref_harness")
PROFILE = os.path.join(HERE, "_ref", "ref_profile")
BLOSUM62 = os.path.join(os.path.dirname(HERE), "tests", "golden", "BLOSUM62")


def available():
    return os.path.exists(HARNESS)


def _f(hexs):
    return struct.unpack("<f", struct.pack("<I", int(hexs, 16)))[0]


def _farr(toks):
    return np.array([int(x, 16) for x in toks], dtype=np.uint32).view(np.float32)


def parse(out):
    """-> dict with optional keys dim,H,PQ,PT,S,corner,sets{OPT|CW|UCW: {tstr, alis:[...]}},subali,throw"""
    res = {"sets": {}}
    cur = None
    for line in out.split("\n"):
        if not line:
            continue
        tag, _, rest = line.partition(" ")
        tk = rest.split()
        if tag == "DIM":
            res["dim"] = (int(tk[0]), int(tk[1]))
        elif tag in ("H", "S"):
            res[tag] = _farr(tk).reshape(res["dim"])
        elif tag in ("PQ", "PT"):
            res[tag] = np.array(tk, dtype=np.int32).reshape(res["dim"])
        elif tag == "CORNER":
            res["corner"] = (_f(tk[0]), _f(tk[1]))
        elif tag in ("OPT", "CW", "UCW"):
            cur = {"n": int(tk[0]), "alis": []}
            res["sets"][tag] = cur
        elif tag == "TSTR":
            cur["tstr"] = rest
        elif tag == "ALI":
            n = int(tk[3])
            pairs = np.array(tk[4:4 + 2 * n], dtype=np.int32).reshape(-1, 2)
            cur["alis"].append({"score": np.float32(_f(tk[0])), "identity": np.float32(_f(tk[1])), "uid": int(tk[2]), "pairs": pairs})
        elif tag == "QSTR":
            cur["alis"][-1]["qstr"] = rest
        elif tag == "ANNOT":
            cur["alis"][-1]["annot"] = rest
        elif tag == "SUBALI":
            n = int(tk[1])
            res["subali"] = {"score": np.float32(_f(tk[0])), "pairs": np.array(tk[2:2 + 2 * n], dtype=np.int32).reshape(-1, 2)}
        elif tag == "THROW":
            res["throw"] = rest
    return res


def run_aa(q, t, mode, gi, ge, direction="fwd", ops=("dump", "opt"), blosum=BLOSUM62, timeout=600):
    args = [HARNESS, "aa", blosum, str(mode), repr(float(gi)), repr(float(ge)), direction, q, t] + [str(o) for o in ops]
    out = subprocess.run(args, capture_output=True, text=True, timeout=timeout, check=True).stdout
    return parse(out)


def run_sub(q, t, mode, gi, ge, direction, q1, t1, q2, t2, blosum=BLOSUM62):
    args = [HARNESS, "sub", blosum, str(mode), repr(float(gi)), repr(float(ge)), direction, q, t, str(q1), str(t1), str(q2), str(t2)]
    out = subprocess.run(args, capture_output=True, text=True, timeout=600, check=True).stdout
    return parse(out)


def run_profile(qp, tp, mode, alpha, beta, zero_shift, gi, ge, direction=1, timeout=600, env=None):
    """qp/tp: dicts with float32 arrays aa[L,20], sse[L,3], conf[L] (sentinel rows included)."""
    def seq(p):
        L = len(p["conf"])
        lines = [str(L)]
        for i in range(L):
            olc = "^" if i == 0 else "$" if i == L - 1 else "A"
            vals = list(p["aa"][i]) + list(p["sse"][i]) + [p["conf"][i]]
            lines.append(olc + " " + " ".join("%.9g" % float(v) for v in vals))
        return "\n".join(lines)
    text = "%d %.9g %.9g %.9g %.9g %.9g %d\n%s\n%s\n" % (mode, alpha, beta, zero_shift, gi, ge, direction, seq(qp), seq(tp))
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([PROFILE], input=text, capture_output=True, text=True, timeout=timeout, check=True, env=e).stdout
    res = parse(out)
    for line in out.split("\n"):
        tag, _, rest = line.partition(" ")
        if tag in ("TGI", "TGE", "PRIM"):
            res[tag] = _farr(rest.split())
    return res
