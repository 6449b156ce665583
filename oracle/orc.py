"""TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GLOBAL_LOCAL, GLOBAL, LOCAL_GLOBAL, LOCAL, SEMI_LOCAL = 0, 1, 2, 3, 4
FWD, REV = 1, 2
GAP_AFFINE_CONST, GAP_AFFINE_TPOS_MIN, GAP_GN2, GAP_CALLBACK = 0, 1, 2, 3
GAPFN = C.CFUNCTYPE(C.c_float, C.c_int, C.c_int, C.c_int, C.c_int)


class OrcGap(C.Structure):
    _fields_ = [("model", C.c_int), ("align_type", C.c_int), ("gi", C.c_float), ("ge", C.c_float),
                ("tgi", C.POINTER(C.c_float)), ("tge", C.POINTER(C.c_float)), ("tcn", C.POINTER(C.c_float)),
                ("dist", C.POINTER(C.c_float)), ("vvgi", C.POINTER(C.c_float)), ("vvge", C.POINTER(C.c_float)),
                ("vvcd", C.POINTER(C.c_float)), ("del_cb", GAPFN), ("ins_cb", GAPFN)]


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orc_deletion.restype = C.c_float
        L.orc_insertion.restype = C.c_float
        L.orc_dot.restype = C.c_float
        L.orc_pearson.restype = C.c_float
        L.orc_set_new.restype = C.c_void_p
        for f in ("orc_set_free", "orc_set_size", "orc_set_push", "orc_set_npairs", "orc_set_get",
                  "orc_set_sort", "orc_set_identity"):
            getattr(L, f).argtypes = None
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def f32(x):
    return np.float32(x)


class Gap:
    """Gap model descriptor; keeps numpy arrays alive."""

    def __init__(self, align_type, gi=0.0, ge=0.0, tgi=None, tge=None, gn2=None, callbacks=None):
        self.g = OrcGap()
        self.g.align_type = int(align_type)
        if callbacks is not None:
            # (deletion(q1,q2,t1,t2), insertion(q1,q2,t1,t2)) python callables returning float: an arbitrary plugin's gap functions
            self.g.model = GAP_CALLBACK
            self.cbs = (GAPFN(callbacks[0]), GAPFN(callbacks[1]))
            self.g.del_cb, self.g.ins_cb = self.cbs
        elif gn2 is not None:
            # Gn2Eval tables: dict v_gi, v_ge, v_cn [T]; dist, vv_gi, vv_ge, vv_cd [T,T] indexed [p2, p1]
            self.g.model = GAP_GN2
            self.keep = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in gn2.items()}
            self.g.tgi, self.g.tge, self.g.tcn = _fp(self.keep["v_gi"]), _fp(self.keep["v_ge"]), _fp(self.keep["v_cn"])
            self.g.dist, self.g.vvgi = _fp(self.keep["dist"]), _fp(self.keep["vv_gi"])
            self.g.vvge, self.g.vvcd = _fp(self.keep["vv_ge"]), _fp(self.keep["vv_cd"])
        elif tgi is None:
            self.g.model = GAP_AFFINE_CONST
            self.g.gi = float(np.float32(gi))
            self.g.ge = float(np.float32(ge))
        else:
            self.g.model = GAP_AFFINE_TPOS_MIN
            self.tgi = np.ascontiguousarray(tgi, dtype=np.float32)
            self.tge = np.ascontiguousarray(tge, dtype=np.float32)
            self.g.tgi = _fp(self.tgi)
            self.g.tge = _fp(self.tge)

    @property
    def ref(self):
        return C.byref(self.g)


def deletion(gap, Q, T, q1, q2, t1, t2):
    """Evaluator::deletion as the oracle evaluates it (aasubalib.h:27-51 / hmap2_eval.h:41-67 / gn2_eval.h:100-130)."""
    err = C.c_int(0)
    return np.float32(lib().orc_deletion(gap.ref, int(Q), int(T), int(q1), int(q2), int(t1), int(t2), C.byref(err)))


def insertion(gap, Q, T, q1, q2, t1, t2):
    err = C.c_int(0)
    return np.float32(lib().orc_insertion(gap.ref, int(Q), int(T), int(q1), int(q2), int(t1), int(t2), C.byref(err)))


def load_blosum(path):
    """Parse a BLOSUM-format file the way submatrix.cpp:16-54 does."""
    lines = open(path).read().split("\n")
    k = 0
    while lines[k].startswith("#"):
        k += 1
    alphabet = "".join(ch for ch in lines[k] if ch not in " \n")
    n = len(alphabet)
    toks = " ".join(lines[k + 1:]).split()
    table = np.zeros((n, n), dtype=np.float32)
    p = 0
    for i in range(n):
        p += 1  # row label
        for j in range(n):
            table[i, j] = np.float32(float(toks[p]))
            p += 1
    return alphabet, table


def sim_submatrix(q, t, alphabet, table):
    """q, t: residue strings WITHOUT sentinels."""
    qs, ts = "^" + q + "$", "^" + t + "$"
    Q, T = len(qs), len(ts)
    S = np.zeros((Q, T), dtype=np.float32)
    tab = np.ascontiguousarray(table, dtype=np.float32)
    rc = lib().orc_sim_submatrix(Q, T, qs.encode(), ts.encode(), alphabet.encode(), len(alphabet), _fp(tab), _fp(S))
    if rc != 0:
        raise ValueError("orc_sim_submatrix rc=%d" % rc)
    return S


def dp_build(S, gap, direction=FWD, islocal=None, bounds=None, bug_b4=False):
    Q, T = S.shape
    if islocal is None:
        islocal = gap.g.align_type == LOCAL
    D = np.zeros((Q, T), dtype=np.float32)
    PQ = np.full((Q, T), -1, dtype=np.int32)
    PT = np.full((Q, T), -1, dtype=np.int32)
    q0, q1, t0, t1 = (0, Q - 1, 0, T - 1) if bounds is None else bounds
    Sc = np.ascontiguousarray(S, dtype=np.float32)
    rc = lib().orc_dp_build(Q, T, _fp(Sc), gap.ref, direction, int(bool(islocal)), q0, q1, t0, t1,
                            int(bool(bug_b4)), _fp(D), _ip(PQ), _ip(PT))
    return rc, D, PQ, PT


def optimal(D, PQ, PT, islocal, kind="fwd", sub=None):
    Q, T = D.shape
    pairs = np.zeros(2 * (Q + T + 4), dtype=np.int32)
    n = C.c_int(0)
    sc = C.c_float(0)
    D = np.ascontiguousarray(D, dtype=np.float32)
    PQ = np.ascontiguousarray(PQ, dtype=np.int32)
    PT = np.ascontiguousarray(PT, dtype=np.int32)
    if sub is not None:
        rc = lib().orc_optimal_subali(Q, T, _fp(D), _ip(PQ), _ip(PT), sub[0], sub[1], sub[2], sub[3],
                                      _ip(pairs), C.byref(n), C.byref(sc))
    elif kind == "rev":
        rc = lib().orc_optimal_rev(Q, T, _fp(D), _ip(PQ), _ip(PT), int(bool(islocal)), _ip(pairs), C.byref(n), C.byref(sc))
    else:
        rc = lib().orc_optimal(Q, T, _fp(D), _ip(PQ), _ip(PT), int(bool(islocal)), _ip(pairs), C.byref(n), C.byref(sc))
    return rc, np.float32(sc.value), pairs[:2 * n.value].reshape(-1, 2).copy()


class AliSet:
    def __init__(self):
        self.h = C.c_void_p(lib().orc_set_new())

    def __del__(self):
        try:
            lib().orc_set_free(self.h)
        except Exception:
            pass

    def push(self, pairs, score, uid=-1):
        p = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1)
        lib().orc_set_push(self.h, _ip(p), C.c_int(len(p) // 2), C.c_float(float(score)), C.c_int(uid))

    def __len__(self):
        return lib().orc_set_size(self.h)

    def get(self, k):
        n = lib().orc_set_npairs(self.h, C.c_int(k))
        p = np.zeros(2 * n, dtype=np.int32)
        sc, idn, uid = C.c_float(0), C.c_float(0), C.c_int(0)
        lib().orc_set_get(self.h, C.c_int(k), _ip(p), C.byref(sc), C.byref(idn), C.byref(uid))
        return dict(score=np.float32(sc.value), identity=np.float32(idn.value), uid=uid.value, pairs=p.reshape(-1, 2))

    def sort(self, mx):
        lib().orc_set_sort(self.h, C.c_int(mx))

    def identity(self, q, t):
        lib().orc_set_identity(self.h, ("^" + q + "$").encode(), ("^" + t + "$").encode())

    def strings(self, q, t):
        qs, ts = "^" + q + "$", "^" + t + "$"
        Q, T = len(qs), len(ts)
        stride = lib().orc_gapped_len(self.h, C.c_int(T)) + 8
        n = len(self)
        tl = C.create_string_buffer(stride)
        ql = C.create_string_buffer(stride * max(n, 1))
        rc = lib().orc_gapped_strings(self.h, Q, T, qs.encode(), ts.encode(), tl, ql, stride)
        if rc != 0:
            raise ValueError("orc_gapped_strings rc=%d" % rc)
        raw = ql.raw
        return tl.value.decode(), [raw[k * stride:(k + 1) * stride].split(b"\0")[0].decode() for k in range(n)]


def enumerate_noa(kind, D, PQ, PT, S, gap, flags, number_suboptimal, delta_ratio, aset, user_limit=0):
    Q, T = D.shape
    fl = np.ascontiguousarray(flags if flags is not None else np.ones(T), dtype=np.uint8)
    D = np.ascontiguousarray(D, dtype=np.float32)
    PQ = np.ascontiguousarray(PQ, dtype=np.int32)
    PT = np.ascontiguousarray(PT, dtype=np.int32)
    S = np.ascontiguousarray(S, dtype=np.float32)
    return lib().orc_enumerate(0 if kind == "cw" else 1, Q, T, _fp(D), _ip(PQ), _ip(PT), _fp(S), gap.ref,
                               fl.ctypes.data_as(C.POINTER(C.c_ubyte)), int(number_suboptimal),
                               C.c_float(float(np.float32(delta_ratio))), C.c_uint(user_limit), aset.h)


def enumerate_ks(D, PQ, PT, S, gap, flags, number_suboptimal, delta_ratio, k_limit, aset, user_limit=100000):
    """KSConstrainedNearOptimal (kscw.h:109-351); NOaliParams defaults k_limit 16, user_limit 100000 (noalib.cpp:17-20)."""
    Q, T = D.shape
    fl = np.ascontiguousarray(flags if flags is not None else np.ones(T), dtype=np.uint8)
    D = np.ascontiguousarray(D, dtype=np.float32)
    PQ = np.ascontiguousarray(PQ, dtype=np.int32)
    PT = np.ascontiguousarray(PT, dtype=np.int32)
    S = np.ascontiguousarray(S, dtype=np.float32)
    return lib().orc_enumerate_ks(Q, T, _fp(D), _ip(PQ), _ip(PT), _fp(S), gap.ref, fl.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                  int(number_suboptimal), C.c_float(float(np.float32(delta_ratio))), C.c_uint(k_limit),
                                  C.c_uint(user_limit), aset.h)


def enumerate_cr(D, PQ, PT, S, gap, flags, number_suboptimal, delta_ratio, k_limit, aset, sort_limit=100, user_limit=100000,
                 max_overlap=0.30):
    """CRConstrainedNearOptimal (crcw.h:134-594); NOaliParams defaults k_limit 16, sort_limit 100, user_limit 100000, max_overlap 0.30
    (noalib.cpp:15-21).  -> (status, times the reference would have read regions[-1])"""
    Q, T = D.shape
    fl = np.ascontiguousarray(flags if flags is not None else np.ones(T), dtype=np.uint8)
    D = np.ascontiguousarray(D, dtype=np.float32)
    PQ = np.ascontiguousarray(PQ, dtype=np.int32)
    PT = np.ascontiguousarray(PT, dtype=np.int32)
    S = np.ascontiguousarray(S, dtype=np.float32)
    oob = C.c_long(0)
    rc = lib().orc_enumerate_cr(Q, T, _fp(D), _ip(PQ), _ip(PT), _fp(S), gap.ref, fl.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                int(number_suboptimal), C.c_float(float(np.float32(delta_ratio))), C.c_uint(k_limit), C.c_uint(sort_limit),
                                C.c_uint(user_limit), C.c_float(float(np.float32(max_overlap))), aset.h, C.byref(oob))
    return rc, oob.value


def annot(score, identity, significance=9999.0):
    """FASTA annotation of fastaio.h:79-91 (ostream default precision == %g)."""
    def g(x):
        x = float(x)
        if x != x:  # glibc prints the NaN sign bit
            return "-nan" if np.signbit(x) else "nan"
        return "%g" % x
    return "(sc=%s,ev=%s,id=%s%%)" % (g(score), g(significance), g(identity))


def make_subopt_regions(T, regs):
    """gn2.cpp:268-283 — evenly divide T template positions into `regs` alternating regions."""
    length = np.float32(T) / np.float32(regs)
    flags = np.zeros(T, dtype=np.uint8)
    flag = True
    place = np.float32(length)
    for i in range(T):
        flags[i] = flag
        if np.float32(i) > place:
            flag = not flag
            place = np.float32(place + length)
    flags[T - 1] = 1
    return flags


def hmap2_sim(qp, tp, alpha, zero_shift=None):
    """Hmap2Eval similarity matrix (+ post_process when zero_shift is given)."""
    Q, T = len(qp["conf"]), len(tp["conf"])
    S = np.zeros((Q, T), dtype=np.float32)
    arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (qp["aa"], qp["sse"], qp["conf"], tp["aa"], tp["sse"], tp["conf"])]
    lib().orc_sim_hmap2(Q, T, *[_fp(a) for a in arrs], C.c_float(float(np.float32(alpha))), _fp(S))
    if zero_shift is not None:
        lib().orc_norm_shift(Q, T, _fp(S), C.c_float(float(np.float32(zero_shift))))
    return S


def hmap2_precalc(tp, gi, ge, beta):
    T = len(tp["conf"])
    pc = np.ascontiguousarray(tp["sse"][:, 2], dtype=np.float32)
    tgi = np.zeros(T, dtype=np.float32)
    tge = np.zeros(T, dtype=np.float32)
    lib().orc_hmap2_precalc(T, _fp(pc), C.c_float(float(np.float32(gi))), C.c_float(float(np.float32(ge))),
                            C.c_float(float(np.float32(beta))), _fp(tgi), _fp(tge))
    return tgi, tge
