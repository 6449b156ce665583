"""aln_amd — thin ctypes binding of libalnhip.so (include/aln_hip.h), used by tests/ and bench.py.

The product is the C-ABI shared library built from csrc/*.hip for gfx950; this module only loads it
and marshals numpy arrays.  There is NO CPU fallback: if the library is missing or there is no GPU the
calls fail loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "libalnhip.so")

GLOBAL_LOCAL, GLOBAL, LOCAL_GLOBAL, LOCAL, SEMI_LOCAL = 0, 1, 2, 3, 4
FWD, REV = 1, 2
GAP_AFFINE_CONST, GAP_AFFINE_TPOS_MIN, GAP_DEL_TABLE_INS_TPOS, GAP_TABLES = 0, 1, 2, 3
SIM_SUBMATRIX, SIM_MATRIX, SIM_HMAP2 = 0, 1, 2
DP_AUTO, DP_EXACT, DP_FAST = 0, 1, 2
ENUM_CW, ENUM_UCW, ENUM_KSCW, ENUM_CRCW = 0, 1, 2, 3
_ENUM = {"cw": ENUM_CW, "ucw": ENUM_UCW, "kscw": ENUM_KSCW, "crcw": ENUM_CRCW}

E_BOUNDS, E_GAPSTYLE, E_STARTPAIR, E_RESIDUE, E_ARG, E_HIP, E_NOMEM, E_TOO_LONG, E_NOT_INTEGRAL, E_STATE, E_OVERFLOW = range(-1, -12, -1)

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class AlnSeqs(C.Structure):
    _fields_ = [("n_seqs", C.c_int32), ("offsets", _lp), ("residues", C.c_char_p)]


class AlnSubmatrix(C.Structure):
    _fields_ = [("n", C.c_int32), ("alphabet", C.c_char_p), ("table", _fp)]


class AlnProfiles(C.Structure):
    _fields_ = [("aa", _fp), ("sse", _fp), ("conf", _fp)]


class AlnGap(C.Structure):
    _fields_ = [("model", C.c_int32), ("align_type", C.c_int32), ("gap_init", C.c_float), ("gap_extn", C.c_float),
                ("t_gap_init", _fp), ("t_gap_extn", _fp), ("dp_local", C.c_int32), ("t_gap_cn", _fp), ("del_table", _fp),
                ("del_table_off", _lp), ("ins_tables", _fp), ("ins_table_off", _lp)]


class AlnSim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("sub", AlnSubmatrix), ("planes", _fp), ("plane_off", _lp),
                ("q_prof", AlnProfiles), ("t_prof", AlnProfiles), ("alpha", C.c_float), ("zero_shift", C.c_float),
                ("normalize", C.c_int32)]


class AlnNoa(C.Structure):
    _fields_ = [("kind", C.c_int32), ("number_suboptimal", C.c_int32), ("delta_ratio", C.c_float), ("user_limit", C.c_uint32),
                ("n_existing", C.c_int32), ("existing_scores", _fp), ("k_limit", C.c_uint32), ("sort_limit", C.c_uint32),
                ("max_overlap", C.c_float)]


class AlnAlignment(C.Structure):
    _fields_ = [("score", C.c_float), ("identity", C.c_float), ("uid", C.c_int32), ("n_pairs", C.c_int32), ("pair_off", C.c_int64)]


EXPORTS = [
    "aln_ctx_create", "aln_ctx_destroy", "aln_error_string", "aln_last_error", "aln_ctx_synchronize", "aln_has_gfx950",
    "aln_batch_create", "aln_batch_destroy", "aln_batch_n_pairs", "aln_batch_device_bytes", "aln_batch_dp",
    "aln_batch_reevaluate", "aln_batch_dp_kernel_name", "aln_batch_dp_sub", "aln_batch_get_cells", "aln_batch_get_sim",
    "aln_batch_get_corner_scores", "aln_batch_optimal", "aln_batch_optimal_enqueue", "aln_batch_optimal_collect", "aln_batch_optimal_subali", "aln_batch_enumerate", "aln_batch_enumerate_all", "aln_batch_last_enum_ms", "aln_batch_last_enum_usage", "aln_identity",
    "aln_gapped_length", "aln_gapped_strings", "aln_hmap2_gap_arrays", "aln_score_all_vs_all", "aln_batch_last_dp_ms", "aln_batch_dp_ms_history", "aln_batch_dp_algorithmic_bytes", "aln_batch_cells",
    "aln_batch_optimal_strings", "aln_batch_optimal_strings_enqueue", "aln_batch_optimal_strings_collect", "aln_batch_last_exact_stats", "aln_batch_set_gap", "aln_ctx_set_hint", "aln_ctx_get_hint", "aln_batch_dp_contract_bytes", "aln_batch_plane_bytes_per_cell",
    "aln_deal_units", "aln_comm_unique_id", "aln_comm_create", "aln_ctx_create_multi", "aln_comm_destroy", "aln_comm_n_ranks",
    "aln_comm_last_error", "aln_gather_scores",
]
COMM_ID_BYTES = 128

_LIB = None


class AlnError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "%s (aln status %d)" % (msg, code))
        self.code = code


def build_library():
    subprocess.check_call(["make", "-s", "-C", PKG_DIR, "-j8"])


def lib():
    """Load libalnhip.so (never a fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libalnhip.so not built: run `make -C alignment-algos_amd` (hipcc --offload-arch=gfx950)")
        L = C.CDLL(LIB_PATH)
        L.aln_error_string.restype = C.c_char_p
        L.aln_last_error.restype = C.c_char_p
        L.aln_last_error.argtypes = [C.c_void_p]
        L.aln_batch_dp_kernel_name.restype = C.c_char_p
        L.aln_batch_dp_kernel_name.argtypes = [C.c_void_p]
        L.aln_identity.restype = C.c_float
        for f in ("aln_batch_device_bytes", "aln_batch_dp_algorithmic_bytes", "aln_batch_dp_contract_bytes", "aln_batch_cells"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        L.aln_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.aln_ctx_destroy.argtypes = [C.c_void_p]
        L.aln_ctx_synchronize.argtypes = [C.c_void_p]
        L.aln_batch_create.argtypes = [C.c_void_p, C.POINTER(AlnSeqs), C.POINTER(AlnSeqs), C.c_int32, _ip, _ip, C.c_int32, C.POINTER(C.c_void_p)]
        L.aln_batch_destroy.argtypes = [C.c_void_p]
        L.aln_batch_n_pairs.argtypes = [C.c_void_p]
        L.aln_batch_dp.argtypes = [C.c_void_p, C.POINTER(AlnSim), C.POINTER(AlnGap), C.c_int32, C.c_int32, C.c_int32]
        L.aln_batch_reevaluate.argtypes = [C.c_void_p]
        L.aln_batch_set_gap.argtypes = [C.c_void_p, C.POINTER(AlnGap)]
        L.aln_batch_dp_sub.argtypes = [C.c_void_p, C.POINTER(AlnSim), C.POINTER(AlnGap), C.c_int32, _ip]
        L.aln_batch_get_cells.argtypes = [C.c_void_p, C.c_int32, _fp, _ip, _ip]
        L.aln_batch_get_sim.argtypes = [C.c_void_p, C.c_int32, _fp]
        L.aln_batch_get_corner_scores.argtypes = [C.c_void_p, _fp]
        L.aln_batch_optimal.argtypes = [C.c_void_p, _fp, _ip, _ip, C.c_int32, _ip]
        L.aln_batch_optimal_enqueue.argtypes = [C.c_void_p]
        L.aln_batch_optimal_collect.argtypes = [C.c_void_p, _fp, _ip, _ip]
        L.aln_batch_optimal_subali.argtypes = [C.c_void_p, _fp, _ip, _ip, C.c_int32, _ip]
        L.aln_batch_enumerate.argtypes = [C.c_void_p, C.c_int32, C.POINTER(AlnNoa), C.POINTER(C.c_uint8), C.POINTER(AlnAlignment),
                                          C.c_int32, _ip, C.c_int64, _ip]
        L.aln_batch_enumerate_all.argtypes = [C.c_void_p, C.POINTER(AlnNoa), C.POINTER(C.c_uint8), C.c_int32, C.c_uint32, C.c_uint32,
                                              C.c_int32, _ip, _fp, _ip, _ip, C.c_int32, _ip]
        L.aln_batch_last_enum_ms.argtypes = [C.c_void_p, _fp, _fp]
        L.aln_batch_last_enum_usage.argtypes = [C.c_void_p, _ip, _ip]
        L.aln_identity.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, _ip, C.c_int32]
        L.aln_gapped_length.argtypes = [C.c_int32, C.POINTER(AlnAlignment), C.c_int32, _ip]
        L.aln_gapped_strings.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(AlnAlignment), C.c_int32, _ip,
                                         C.c_char_p, C.c_char_p, C.c_int32]
        L.aln_batch_last_dp_ms.argtypes = [C.c_void_p, _fp]
        L.aln_batch_dp_ms_history.argtypes = [C.c_void_p, _fp, C.c_int32]
        L.aln_score_all_vs_all.argtypes = [C.c_void_p, C.POINTER(AlnSeqs), C.POINTER(AlnSeqs), C.POINTER(AlnSubmatrix), C.POINTER(AlnGap),
                                           C.c_int32, C.c_int32, _fp]
        L.aln_hmap2_gap_arrays.argtypes = [_fp, C.c_int64, C.c_float, C.c_float, C.c_float, _fp, _fp]
        L.aln_batch_plane_bytes_per_cell.argtypes = [C.c_void_p]
        L.aln_batch_optimal_strings.argtypes = [C.c_void_p, _fp, _fp, _ip, C.c_char_p, C.c_char_p, C.c_int32, _ip]
        L.aln_batch_last_exact_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.aln_batch_optimal_strings_enqueue.argtypes = [C.c_void_p, C.c_int32]
        L.aln_batch_optimal_strings_collect.argtypes = [C.c_void_p, _fp, _fp, _ip, C.c_char_p, C.c_char_p, C.c_int32, _ip]
        L.aln_ctx_set_hint.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.aln_ctx_get_hint.argtypes = [C.c_void_p, C.c_char_p, _lp]
        L.aln_deal_units.argtypes = [_lp, C.c_int64, C.c_int32, _ip, _ip]
        L.aln_comm_unique_id.argtypes = [C.c_void_p]
        L.aln_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.aln_ctx_create_multi.argtypes = [_ip, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.aln_comm_destroy.argtypes = [C.c_void_p]
        L.aln_comm_n_ranks.argtypes = [C.c_void_p]
        L.aln_comm_last_error.argtypes = [C.c_void_p]
        L.aln_comm_last_error.restype = C.c_char_p
        L.aln_gather_scores.argtypes = [C.c_void_p, C.POINTER(_fp), C.POINTER(_ip), _ip, C.c_int32, _fp, C.c_int64]
        _LIB = L
    return _LIB


def _check(rc, ctx=None):
    if rc != 0:
        msg = lib().aln_error_string(rc).decode()
        if ctx is not None and rc == E_HIP:
            msg += ": " + lib().aln_last_error(ctx).decode()
        raise AlnError(rc, msg)


def _f(a):
    return a.ctypes.data_as(_fp)


def _i(a):
    return a.ctypes.data_as(_ip)


class SeqPool:
    """A pool of sequences; each entry is given WITHOUT sentinels and stored as '^' + s + '$'."""

    def __init__(self, seqs):
        self.seqs = ["^" + s + "$" for s in seqs]
        self.offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum([len(s) for s in self.seqs], out=self.offsets[1:])
        self.blob = "".join(self.seqs).encode()
        self.c = AlnSeqs(len(seqs), self.offsets.ctypes.data_as(_lp), self.blob)


class Context:
    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        self.device = device
        _check(lib().aln_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self.h)))

    def synchronize(self):
        _check(lib().aln_ctx_synchronize(self.h), self.h)

    def set_hint(self, key, value):
        """aln_ctx_set_hint: a tuning / kernel-selection switch of this context (never changes a result)."""
        _check(lib().aln_ctx_set_hint(self.h, key.encode(), int(value)), self.h)

    def get_hint(self, key):
        v = C.c_int64(0)
        _check(lib().aln_ctx_get_hint(self.h, key.encode(), C.byref(v)), self.h)
        return v.value

    def hints(self, **kv):
        """Context manager: set hints, restore the previous values on exit."""
        ctx = self

        class _H:
            def __enter__(self_):
                self_.old = {k: ctx.get_hint(k) for k in kv}
                for k, v in kv.items():
                    ctx.set_hint(k, v)

            def __exit__(self_, *a):
                for k, v in self_.old.items():
                    ctx.set_hint(k, v)
        return _H()

    def close(self):
        if self.h:
            lib().aln_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def score_all_vs_all(ctx, queries, templates, alphabet, table, gi, ge, q_begin=0, q_end=None, align_type=LOCAL):
    """The score Optimal(align_type) reports for queries[q_begin:q_end] against every template, no planes (aln_score_all_vs_all)."""
    qpool = queries if isinstance(queries, SeqPool) else SeqPool(queries)
    tpool = templates if isinstance(templates, SeqPool) else SeqPool(templates)
    if q_end is None:
        q_end = len(qpool.seqs)
    tab = np.ascontiguousarray(table, dtype=np.float32)
    ab = alphabet.encode()
    sub = AlnSubmatrix(len(alphabet), ab, _f(tab))
    g = AlnGap()
    g.model = GAP_AFFINE_CONST
    g.align_type = int(align_type)
    g.gap_init = float(np.float32(gi))
    g.gap_extn = float(np.float32(ge))
    out = np.empty((q_end - q_begin, len(tpool.seqs)), dtype=np.float32)
    _check(lib().aln_score_all_vs_all(ctx.h, C.byref(qpool.c), C.byref(tpool.c), C.byref(sub), C.byref(g), q_begin, q_end, _f(out)), ctx.h)
    return out


class Batch:
    """Many DPMatrix objects resident in HBM (aln_batch)."""

    def __init__(self, ctx, queries, templates, q_idx=None, t_idx=None, score_only=False):
        self.ctx = ctx
        self.qpool = queries if isinstance(queries, SeqPool) else SeqPool(queries)
        self.tpool = templates if isinstance(templates, SeqPool) else SeqPool(templates)
        if q_idx is None:
            q_idx = np.arange(len(self.qpool.seqs))
            t_idx = np.arange(len(self.tpool.seqs))
        self.q_idx = np.ascontiguousarray(q_idx, dtype=np.int32)
        self.t_idx = np.ascontiguousarray(t_idx, dtype=np.int32)
        self.n = len(self.q_idx)
        self.h = C.c_void_p()
        _check(lib().aln_batch_create(ctx.h, C.byref(self.qpool.c), C.byref(self.tpool.c), self.n, _i(self.q_idx), _i(self.t_idx),
                                      int(score_only), C.byref(self.h)), ctx.h)
        self._keep = []

    def close(self):
        if self.h:
            lib().aln_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dims(self, p):
        return len(self.qpool.seqs[self.q_idx[p]]), len(self.tpool.seqs[self.t_idx[p]])

    # --- DP ---------------------------------------------------------------------------------
    def _gap(self, align_type, gi, ge, tgi=None, tge=None, tcn=None, del_tables=None, ins_tables=None):
        g = AlnGap()
        g.align_type = int(align_type)
        g.gap_init = float(np.float32(gi))
        g.gap_extn = float(np.float32(ge))
        if ins_tables is not None:
            # fully tabulated gap functions: one T x T deletion table per template sequence, three T x Q insertion planes per pair
            g.model = GAP_TABLES
            dflat = [np.ascontiguousarray(t, dtype=np.float32).reshape(-1) for t in del_tables]
            doff = np.zeros(len(dflat), dtype=np.int64)
            doff[1:] = np.cumsum([len(t) for t in dflat])[:-1]
            iflat = [np.ascontiguousarray(t, dtype=np.float32).reshape(-1) for t in ins_tables]
            ioff = np.zeros(len(iflat), dtype=np.int64)
            ioff[1:] = np.cumsum([len(t) for t in iflat])[:-1]
            dblob, iblob = np.concatenate(dflat), np.concatenate(iflat)
            self._keep += [dblob, doff, iblob, ioff]
            g.del_table, g.del_table_off = _f(dblob), doff.ctypes.data_as(_lp)
            g.ins_tables, g.ins_table_off = _f(iblob), ioff.ctypes.data_as(_lp)
        elif del_tables is not None:
            # Gn2Eval model: v_gi / v_ge / v_cn over the template pool + one T x T deletion table per template sequence
            g.model = GAP_DEL_TABLE_INS_TPOS
            a, b, c = (np.ascontiguousarray(x, dtype=np.float32) for x in (tgi, tge, tcn))
            flat = [np.ascontiguousarray(t, dtype=np.float32).reshape(-1) for t in del_tables]
            off = np.zeros(len(flat), dtype=np.int64)
            off[1:] = np.cumsum([len(t) for t in flat])[:-1]
            blob = np.concatenate(flat)
            self._keep += [a, b, c, blob, off]
            g.t_gap_init, g.t_gap_extn, g.t_gap_cn = _f(a), _f(b), _f(c)
            g.del_table, g.del_table_off = _f(blob), off.ctypes.data_as(_lp)
        elif tgi is not None:
            g.model = GAP_AFFINE_TPOS_MIN
            a = np.ascontiguousarray(tgi, dtype=np.float32)
            b = np.ascontiguousarray(tge, dtype=np.float32)
            self._keep += [a, b]
            g.t_gap_init, g.t_gap_extn = _f(a), _f(b)
        else:
            g.model = GAP_AFFINE_CONST
        return g

    def dp_submatrix(self, alphabet, table, align_type, gi, ge, direction=FWD, algo=DP_AUTO, bug_b4=False, tgi=None, tge=None):
        s = AlnSim()
        s.kind = SIM_SUBMATRIX
        tab = np.ascontiguousarray(table, dtype=np.float32)
        ab = alphabet.encode()
        s.sub = AlnSubmatrix(len(alphabet), ab, _f(tab))
        g = self._gap(align_type, gi, ge, tgi, tge)
        _check(lib().aln_batch_dp(self.h, C.byref(s), C.byref(g), direction, algo, int(bug_b4)), self.ctx.h)

    def dp_simmatrix(self, planes, align_type, gi, ge, direction=FWD, algo=DP_AUTO, bug_b4=False, tgi=None, tge=None, tcn=None,
                     del_tables=None, ins_tables=None):
        """planes: list of Q x T float32 arrays, one per pair."""
        flat = [np.ascontiguousarray(p, dtype=np.float32).reshape(-1) for p in planes]
        off = np.zeros(len(flat) + 1, dtype=np.int64)
        np.cumsum([len(p) for p in flat], out=off[1:])
        blob = np.concatenate(flat) if flat else np.zeros(1, np.float32)
        s = AlnSim()
        s.kind = SIM_MATRIX
        s.planes = _f(blob)
        s.plane_off = off.ctypes.data_as(_lp)
        g = self._gap(align_type, gi, ge, tgi, tge, tcn, del_tables, ins_tables)
        _check(lib().aln_batch_dp(self.h, C.byref(s), C.byref(g), direction, algo, int(bug_b4)), self.ctx.h)

    def dp_sub_submatrix(self, alphabet, table, align_type, gi, ge, direction, bounds):
        """7-argument DPMatrix ctor: bounds[p] = (q1_end, t1_end, q2_beg, t2_beg)."""
        s = AlnSim()
        s.kind = SIM_SUBMATRIX
        tab = np.ascontiguousarray(table, dtype=np.float32)
        ab = alphabet.encode()
        s.sub = AlnSubmatrix(len(alphabet), ab, _f(tab))
        g = self._gap(align_type, gi, ge)
        bd = np.ascontiguousarray(bounds, dtype=np.int32).reshape(-1)
        _check(lib().aln_batch_dp_sub(self.h, C.byref(s), C.byref(g), direction, _i(bd)), self.ctx.h)

    def dp_hmap2(self, qprof, tprof, align_type, gi, ge, alpha=0.5, beta=1.0, zero_shift=0.12, normalize=True, direction=FWD,
                 algo=DP_AUTO):
        """Hmap2Eval on the device.  qprof/tprof: dicts aa[n,20], sse[n,3], conf[n] over the POOLS (sentinels included).
        pre_calculate (gap arrays from p_coil) runs on the host through aln_hmap2_gap_arrays."""
        arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (qprof["aa"], qprof["sse"], qprof["conf"], tprof["aa"], tprof["sse"], tprof["conf"])]
        nt = len(arrs[5])
        tgi = np.zeros(nt, dtype=np.float32)
        tge = np.zeros(nt, dtype=np.float32)
        _check(lib().aln_hmap2_gap_arrays(_f(arrs[4]), nt, float(np.float32(gi)), float(np.float32(ge)), float(np.float32(beta)), _f(tgi), _f(tge)))
        s = AlnSim()
        s.kind = SIM_HMAP2
        s.q_prof = AlnProfiles(_f(arrs[0]), _f(arrs[1]), _f(arrs[2]))
        s.t_prof = AlnProfiles(_f(arrs[3]), _f(arrs[4]), _f(arrs[5]))
        s.alpha = float(np.float32(alpha))
        s.zero_shift = float(np.float32(zero_shift))
        s.normalize = int(bool(normalize))
        g = self._gap(align_type, 0, 0, tgi, tge)
        _check(lib().aln_batch_dp(self.h, C.byref(s), C.byref(g), direction, algo, 0), self.ctx.h)
        return tgi, tge

    def set_gap(self, align_type, gi=0, ge=0, tgi=None, tge=None, tcn=None, del_tables=None, ins_tables=None):
        """aln_batch_set_gap: new gap parameters for the resident batch (similarity untouched); follow with reevaluate()."""
        g = self._gap(align_type, gi, ge, tgi, tge, tcn, del_tables, ins_tables)
        _check(lib().aln_batch_set_gap(self.h, C.byref(g)), self.ctx.h)

    def reevaluate(self):
        _check(lib().aln_batch_reevaluate(self.h), self.ctx.h)

    def kernel_name(self):
        return lib().aln_batch_dp_kernel_name(self.h).decode()

    def last_exact_stats(self):
        """hint exact_debug: far chunks (tested, skipped) of the deletion and the insertion scans of the last tiled exact build"""
        out = (C.c_uint64 * 4)()
        _check(lib().aln_batch_last_exact_stats(self.h, out), self.ctx.h)
        return [int(x) for x in out]

    def last_dp_ms(self):
        ms = C.c_float(0)
        _check(lib().aln_batch_last_dp_ms(self.h, C.byref(ms)), self.ctx.h)
        return ms.value

    def dp_ms_history(self, n):
        ms = np.zeros(n, dtype=np.float32)
        got = lib().aln_batch_dp_ms_history(self.h, _f(ms), n)
        if got < 0:
            raise AlnError(E_HIP, "aln_batch_dp_ms_history")
        return ms[:got]

    def cells(self):
        return lib().aln_batch_cells(self.h)

    def algorithmic_bytes(self):
        """bytes one DP launch must write with the chosen plane layout (4, 6 or 8 B per matrix cell)"""
        return lib().aln_batch_dp_algorithmic_bytes(self.h)

    def contract_bytes(self):
        """SURVEY 8(d)'s figure: 8 B per matrix cell (fp32 score + 32-bit pointer), whatever layout was chosen"""
        return lib().aln_batch_dp_contract_bytes(self.h)

    def plane_bytes_per_cell(self):
        return lib().aln_batch_plane_bytes_per_cell(self.h)

    def device_bytes(self):
        return lib().aln_batch_device_bytes(self.h)

    # --- results -----------------------------------------------------------------------------
    def get_cells(self, p):
        Q, T = self.dims(p)
        D = np.empty((Q, T), dtype=np.float32)
        PQ = np.empty((Q, T), dtype=np.int32)
        PT = np.empty((Q, T), dtype=np.int32)
        _check(lib().aln_batch_get_cells(self.h, p, _f(D), _i(PQ), _i(PT)), self.ctx.h)
        return D, PQ, PT

    def get_sim(self, p):
        Q, T = self.dims(p)
        S = np.empty((Q, T), dtype=np.float32)
        _check(lib().aln_batch_get_sim(self.h, p, _f(S)), self.ctx.h)
        return S

    def corner_scores(self):
        s = np.empty(self.n, dtype=np.float32)
        _check(lib().aln_batch_get_corner_scores(self.h, _f(s)), self.ctx.h)
        return s

    def enumerate(self, p, kind, number_suboptimal, delta_ratio, flags=None, user_limit=0, max_alignments=None, pairs_capacity=None,
                  k_limit=0, sort_limit=0, max_overlap=0.30):
        """ConstrainedNearOptimal ("cw") / UnconstrainedNearOptimal ("ucw") / KSConstrainedNearOptimal ("kscw") /
        CRConstrainedNearOptimal ("crcw") for pair p -> list of dicts in set order."""
        Q, T = self.dims(p)
        noa = AlnNoa(_ENUM[kind], int(number_suboptimal), float(np.float32(delta_ratio)),
                     int(user_limit), -1, None, int(k_limit), int(sort_limit), float(np.float32(max_overlap)))
        if max_alignments is None:
            max_alignments = max(int(number_suboptimal), 1) + 2
        if pairs_capacity is None:
            pairs_capacity = max_alignments * (min(Q, T) + 3)
        out = (AlnAlignment * max_alignments)()
        pairs = np.zeros((pairs_capacity, 2), dtype=np.int32)
        n = C.c_int32(0)
        fl = None
        if flags is not None:
            fl = np.ascontiguousarray(flags, dtype=np.uint8)
            assert len(fl) == T
        _check(lib().aln_batch_enumerate(self.h, p, C.byref(noa), fl.ctypes.data_as(C.POINTER(C.c_uint8)) if fl is not None else None,
                                         out, max_alignments, _i(pairs), pairs_capacity, C.byref(n)), self.ctx.h)
        res = []
        for k in range(n.value):
            a = out[k]
            res.append({"score": np.float32(a.score), "identity": np.float32(a.identity), "uid": a.uid,
                        "pairs": pairs[a.pair_off:a.pair_off + a.n_pairs].copy()})
        return res

    def enumerate_all(self, kind, number_suboptimal, delta_ratio, flags=None, K=None, user_limit=0, node_cap=0, ali_cap=0,
                      want_pairs=True, raise_on_overflow=True, k_limit=0, sort_limit=0, max_overlap=0.30):
        """aln_batch_enumerate_all: every pair of the batch in one launch.  flags: None, one shared row, or an
        [n, stride] uint8 array.  -> n_out[n], scores[n,K], lengths[n,K], pairs[n,K,stride,2] or None, status[n]"""
        noa = AlnNoa(_ENUM[kind], int(number_suboptimal), float(np.float32(delta_ratio)),
                     int(user_limit), -1, None, int(k_limit), int(sort_limit), float(np.float32(max_overlap)))
        if K is None:
            K = max(int(number_suboptimal), 1) + 2
        fl, fstride = None, 0
        if flags is not None:
            fl = np.ascontiguousarray(flags, dtype=np.uint8)
            fstride = fl.shape[1] if fl.ndim == 2 else 0
        if not hasattr(self, "_stride"):
            self._stride = max(min(max(self.dims(p)[0] for p in range(self.n)), max(self.dims(p)[1] for p in range(self.n))) + 3, 4) if self.n else 4
        stride = self._stride
        n_out = np.zeros(self.n, dtype=np.int32)
        scores = np.zeros((self.n, K), dtype=np.float32)
        lengths = np.zeros((self.n, K), dtype=np.int32)
        status = np.zeros(self.n, dtype=np.int32)
        pairs = np.zeros((self.n, K, stride, 2), dtype=np.int32) if want_pairs else None
        rc = lib().aln_batch_enumerate_all(self.h, C.byref(noa), fl.ctypes.data_as(C.POINTER(C.c_uint8)) if fl is not None else None, fstride,
                                           int(node_cap), int(ali_cap), K, _i(n_out), _f(scores), _i(lengths),
                                           _i(pairs) if want_pairs else None, stride, _i(status))
        if rc != 0 and (raise_on_overflow or rc != E_OVERFLOW):
            _check(rc, self.ctx.h)
        return n_out, scores, lengths, pairs, status

    def last_enum_usage(self):
        """-> alignments created, trie nodes used, per pair, by the last enumerate_all (also for overflowed pairs)"""
        a = np.zeros(self.n, dtype=np.int32)
        nd = np.zeros(self.n, dtype=np.int32)
        _check(lib().aln_batch_last_enum_usage(self.h, _i(a), _i(nd)), self.ctx.h)
        return a, nd

    def last_enum_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().aln_batch_last_enum_ms(self.h, C.byref(a), C.byref(b)), self.ctx.h)
        return a.value, b.value

    def optimal_enqueue(self):
        """Launch find_max + traceback + the copy of the results into a pinned slot; returns at once (two slots)."""
        _check(lib().aln_batch_optimal_enqueue(self.h), self.ctx.h)

    def optimal_collect(self):
        """Wait for the oldest enqueued slot -> scores[n], list lengths[n], status[n]."""
        scores = np.empty(self.n, dtype=np.float32)
        cnt = np.zeros(self.n, dtype=np.int32)
        status = np.zeros(self.n, dtype=np.int32)
        _check(lib().aln_batch_optimal_collect(self.h, _f(scores), _i(cnt), _i(status)), self.ctx.h)
        return scores, cnt, status

    def _string_buffers(self):
        stride = max(sum(self.dims(p)) for p in range(self.n)) + 2 if self.n else 4
        if not hasattr(self, "_tl") or len(self._tl) < self.n * stride:
            self._tl = C.create_string_buffer(max(self.n * stride, 1))
            self._ql = C.create_string_buffer(max(self.n * stride, 1))
        return stride

    def optimal_strings_enqueue(self):
        """Launch find_max + traceback + the string kernel and the copy of the lines into a pinned slot; returns at once (two slots)."""
        _check(lib().aln_batch_optimal_strings_enqueue(self.h, self._string_buffers()), self.ctx.h)

    def optimal_strings_collect(self, decode=True):
        """Wait for the oldest enqueued slot -> like optimal_strings."""
        return self._strings(lib().aln_batch_optimal_strings_collect, decode)

    def optimal_strings(self, decode=True):
        """aln_batch_optimal_strings -> scores[n], identity[n], status[n], template lines, query lines (lists of str, or the
        raw buffers + lengths when decode is False)."""
        return self._strings(lib().aln_batch_optimal_strings, decode)

    def _strings(self, fn, decode):
        scores = np.empty(self.n, dtype=np.float32)
        ident = np.zeros(self.n, dtype=np.float32)
        status = np.zeros(self.n, dtype=np.int32)
        lengths = np.zeros(self.n, dtype=np.int32)
        stride = self._string_buffers()
        rc = fn(self.h, _f(scores), _f(ident), _i(status), self._tl, self._ql, stride, _i(lengths))
        if rc != 0 and rc != E_STARTPAIR:
            _check(rc, self.ctx.h)
        if not decode:
            return scores, ident, status, self._tl, self._ql, lengths, stride
        tl = [self._tl.raw[p * stride:p * stride + lengths[p]].decode() for p in range(self.n)]
        ql = [self._ql.raw[p * stride:p * stride + lengths[p]].decode() for p in range(self.n)]
        return scores, ident, status, tl, ql

    def optimal(self, want_pairs=True, subali=False):
        """-> scores[n], list of pair arrays (list order), status[n]"""
        scores = np.empty(self.n, dtype=np.float32)
        cnt = np.zeros(self.n, dtype=np.int32)
        status = np.zeros(self.n, dtype=np.int32)
        if not hasattr(self, "_stride"):
            self._stride = max(min(max(self.dims(p)[0] for p in range(self.n)), max(self.dims(p)[1] for p in range(self.n))) + 3, 4) if self.n else 4
        stride = self._stride
        pairs = np.zeros((self.n, stride, 2), dtype=np.int32) if want_pairs else None
        fn = lib().aln_batch_optimal_subali if subali else lib().aln_batch_optimal
        _check(fn(self.h, _f(scores), _i(cnt), _i(pairs) if want_pairs else None, stride, _i(status)), self.ctx.h)
        lists = [pairs[p, :cnt[p]].copy() for p in range(self.n)] if want_pairs else None
        return scores, lists, status
