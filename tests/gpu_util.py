"""Helpers for the parity tests: everything goes through the C ABI (aln_amd -> libalnhip.so)."""
import ctypes as C

import numpy as np

import aln_amd

_CTX = None


def ctx():
    global _CTX
    if _CTX is None:
        _CTX = aln_amd.Context(0)
    return _CTX


def strings_for(q, t, lists):
    """Gapped template line + query lines + identities for a set of alignments of ONE pair, via the C ABI host helpers."""
    L = aln_amd.lib()
    qs, ts = ("^" + q + "$").encode(), ("^" + t + "$").encode()
    n = len(lists)
    alis = (aln_amd.AlnAlignment * max(n, 1))()
    flat = []
    off = 0
    for k, p in enumerate(lists):
        alis[k].n_pairs = len(p)
        alis[k].pair_off = off
        off += len(p)
        flat.append(np.asarray(p, np.int32).reshape(-1))
    blob = np.ascontiguousarray(np.concatenate(flat)) if flat else np.zeros(2, np.int32)
    ip = blob.ctypes.data_as(C.POINTER(C.c_int32))
    ln = L.aln_gapped_length(len(ts), alis, n, ip)
    stride = ln + 1
    tl = C.create_string_buffer(stride)
    ql = C.create_string_buffer(stride * max(n, 1))
    rc = L.aln_gapped_strings(qs, len(qs), ts, len(ts), alis, n, ip, tl, ql, stride)
    assert rc == 0, rc
    qls = [ql.raw[k * stride:(k + 1) * stride].split(b"\0")[0].decode() for k in range(n)]
    idn = []
    for p in lists:
        a = np.ascontiguousarray(np.asarray(p, np.int32).reshape(-1))
        idn.append(np.float32(L.aln_identity(qs, len(qs), ts, len(ts), a.ctypes.data_as(C.POINTER(C.c_int32)), len(a) // 2)))
    return tl.value.decode(), qls, idn


def identity_for(q, t, pairs):
    """calcIdentity of one pair list via the C ABI host helper."""
    L = aln_amd.lib()
    qs, ts = ("^" + q + "$").encode(), ("^" + t + "$").encode()
    a = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1))
    if len(a) == 0:
        a = np.zeros(2, np.int32)
        return np.float32(L.aln_identity(qs, len(qs), ts, len(ts), a.ctypes.data_as(C.POINTER(C.c_int32)), 0))
    return np.float32(L.aln_identity(qs, len(qs), ts, len(ts), a.ctypes.data_as(C.POINTER(C.c_int32)), len(a) // 2))
