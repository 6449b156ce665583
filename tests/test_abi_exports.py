"""CPU-only: libalnhip.so loads and exports every entry point include/aln_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import aln_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "aln_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(aln_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(aln_amd.LIB_PATH):
        aln_amd.build_library()
    L = ctypes.CDLL(aln_amd.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "missing export " + n
    assert sorted(aln_amd.EXPORTS) == names
    assert L.aln_has_gfx950() == 1
    L.aln_error_string.restype = ctypes.c_char_p
    assert L.aln_error_string(-1) == b"Illegal bounds building DPM"          # dpmatrix.h:361
    assert L.aln_error_string(-3) == b"Illegal alignment start pair"         # optimal.h:74


def test_host_helpers_identity_and_strings():
    """aln_identity / aln_gapped_strings need no GPU: check them on the SURVEY App. C known answers."""
    import numpy as np
    L = aln_amd.lib()
    q, t = b"^PAWHEAE$", b"^HEAGAWGHEE$"
    pairs = np.array([0, 0, 2, 5, 3, 6, 4, 7, 5, 8, 6, 9, 7, 10, 8, 11], dtype=np.int32)
    ali = (aln_amd.AlnAlignment * 1)()
    ali[0].n_pairs = 8
    ali[0].pair_off = 0
    ip = pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    n = L.aln_gapped_length(12, ali, 1, ip)
    assert n == 13
    tl = ctypes.create_string_buffer(n + 1)
    ql = ctypes.create_string_buffer(n + 1)
    assert L.aln_gapped_strings(q, 9, t, 12, ali, 1, ip, tl, ql, n + 1) == 0
    assert (tl.value, ql.value) == (b"^-HEAGAWGHEE$", b"^p----AWHEAE$")
    idn = L.aln_identity(q, 9, t, 12, ip, 8)
    assert abs(idn - 42.857143) < 1e-4


def test_host_strings_match_golden_sets():
    """The product's own SequenceGaps/calcIdentity restatement against every alignment set the real reference printed."""
    import numpy as np
    import goldens
    import gpu_util
    import orc
    n = 0
    for case in goldens.cases():
        for key, s in case["sets"].items():
            if "tstr" not in s or not all("pairs" in a for a in s["alis"]):
                continue
            lists = [np.array(a["pairs"], np.int32).reshape(-1, 2) for a in s["alis"]]
            tl, qls, idn = gpu_util.strings_for(case["q"], case["t"], lists)
            assert tl == s["tstr"], case["name"]
            for k, a in enumerate(s["alis"]):
                assert qls[k] == a["qstr"], (case["name"], key, k)
                assert goldens.f32bits(idn[k]) == a["identity"], (case["name"], key, k)
                sc = np.array([a["score"]], np.uint32).view(np.float32)[0]
                assert orc.annot(sc, idn[k]) == a["annot"]
                n += 1
    assert n > 500
