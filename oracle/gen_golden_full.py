#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — full-size goldens from the REAL reference (oracle/_ref, built by oracle/Makefile from
/root/reference; this container only).  Writes tests/golden/full_cases.json:

  c2    pairs of bench.py's config-2 workload (seed 1000+p, 2000 x 2000, local, 11/1, BLOSUM62; odd p = homolog):
        sha256 of the reference's score / prev_query / prev_template planes (getCell over the whole matrix), per-row
        CRC32s to localise a mismatch, Optimal (score, pair list), and the ConstrainedNearOptimal set at NUM_SUBOPT=256 with
        make_subopt_regions(T, 10) flags (config 4) for every DELTA_RATIO in C4_DELTAS the reference finishes.
  c3    one 2000 x 2000 GLOBAL profile pair (bench_c3's generator, seeds 3000 / 4000) through oracle/_ref/ref_profile
        (real hmath.h / SimilarityMatrix / DPMatrix / Optimal): sha256 of S, H, PQ, PT + Optimal.
  long  four pairs between 2049 and 4094 residues (the kernels' paths beyond 2048 columns / rows): plane sha256 + Optimal.
  c3sl  a second 2000 x 2000 profile pair, semi_local.
  c2ucw UnconstrainedNearOptimal sets of three of the c2 homologs (added to their c2 entries; needs the c2 part in the file).
  longm four more such pairs in the non-local align_t (global, global-local, semi-local, mode 0).
  c5    a 32 x 32 block of config 5's sequence set (seed 5000+s, 400..600 aa): the score Optimal reports, local 11/1;
        and 12 x 12 blocks for the other four align_t.

The O(n^3) reference needs ~25-40 s per 2000 x 2000 pair and core; everything runs once, in parallel, here.
usage: gen_golden_full.py [c2] [c3] [c5] [long] [longm] [c2ucw] [c3sl]   (default: all; parts not regenerated are kept from the existing file)
"""
import hashlib
import json
import os
import subprocess
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import refrun  # noqa: E402
from aln_amd.synth import MT19937, homolog_pair, make_subopt_regions, random_pair, random_profile, residues  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(GOLD, "full_cases.json")
TMP = os.path.join(HERE, "_ref", "tmp")
C2_PAIRS = [0, 1, 2, 3, 512, 513, 1022, 1023]
C2_LEN = 2000
C4_DELTAS = [0.05, 0.01, 0.005]
C4_MEM_KB = 6000000           # ulimit -v of one reference enumeration: it holds one list copy per alignment (cw.h:158) up to
                              # user_limit = 10^6 lists (cw.h:76); pair 1 at DELTA_RATIO 0.05 was also tried with 14 GB: std::bad_alloc


def bits(x):
    return int(np.float32(x).view(np.uint32))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def row_crc(a):
    a = np.ascontiguousarray(a)
    return [zlib.crc32(a[i].tobytes()) for i in range(a.shape[0])]


def c2_pair(p):
    """bench.py make_workload(rank 0): pair p."""
    return homolog_pair(1000 + p, C2_LEN) if p % 2 else random_pair(1000 + p, C2_LEN)


def compact_set(s):
    d = {"n": s["n"], "alis": []}
    if "tstr" in s:
        d["tstr_sha"] = hashlib.sha256(s["tstr"].encode()).hexdigest()
    for a in s["alis"]:
        e = {"score": bits(a["score"]), "identity": bits(a["identity"]), "uid": a["uid"], "n_pairs": int(len(a["pairs"])),
             "pairs_sha": sha(a["pairs"].astype(np.int32)), "annot": a["annot"]}
        if "qstr" in a:
            e["qstr_sha"] = hashlib.sha256(a["qstr"].encode()).hexdigest()
        d["alis"].append(e)
    return d


def run_c2(p):
    q, t = c2_pair(p)
    os.makedirs(TMP, exist_ok=True)
    path = os.path.join(TMP, "c2_%d.bin" % p)
    r = refrun.run_aa(q, t, 3, 11, 1, "fwd", ops=["bin", path, "opt"], timeout=3600)
    raw = np.fromfile(path, dtype=np.int32)
    os.remove(path)
    Q, T = int(raw[0]), int(raw[1])
    planes = raw[2:].reshape(3, Q, T)
    H, PQ, PT = planes[0].view(np.uint32), planes[1], planes[2]
    opt = r["sets"]["OPT"]["alis"][0]
    c = {"pair": p, "seed": 1000 + p, "homolog": bool(p % 2), "len": C2_LEN, "mode": 3, "gi": 11, "ge": 1,
         "q_sha": hashlib.sha256(q.encode()).hexdigest(), "t_sha": hashlib.sha256(t.encode()).hexdigest(),
         "sha": {"H": sha(H), "PQ": sha(PQ), "PT": sha(PT)},
         "row_crc": {"H": row_crc(H), "P": row_crc(np.stack([PQ, PT], axis=2))},
         "corner": int(H[-1, -1]), "hmax": int(H.view(np.float32).max()),
         "opt": {"score": bits(opt["score"]), "identity": bits(opt["identity"]), "pairs": opt["pairs"].reshape(-1).tolist()},
         "cw": {}}
    print("c2 pair %d: planes done, opt score %g (%d pairs)" % (p, float(opt["score"]), len(opt["pairs"])), flush=True)
    return c


def run_c4(p, delta):
    """ConstrainedNearOptimal at K=256, 10 regions; None when the reference exceeds C4_MEM_KB (it copies a whole list per
    accepted branch, cw.h:158) — recorded as 'reference does not finish'."""
    q, t = c2_pair(p)
    fl = "".join(str(int(x)) for x in make_subopt_regions(C2_LEN + 2, 10))
    cmd = "ulimit -v %d; exec %s aa %s 3 11.0 1.0 fwd %s %s cw 256 %r %s" % (C4_MEM_KB, refrun.HARNESS, refrun.BLOSUM62, q, t, delta, fl)
    pr = subprocess.run(["/bin/bash", "-c", cmd], capture_output=True, text=True)
    if pr.returncode != 0:
        print("c4 pair %d delta %g: reference did not finish (rc %d)" % (p, delta, pr.returncode), flush=True)
        return p, delta, None
    s = refrun.parse(pr.stdout)["sets"]["CW"]
    print("c4 pair %d delta %g: %d alignments" % (p, delta, s["n"]), flush=True)
    return p, delta, compact_set(s)


C4_UCW = [(1, 0.002), (3, 0.002), (513, 0.002)]    # UnconstrainedNearOptimal at full size: DELTA_RATIOs the reference finishes in ~1 min


def run_ucw(p, delta):
    """UnconstrainedNearOptimal (ucw.h:64-236) on a config-2 homolog at K=256: the sorted set, or None (out of memory)."""
    q, t = c2_pair(p)
    cmd = "ulimit -v %d; exec %s aa %s 3 11.0 1.0 fwd %s %s ucw 256 %r" % (2 * C4_MEM_KB, refrun.HARNESS, refrun.BLOSUM62, q, t, delta)
    pr = subprocess.run(["/bin/bash", "-c", cmd], capture_output=True, text=True)
    if pr.returncode != 0:
        print("ucw pair %d delta %g: reference did not finish (rc %d)" % (p, delta, pr.returncode), flush=True)
        return p, delta, None
    s = refrun.parse(pr.stdout)["sets"]["UCW"]
    print("ucw pair %d delta %g: %d alignments" % (p, delta, s["n"]), flush=True)
    return p, delta, compact_set(s)


def gen_c2(pool):
    cases = list(pool.map(run_c2, C2_PAIRS))
    jobs = [(p, d) for p in C2_PAIRS for d in C4_DELTAS]
    by = {c["pair"]: c for c in cases}
    for p, d, s in pool.map(lambda a: run_c4(*a), jobs):
        by[p]["cw"]["%g" % d] = s
    return {"note": "bench.py config-2 workload, rank 0; cw = ConstrainedNearOptimal NUM_SUBOPT=256, make_subopt_regions(T,10); "
                    "a null cw entry = the reference ran out of %d kB of address space at that DELTA_RATIO" % C4_MEM_KB,
            "flags": "".join(str(int(x)) for x in make_subopt_regions(C2_LEN + 2, 10)), "pairs": cases}


def gen_c3(mode=1, q_seed=3000, t_seed=4000):
    L = 2000
    qp, tp = random_profile(q_seed, L), random_profile(t_seed, L)
    os.makedirs(TMP, exist_ok=True)
    path = os.path.join(TMP, "c3_%d.bin" % mode)
    r = refrun.run_profile(qp, tp, mode, 0.5, 1.0, 0.12, 4.73, 0.34, 1, timeout=7200, env={"REF_PROFILE_BIN": path})
    raw = np.fromfile(path, dtype=np.int32)
    os.remove(path)
    Q, T = int(raw[0]), int(raw[1])
    pl = raw[2:].reshape(4, Q, T)
    opt = r["sets"]["OPT"]["alis"][0]
    print("c3 mode %d: opt score %g (%d pairs)" % (mode, float(opt["score"]), len(opt["pairs"])), flush=True)
    return {"note": "random_profile(%d, 2000) x random_profile(%d, 2000) (bench_c3 pair 0 for 3000/4000), align_t %d, alpha 0.5 beta 1 zero_shift 0.12, "
                    "gaps 4.73/0.34, through oracle/_ref/ref_profile" % (q_seed, t_seed, mode),
            "len": L, "q_seed": q_seed, "t_seed": t_seed, "mode": mode, "alpha": 0.5, "beta": 1.0, "zero_shift": 0.12, "gi": 4.73, "ge": 0.34,
            "sha": {"S": sha(pl[0].view(np.uint32)), "H": sha(pl[1].view(np.uint32)), "PQ": sha(pl[2]), "PT": sha(pl[3])},
            "row_crc": {"S": row_crc(pl[0]), "H": row_crc(pl[1]), "P": row_crc(np.stack([pl[2], pl[3]], axis=2))},
            "tgi_sha": sha(r["TGI"].view(np.uint32)), "tge_sha": sha(r["TGE"].view(np.uint32)),
            "opt": {"score": bits(opt["score"]), "pairs": opt["pairs"].reshape(-1).tolist()}}


def c5_seqs(n):
    out = []
    for s in range(n):
        g = MT19937(5000 + s)
        ln = 400 + int(g.draw(1)[0] % 201)
        out.append(residues(g, ln))
    return out


def run_block(args):
    mode, seqfile, r0, r1, n = args
    out = subprocess.run([refrun.HARNESS, "block", refrun.BLOSUM62, str(mode), "11", "1", seqfile, str(r0), str(r1), "0", str(n)],
                         capture_output=True, text=True, check=True).stdout
    rows = {}
    for line in out.split("\n"):
        tk = line.split()
        if tk and tk[0] == "ROW":
            rows[int(tk[1])] = [int(x, 16) for x in tk[2:]]
    print("c5 mode %d rows %d..%d done" % (mode, r0, r1), flush=True)
    return mode, rows


def gen_c5(pool):
    os.makedirs(TMP, exist_ok=True)
    seqs = c5_seqs(32)
    seqfile = os.path.join(TMP, "c5_seqs.txt")
    with open(seqfile, "w") as f:
        f.write("\n".join(seqs) + "\n")
    jobs = [(3, seqfile, r, r + 2, 32) for r in range(0, 32, 2)]
    for mode in (0, 1, 2, 4):
        jobs += [(mode, seqfile, r, r + 2, 12) for r in range(0, 12, 2)]
    res = {}
    for mode, rows in pool.map(run_block, jobs):
        res.setdefault(mode, {}).update(rows)
    os.remove(seqfile)
    blocks = {}
    for mode, rows in res.items():
        blocks[str(mode)] = [rows[r] for r in sorted(rows)]
    return {"note": "config-5 sequence set s = 0..31 (seed 5000+s, length 400 + g() % 201); scores[mode][r][c] = bits of the score "
                    "Optimal(align_t = mode) reports for DPMatrix(seq r as query, seq c as template, fwd), gaps 11/1, BLOSUM62",
            "n_local": 32, "n_other": 12, "lengths": [len(s) for s in seqs], "seq_sha": [hashlib.sha256(s.encode()).hexdigest() for s in seqs],
            "scores": blocks}


def run_long(args):
    """One pair beyond 2048 residues (the tagged kernel's 12-bit tag layout, the int kernel, the exact kernel)."""
    name, seed, qlen, tlen, homolog = args[:5]
    mode = args[5] if len(args) > 5 else 3
    if homolog:
        q, t = homolog_pair(seed, max(qlen, tlen))
        q, t = q[:qlen], t[:tlen]
    else:
        q, t = random_pair(seed, qlen, tlen)
    os.makedirs(TMP, exist_ok=True)
    path = os.path.join(TMP, "long_%s.bin" % name)
    r = refrun.run_aa(q, t, mode, 11, 1, "fwd", ops=["bin", path, "opt"], timeout=7200)
    raw = np.fromfile(path, dtype=np.int32)
    os.remove(path)
    Q, T = int(raw[0]), int(raw[1])
    planes = raw[2:].reshape(3, Q, T)
    opt = r["sets"]["OPT"]["alis"][0]
    print("long %s: %d x %d, opt score %g (%d pairs)" % (name, qlen, tlen, float(opt["score"]), len(opt["pairs"])), flush=True)
    return {"name": name, "seed": seed, "qlen": qlen, "tlen": tlen, "homolog": bool(homolog), "mode": mode, "gi": 11, "ge": 1,
            "q_sha": hashlib.sha256(q.encode()).hexdigest(), "t_sha": hashlib.sha256(t.encode()).hexdigest(),
            "sha": {"H": sha(planes[0].view(np.uint32)), "PQ": sha(planes[1]), "PT": sha(planes[2])},
            "row_crc": {"H": row_crc(planes[0]), "P": row_crc(np.stack([planes[1], planes[2]], axis=2))},
            "corner": int(planes[0].view(np.uint32)[-1, -1]),
            "opt": {"score": bits(opt["score"]), "pairs": opt["pairs"].reshape(-1).tolist()}}


LONG_CASES = [("t2049", 2100, 1500, 2049, True), ("q2049", 2101, 2049, 700, True), ("sq3000", 2102, 3000, 3000, True),
              ("max4094", 2103, 4094, 4094, True)]
# the same paths in the other align_t (fp32 score plane instead of uint16): global, global-local, semi-local
LONGM_CASES = [("g2100", 2110, 2100, 2060, True, 1), ("gl2300", 2111, 2300, 2049, True, 2), ("sl2200", 2112, 2049, 2200, True, 4),
               ("g3000r", 2113, 2500, 3000, False, 0)]


def main():
    if not refrun.available():
        raise SystemExit("oracle/_ref/ref_harness missing: run `make -C oracle` where /root/reference exists")
    parts = [a for a in sys.argv[1:] if a in ("c2", "c3", "c5", "long", "longm", "c2ucw", "c3sl")] or ["c2", "c3", "c5", "long", "longm", "c2ucw", "c3sl"]
    doc = {"generator": "oracle/gen_golden_full.py via oracle/_ref (real reference, g++ -O2, no -ffast-math)"}
    if os.path.exists(OUT):
        with open(OUT) as f:
            doc.update(json.load(f))
    with ThreadPoolExecutor(8) as pool:
        fut3 = pool.submit(gen_c3) if "c3" in parts else None
        futl = [pool.submit(run_long, c) for c in LONG_CASES] if "long" in parts else []
        futm = [pool.submit(run_long, c) for c in LONGM_CASES] if "longm" in parts else []
        if "c5" in parts:
            doc["c5"] = gen_c5(pool)
        if "c2" in parts:
            doc["c2"] = gen_c2(pool)
        if fut3:
            doc["c3"] = fut3.result()
        if futl:
            doc["long"] = {"note": "pairs beyond the 2048-residue limit of the 11-bit tag layout, local 11/1 BLOSUM62; homologs truncated to "
                                   "(qlen, tlen)", "pairs": [f.result() for f in futl]}
        if "c3sl" in parts:                              # config 3 "global ... and semi_local"
            doc["c3sl"] = pool.submit(gen_c3, 4, 3001, 4001).result()
        if "c2ucw" in parts:
            by = {c["pair"]: c for c in doc["c2"]["pairs"]}
            for p, d, st in pool.map(lambda a: run_ucw(*a), C4_UCW):
                by[p].setdefault("ucw", {})["%g" % d] = st
        if futm:
            doc["longm"] = {"note": "pairs beyond 2048 residues in the non-local align_t (mode field), 11/1 BLOSUM62", "pairs": [f.result() for f in futm]}
    with open(OUT, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
