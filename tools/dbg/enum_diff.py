"""One-off differential run: the several-wave search against the one-wave search on many random small cases."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import aln_amd, orc
from aln_amd.synth import homolog_pair, random_pair, make_subopt_regions
alpha, table = orc.load_blosum(os.path.join(ROOT, "tests", "golden", "BLOSUM62"))
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = aln_amd.Context(0)
bad = 0; ncase = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    n = 6
    pairs = []
    for p in range(n):
        L = int(rng.randint(3, 260))
        if rng.rand() < 0.7:
            q, t = homolog_pair(int(rng.randint(1, 10**6)), L, sub_rate=float(rng.choice([0.1, 0.25, 0.4])), indel=int(rng.randint(1, 6)))
        else:
            q, t = random_pair(int(rng.randint(1, 10**6)), L, int(rng.randint(3, 260)))
        pairs.append((q, t))
    mode = int(rng.choice([0, 1, 2, 3, 4]))
    gi, ge = [(11, 1), (4.73, 0.34), (3, 0), (7, 2)][int(rng.randint(0, 4))]
    kind = "cw" if (rng.rand() < 0.75 or max(len(q) for q, t in pairs) > 70) else "ucw"
    delta = float(rng.choice([0.01, 0.05, 0.1, 0.2]))
    if kind == "ucw":
        delta = min(delta, 0.05)
    K = int(rng.choice([3, 20, 256]))
    b = aln_amd.Batch(ctx, [p[0] for p in pairs], [p[1] for p in pairs])
    b.dp_submatrix(alpha, table, mode, gi, ge)
    res = {}
    out_flags = [make_subopt_regions(len(t) + 2, int(rng.randint(1, 9))) for q, t in pairs]
    for w in (1, 16, 5):
        ctx.set_hint("enum_waves", w)
        out = []
        for p, (q, t) in enumerate(pairs):
            try:
                got = b.enumerate(p, kind, K, delta, out_flags[p] if kind == "cw" else None, max_alignments=4096)
            except aln_amd.AlnError as e:
                got = "ERR %d" % e.code
            out.append(got)
        res[w] = out
        # batched entry too
        if kind == "cw":
            fl = np.zeros((n, max(len(t) for q, t in pairs) + 2), dtype=np.uint8)
            for p in range(n):
                fl[p, :len(out_flags[p])] = out_flags[p]
            try:
                r = b.enumerate_all(kind, K, delta, fl, K=K + 2 if K < 256 else 258, node_cap=1 << 16, ali_cap=1 << 12, raise_on_overflow=False)
                res[(w, "all")] = r
            except aln_amd.AlnError as e:
                res[(w, "all")] = "ERR %d" % e.code
    b.close()
    print("it %d: %s mode %d gaps %s delta %g K %d, set sizes %s" % (it, kind, mode, (gi, ge), delta, K, [len(x) if not isinstance(x, str) else x for x in res[1]]), flush=True)
    for p in range(n):
        ncase += 1
        a = res[1][p]
        for w in (16, 5):
            c = res[w][p]
            same = (isinstance(a, str) and a == c) or (not isinstance(a, str) and not isinstance(c, str) and len(a) == len(c) and all(
                np.float32(x["score"]).view(np.uint32) == np.float32(y["score"]).view(np.uint32) and np.array_equal(x["pairs"], y["pairs"]) and x["uid"] == y["uid"] for x, y in zip(a, c)))
            if not same:
                bad += 1
                print("MISMATCH it %d pair %d waves %d kind %s mode %d gaps %s delta %g K %d: %s vs %s" % (it, p, w, kind, mode, (gi, ge), delta, K, len(a) if not isinstance(a, str) else a, len(c) if not isinstance(c, str) else c), flush=True)
    if kind == "cw" and not isinstance(res[(1, "all")], str):
        for w in (16, 5):
            r1, r2 = res[(1, "all")], res[(w, "all")]
            # (a pair one kernel could not finish in the given pools is not a difference of results: the run-length trie needs fewer nodes)
            okp = (r1[4] == 0) & (r2[4] == 0) if not isinstance(r2, str) else None
            if isinstance(r2, str) or not np.array_equal(r1[0][okp], r2[0][okp]):
                bad += 1; print("MISMATCH batched n_out/status it %d waves %d: one-wave n_out %s status %s | %s" % (it, w, r1[0].tolist(), r1[4].tolist(), r2 if isinstance(r2, str) else (r2[0].tolist(), r2[4].tolist())), flush=True); continue
            for p in np.nonzero(okp)[0]:
                k = int(r1[0][p])
                if not (np.array_equal(r1[1][p, :k].view(np.uint32), r2[1][p, :k].view(np.uint32)) and np.array_equal(r1[2][p, :k], r2[2][p, :k])):
                    bad += 1; print("MISMATCH batched scores it %d pair %d waves %d" % (it, p, w), flush=True)
                elif r1[3] is not None and any(not np.array_equal(r1[3][p, a, :r1[2][p, a]], r2[3][p, a, :r2[2][p, a]]) for a in range(k)):
                    bad += 1; print("MISMATCH batched pairs it %d pair %d waves %d" % (it, p, w), flush=True)
print("cases %d, mismatches %d" % (ncase, bad))
