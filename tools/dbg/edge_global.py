import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import aln_amd, gpu_util
import orc
from aln_amd.synth import homolog_pair
alpha, table = orc.load_blosum(os.path.join(ROOT, 'tests', 'golden', 'BLOSUM62'))
q3, t3 = homolog_pair(77002, 4095)
for mode in (1, 3):
    for qq, tt in ((q3[:4094], t3[:4094]), (q3, t3[:4000]), (q3[:3000], t3[:3000])):
        out = {}
        for name, hint, algo in (("fast", None, aln_amd.DP_FAST), ("int", ("tag_kernel", 0), aln_amd.DP_FAST), ("exact", None, aln_amd.DP_EXACT)):
            ctx = aln_amd.Context(0)
            if hint: ctx.set_hint(*hint)
            b = aln_amd.Batch(ctx, [qq], [tt])
            b.dp_submatrix(alpha, table, mode, 11, 1, aln_amd.FWD, algo)
            out[name] = (b.kernel_name(), b.get_cells(0))
            b.close()
        for name in ("fast", "int"):
            D, PQ, PT = out[name][1]; E, EQ, ET = out["exact"][1]
            bad = np.argwhere(D.view(np.uint32) != E.view(np.uint32))
            badp = np.argwhere((PQ != EQ) | (PT != ET))
            print(mode, len(qq), len(tt), name, out[name][0], "score diffs", len(bad), bad[:3].tolist(), [(float(D[i, j]), float(E[i, j])) for i, j in bad[:3]], "ptr diffs", len(badp), badp[:3].tolist(), flush=True)
