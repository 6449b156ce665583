#!/usr/bin/env python3
"""development aid: config-3 launches overlapping on two streams vs back to back (refill effect of freed SIMD slots)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import torch
import aln_amd
from aln_amd.synth import random_profile

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = 2000
dev = torch.device("cuda", 0)
qps = [random_profile(3000 + p, L) for p in range(64)]
tps = [random_profile(4000 + p, L) for p in range(64)]
qpool = {k: np.concatenate([qps[p % 64][k] for p in range(n)]) for k in ("aa", "sse", "conf")}
tpool = {k: np.concatenate([tps[p % 64][k] for p in range(n)]) for k in ("aa", "sse", "conf")}
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
ctxs = [aln_amd.Context(0, s.cuda_stream) for s in streams]
bs = [aln_amd.Batch(c, ["A" * L] * n, ["A" * L] * n) for c in ctxs]
for b in bs:
    b.dp_hmap2(qpool, tpool, aln_amd.GLOBAL, 4.73, 0.34, 0.5, 1.0, 0.12)
torch.cuda.synchronize()
def run(order):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in order:
        bs[i].reevaluate()
        if len(set(order)) == 1: pass
    torch.cuda.synchronize(); return time.perf_counter() - t0
print("one launch        %.3f s" % run([0]))
print("two, same stream  %.3f s" % run([0, 0]))
print("two, two streams  %.3f s" % run([0, 1]))
print("four, two streams %.3f s" % run([0, 1, 0, 1]))
print(bs[0].kernel_name(), bs[0].last_dp_ms())
