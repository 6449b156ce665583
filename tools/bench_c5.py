#!/usr/bin/env python3
"""BASELINE config 5: n x n synthetic proteins, length U[400,600] (seed 5000+s), local 11/1 BLOSUM62, scores only.

One process per GPU (`python -m torch.distributed.run --nproc-per-node N tools/bench_c5.py n`): rank r owns the query rows the C ABI's
length-sorted deal (aln_deal_units) gives it, every rank holds all templates (2 MB), and the only collective is the C ABI's
aln_gather_scores (one RCCL all-gather of (index, score) records; 16 MiB per rank at n = 4096 on 8 ranks).  Without a launcher it runs the first `rows` query rows on one GPU.
Prints GCUPS = sum |q||t| / wall seconds of aln_score_all_vs_all (+ the gather), upload of the residues and download of the
score block included.   usage: bench_c5.py [n] [rows]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
import aln_amd  # noqa: E402
from aln_amd.shard import Comm, GlooComm, deal_units, local_units  # noqa: E402
from aln_amd.synth import MT19937, residues  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else n
    rehearse = os.environ.get("ALN_BENCH_REHEARSE_ON_ONE_GPU") == "1"      # every rank on device 0, gloo: control flow only
    dist = dev = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dev = torch.device("cpu")
        else:
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    lines = open(os.path.join(ROOT, "tests", "golden", "BLOSUM62")).read().split("\n")
    k = 0
    while lines[k].startswith("#"):
        k += 1
    alphabet = "".join(lines[k].split())
    table = np.array([[float(x) for x in l.split()[1:]] for l in lines[k + 1:k + 1 + len(alphabet)]], dtype=np.float32)
    seqs = []
    for s in range(n):
        g = MT19937(5000 + s)
        ln = 400 + int(g.draw(1)[0] % 201)
        seqs.append(residues(g, ln))
    pool = aln_amd.SeqPool(seqs)
    ctx = aln_amd.Context(local_rank)
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    if world > 1:
        # query rows dealt by length (aln_deal_units), every rank holds all templates; ONE gather of (index, score) records
        owner, slot = deal_units(lens[:rows], world)
        mine = local_units(owner, slot, rank)
        comm = GlooComm(world, rank) if rehearse else Comm(ctx, world, rank)
        n_max = int(np.bincount(owner, minlength=world).max()) * n
    else:
        mine = np.arange(rows, dtype=np.int32)
    qpool = pool if world == 1 else aln_amd.SeqPool([seqs[i] for i in mine])
    aln_amd.score_all_vs_all(ctx, qpool, pool, alphabet, table, 11, 1, 0, min(64, len(mine)))   # warm-up
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    sc = aln_amd.score_all_vs_all(ctx, qpool, pool, alphabet, table, 11, 1, 0, len(mine))
    full = sc
    if world > 1:
        gidx = (mine.astype(np.int64)[:, None] * n + np.arange(n, dtype=np.int64)[None, :]).astype(np.int32).reshape(-1)
        full = comm.gather(sc.reshape(-1), gidx, n_max, rows * n).reshape(rows, n)    # the one collective of the path
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        cells = float(lens[:rows].sum()) * float(lens.sum())
        print("config5 %d x %d on %d rank(s)%s: %.3f s, %.1f GCUPS, checksum %.0f, self-scores ok=%s" % (
            rows, n, world, " [one-GPU rehearsal]" if rehearse else "", dt, cells / dt / 1e9, float(full.sum()),
            bool((np.diag(full[:, :rows]) >= full[:, :rows].max(axis=1) - 1e-6).all())))
    if world > 1:
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
