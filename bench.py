#!/usr/bin/env python3
"""bench.py — BASELINE.json config 2 on N MI355X: batched local affine-gap DP (BLOSUM62, 11/1) over 1024
synthetic 2000 x 2000 pairs per GPU, DP build + find_max + pointer traceback, through the C ABI.

A step = one pass of the hot path (aln_batch_dp + aln_batch_optimal) over the resident batch of 1024 pairs, issued as
--split launches that rotate over --streams HIP streams (see main()); sequences are uploaded to HBM before the timed region.  Pairs are independent, so ranks own disjoint batches (weak scaling)
and the only collective is one all_gather of the fp32 scores per step (RCCL over xGMI).

Prints ONE JSON line: metric GCUPS = sum |q|*|t| of all ranks / wall seconds (max over ranks), plus
  roofline     — dominant kernel (row-sweep DP) algorithmic bytes (8 B/cell) / its mean HIP-event duration
  cpu_baseline — the reference (oracle/_ref, real christang/alignment-algos DPMatrix) or the oracle port timed
                 on this box's host cores, rank 0 at N=1, one pair of the same workload per core (bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def make_workload(rank, n_pairs, length):
    """pair p uses seed 1000+p; every second pair is a mutated homolog (long tracebacks)."""
    from aln_amd.synth import homolog_pair, random_pair
    qs, ts = [], []
    for p in range(n_pairs):
        seed = 1000 + rank * n_pairs + p
        q, t = homolog_pair(seed, length) if p % 2 else random_pair(seed, length)
        qs.append(q)
        ts.append(t)
    return qs, ts


def cpu_baseline(qs, ts, mode, gi, ge):
    """the reference's DPMatrix constructor on this box's host cores — one pair per core on all cores the
    process may use (bounded sample: the first C pairs of rank 0's workload, ~20-30 s), plus the one-core figure
    (the fastest single pair of that run).  The real reference binary (oracle/_ref) if it travelled, else the oracle."""
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    blosum = os.path.join(ROOT, "tests", "golden", "BLOSUM62")
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, len(qs), 16))      # the GPU box gives one GPU a 16-core share
    cells = sum(len(qs[p]) * len(ts[p]) for p in range(cores))
    sample = "%d pairs %dx%d of the bench workload (rank 0, pairs 0..%d), one pair per core, DPMatrix build only" % (
        cores, len(qs[0]), len(ts[0]), cores - 1)
    t0 = time.time()
    if os.path.exists(harness):
        procs = [subprocess.Popen([harness, "aa", blosum, str(mode), str(gi), str(ge), "fwd", qs[p], ts[p], "ctime", "corner"],
                                  stdout=subprocess.PIPE, text=True) for p in range(cores)]
        per = []
        for pr in procs:
            out = pr.communicate()[0]
            if pr.returncode != 0:
                raise RuntimeError("reference harness failed")
            per.append([float(l.split()[1]) for l in out.split("\n") if l.startswith("CTIME")][0])
        kind = "reference"
    else:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc
        from concurrent.futures import ThreadPoolExecutor
        alpha, table = orc.load_blosum(blosum)

        def one(p):   # the oracle is a C library called through ctypes (releases the GIL)
            S = orc.sim_submatrix(qs[p], ts[p], alpha, table)
            t1 = time.time()
            orc.dp_build(S, orc.Gap(mode, gi, ge))
            return time.time() - t1
        with ThreadPoolExecutor(cores) as ex:
            per = list(ex.map(one, range(cores)))
        kind = "port"
    wall = time.time() - t0
    return {"value": cells / wall / 1e9, "unit": "GCUPS", "cores": cores, "kind": kind, "sample": sample, "seconds": round(wall, 3),
            "single_core_value": len(qs[0]) * len(ts[0]) / min(per) / 1e9, "single_core_seconds": round(min(per), 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=1024, help="pairs per GPU (config 2: 1024)")
    ap.add_argument("--length", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", "--batches", dest="batches", type=int, default=4,
                    help="HIP streams (contexts) the launches rotate over (1 = everything on one stream)")
    ap.add_argument("--split", type=int, default=2,
                    help="launches a step's batch is processed in (sub-batches of pairs/split pairs; 1 = one launch per step)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 control flow on a one-GPU box: every rank on device 0, gloo instead of RCCL (not a measurement)
    rehearse = os.environ.get("ALN_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import aln_amd
    blosum = os.path.join(ROOT, "tests", "golden", "BLOSUM62")
    lines = open(blosum).read().split("\n")
    k = 0
    while lines[k].startswith("#"):
        k += 1
    alphabet = "".join(lines[k].split())
    table = np.array([[float(x) for x in l.split()[1:]] for l in lines[k + 1:k + 1 + len(alphabet)]], dtype=np.float32)

    mode, gi, ge = aln_amd.LOCAL, 11, 1
    qs, ts = make_workload(rank, args.pairs, args.length)
    # How a step's batch reaches the GPU.  The batch of `pairs` pairs is processed as `split` sub-batches (pairs/split pairs
    # each, one launch sequence each) and consecutive launches rotate over `streams` contexts, each with its own HIP stream:
    # launch j runs sub-batch j % split on stream j % streams.  Launches of different streams overlap, so there are always
    # undispatched pairs to take the SIMD slots that finished pairs free (a lone 1024-pair launch fills the GPU exactly
    # once and leaves early-finishing SIMDs idle: DESIGN.md 4.1), and the O(Q+T) corner kernel and the traceback run
    # beside the next DP kernel.  Measured on one box, ms per 1024 pairs: 1 stream x 1024 pairs 3.3-3.6; 2 streams x 1024
    # pairs 2.8-3.8 depending on the phase the two streams fall into; 3 streams x 512 pairs 2.70-2.97; 4 streams x 512
    # pairs 2.69-2.76 (the default); 5 x 512: 3.57; 3 x 256: 3.76.  Every step still builds, scans and traces all `pairs`
    # pairs; every (stream, sub-batch) combination that occurs has its own resident planes.
    nb = max(1, args.batches)
    split = args.split if (args.split >= 1 and args.pairs % max(1, args.split) == 0) else 1
    ph = args.pairs // split
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nb - 1)]
    ctxs = [aln_amd.Context(local_rank, st.cuda_stream) for st in streams]
    units = {}                                              # (stream, sub-batch) -> resident batch object
    j = 0
    while (j % nb, j % split) not in units:
        sidx, h = j % nb, j % split
        units[(sidx, h)] = aln_amd.Batch(ctxs[sidx], qs[h * ph:(h + 1) * ph], ts[h * ph:(h + 1) * ph])   # sequences -> HBM, planes allocated
        j += 1
    batches = list(units.values())
    # codes + substitution table -> HBM and the first build (the DPMatrix constructors); timed steps then
    # re-run the build on the resident inputs exactly like DPMatrix::reevaluate (dpmatrix.h:213-218)
    for bt in batches:
        bt.dp_submatrix(alphabet, table, mode, gi, ge, aln_amd.FWD, aln_amd.DP_FAST)
    batch = batches[0]

    from aln_amd.shard import gather_scores

    # A step = DPMatrix::reevaluate (DP + corner kernels) + Optimal (find_max + traceback kernels, per-pair results copied to the
    # host) [+ the gather of the scores over the ranks].  Steps are software-pipelined: step k's kernels and result copy are
    # enqueued, then step k-1's results are collected (and gathered), so the host's launch / copy latency hides behind the
    # kernels of the next step.  Every enqueued step is collected inside the timed region.
    queue = []                                              # batches with an enqueued, not yet collected step (oldest first)
    count = [0]
    side = torch.cuda.Stream(dev) if (world > 1 and not rehearse) else None    # the score gather does not queue behind the kernels

    def collect():
        sc, cnt, status = queue.pop(0).optimal_collect()
        if world > 1:                                       # the one collective of the path: all ranks' scores (RCCL)
            gather_scores(sc, len(sc) * world, world, rank, device=None if rehearse else dev, stream=side)
        return sc, status

    def launch():                                           # one sub-batch: build + find_max + traceback, results to the host
        bt = units[(count[0] % nb, count[0] % split)]
        count[0] += 1
        while bt in queue:                                  # its previous results must be read out before its planes are rebuilt
            collect()
        bt.reevaluate()
        bt.optimal_enqueue()
        queue.append(bt)
        return collect() if len(queue) > nb else (None, None)

    def step():                                             # all sub-batches of the batch
        out = (None, None)
        for _ in range(split):
            out = launch()
        return out

    def drain():
        out = (None, None)
        while queue:
            out = collect()
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sc, status = drain()
    fence()
    elapsed = time.perf_counter() - t0
    # HIP events around the DP kernel of each timed launch, on the stream it was launched on
    n_launch = args.steps * split
    first = args.warmup * split                             # index of the first timed launch
    per = {}
    for j in range(first, first + n_launch):
        k = (j % nb, j % split)
        per[k] = per.get(k, 0) + 1
    kernel_ms = np.concatenate([units[k].dp_ms_history(min(n, 64)) for k, n in per.items()])
    assert (status == 0).all()
    ref_sc = {}                                             # same inputs -> identical results, whichever stream ran them
    for (sidx, h), bt in units.items():
        bt.reevaluate()
        s_b, _, st_b = bt.optimal()
        assert (st_b == 0).all()
        if h not in ref_sc:
            ref_sc[h] = np.array(s_b, copy=True)
        else:
            assert np.array_equal(ref_sc[h], s_b), "resident copies of a sub-batch disagree"
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # this box's plain streaming-store rate (torch fill of 8 GiB, HIP events on the same stream): what "HBM-write bound"
    # can mean on this device today; boxes of the pool differ by ~20 % on it
    fill_gbs = None
    try:
        buf = torch.empty(2 << 30, dtype=torch.float32, device=dev)
        buf.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            buf.fill_(2.0)
        e1.record()
        torch.cuda.synchronize(dev)
        fill_gbs = 3 * buf.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del buf
    except Exception:
        pass

    first_of = {}
    for (sidx, h), bt in units.items():
        first_of.setdefault(h, bt)
    cells_per_step = sum(first_of[h].cells() for h in range(split)) * world
    value = cells_per_step * args.steps / elapsed / 1e9
    dp_ms = float(np.mean(kernel_ms))
    algo_bytes = batch.algorithmic_bytes()                  # of one launch (one sub-batch)
    if nb == 1:
        achieved = algo_bytes / (dp_ms * 1e-3) / 1e9
        how = "algorithmic bytes per launch / average launch duration (HIP events on the launch stream)"
    else:
        # Launches of different resident batches overlap on the GPU, so a launch's own duration (kernel_ms, what rocprofv3
        # also reports) is not the time the device spends per launch.  The aggregate rate of the kernel is bounded from
        # below by all timed launches' algorithmic bytes over the wall time of the timed region, which also contains the
        # corner and traceback kernels; that lower bound is what is reported.
        achieved = algo_bytes * args.steps * split / elapsed / 1e9
        how = ("launches of %d streams overlap: algorithmic bytes of all timed launches / wall time of the timed region (lower "
               "bound; kernel_ms is one launch's own duration while it shares the GPU)" % nb)
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile) and args.length == 2000:       # only if the PMC pass was taken on exactly this launch shape
        try:
            tj = json.load(open(tfile))
            if tj.get("launch_pairs", 1024) == ph:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "GCUPS (DP cell updates/s) at 1/2/4/8 MI355X; bit-exact score vs ref",
        "value": round(value, 3), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "config 2: %d synthetic %dx%d pairs per GPU, local SW, affine gap 11/1, BLOSUM62 submatrix evaluator, "
                               "DP build + find_max + traceback" % (args.pairs, args.length, args.length),
                   "pairs_per_gpu": args.pairs, "parallelism": "pair-batch sharded, %d rank(s), all_gather of scores" % world,
                   "kernel": batch.kernel_name(), "launch_pairs": ph, "launches_per_step": split, "streams": nb},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel_ms": round(dp_ms, 3), "concurrent_launches": nb, "achieved_is": how,
                     "algorithmic_bytes": algo_bytes,
                     "measured_fill_GBs_this_box": round(fill_gbs, 1) if fill_gbs else None},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(qs, ts, mode, gi, ge)
    if rank == 0:
        print(json.dumps(out), flush=True)
    for bt in batches:
        bt.close()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
