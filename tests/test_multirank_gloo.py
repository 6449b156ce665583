"""CPU, world_size 2, gloo: the N>1 path of bench.py — the length-sorted deal of the pair list over ranks (the C ABI's
aln_deal_units, pure host code) and the single gather of the scores.  The kernels and RCCL cannot run here;
each rank scores its share with the oracle (test infrastructure) and gathers with aln_amd.shard.GlooComm, which has the
interface of the RCCL-backed aln_amd.shard.Comm (tests/test_gpu_comm.py covers that one on the GPU box), so that the
gathered vector can be checked against a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "alignment-algos_amd")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import orc
    from aln_amd.shard import GlooComm, deal_units, local_units
    from aln_amd.synth import random_pair
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    alpha, table = orc.load_blosum(os.path.join(ROOT, "tests", "golden", "BLOSUM62"))
    work = [(20 + p + 2) * (25 + 2) for p in range(n_total)]          # Q*T of pair p
    owner, slot = deal_units(work, world)
    mine = local_units(owner, slot, rank)
    local = []
    for p in mine:
        p = int(p)
        a, b = random_pair(1000 + p, 20 + p, 25)
        S = orc.sim_submatrix(a, b, alpha, table)
        rc, D, PQ, PT = orc.dp_build(S, orc.Gap(orc.LOCAL, 11, 1))
        local.append(orc.optimal(D, PQ, PT, True)[1])
    n_max = int(np.bincount(owner, minlength=world).max())
    allv = GlooComm(world, rank).gather(np.array(local, np.float32), mine, n_max, n_total)
    dist.barrier()
    if rank == 0:
        q.put(allv.tolist())
    dist.destroy_process_group()


def test_owned_ranges_partition_the_pair_list():
    sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
    from aln_amd.shard import owned_range
    for n in (0, 1, 7, 8, 1024, 4097):
        for w in (1, 2, 3, 8):
            r = [owned_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_sorted_deal_is_a_balanced_permutation():
    """aln_deal_units: every unit gets exactly one (owner, slot), local lists are sorted by work descending, and the
    per-rank sums of work differ by no more than the largest unit (boustrophedon deal of a sorted list)."""
    sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
    from aln_amd.shard import deal_units, local_units
    rng = np.random.RandomState(5)
    for n in (0, 1, 7, 64, 1000, 4096):
        for w in (1, 2, 3, 8):
            work = (rng.randint(402, 603, size=n).astype(np.int64) * rng.randint(402, 603, size=n)) if n else np.zeros(0, np.int64)
            owner, slot = deal_units(work, w)
            assert ((owner >= 0) & (owner < w)).all()
            seen = np.zeros(n, dtype=bool)
            sums = []
            for r in range(w):
                mine = local_units(owner, slot, r)
                assert sorted(slot[mine].tolist()) == list(range(len(mine)))
                assert not seen[mine].any()
                seen[mine] = True
                wl = work[mine]
                assert (np.diff(wl) <= 0).all()                         # long units first
                sums.append(int(wl.sum()))
            assert seen.all()
            cnt = np.bincount(owner, minlength=w) if n else np.zeros(w, int)
            assert cnt.max() - cnt.min() <= 1
            if n >= w:
                assert max(sums) - min(sums) <= int(work.max())
    # equal work: ties keep the lower index first
    owner, slot = deal_units([5, 5, 5, 5, 5], 2)
    assert owner.tolist() == [0, 1, 1, 0, 0] and slot.tolist() == [0, 0, 1, 1, 2]


def test_two_rank_gather_equals_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "alignment-algos_amd"))
    import orc
    from aln_amd.synth import random_pair
    n_total = 7                      # odd: the two blocks differ by one
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=60)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    alpha, table = orc.load_blosum(os.path.join(ROOT, "tests", "golden", "BLOSUM62"))
    want = []
    for p in range(n_total):
        a, b = random_pair(1000 + p, 20 + p, 25)
        S = orc.sim_submatrix(a, b, alpha, table)
        rc, D, PQ, PT = orc.dp_build(S, orc.Gap(orc.LOCAL, 11, 1))
        want.append(float(orc.optimal(D, PQ, PT, True)[1]))
    assert got == want


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment: the parent starts two rank processes of itself before any
    GPU call, they rendezvous on 127.0.0.1, rotate their launches, gather ONCE per step, agree on the batch shape, and rank 0's
    single JSON line comes back on the parent's stdout with exit code 0.  ALN_BENCH_REHEARSE=cpu replaces the resident batches by
    stand-ins (no GPU here) — the control flow around them is the one the GPU run uses."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ALN_BENCH_REHEARSE"] = "cpu"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "8",
                        "--streams", "2", "--split", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, r.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["steps"] == 3 and doc["scaling"] == "weak" and "rehearsal" in doc
    assert doc["config"]["gathers"] == 4 and doc["config"]["launches_per_step"] == 2        # warmup + steps gathers, not one per launch


def test_bench_ranks_that_disagree_fail_instead_of_hanging():
    """Two ranks started by hand with different --pairs: the agreement step before the first collective ends both."""
    import subprocess
    port = _free_port()
    procs = []
    for r, pairs in enumerate(("8", "16")):
        env = dict(os.environ, ALN_BENCH_REHEARSE="cpu", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", pairs,
                                       "--streams", "1", "--split", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode != 0 for p in procs), outs
    assert any("disagree" in o[1] for o in outs), outs
