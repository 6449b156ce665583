"""-m gpu: the multi-GPU entry points of the C ABI on the one GPU the box has — a 1-rank RCCL communicator
(aln_comm_create / aln_ctx_create_multi) and aln_gather_scores through it.  The N > 1 partition logic (aln_deal_units) is
covered on the CPU by tests/test_multirank_gloo.py; the N > 1 RCCL path itself only runs in the driver's 8-GPU bench."""
import ctypes as C

import numpy as np
import pytest

import aln_amd
import gpu_util
from aln_amd.shard import Comm, deal_units, local_units

pytestmark = pytest.mark.gpu


def test_one_rank_comm_gathers_scores_of_a_real_batch(blosum62):
    alpha, table = blosum62
    from aln_amd.synth import random_pair
    pairs = [random_pair(31000 + n, 40 + 7 * n, 55) for n in range(9)]
    work = [(len(q) + 2) * (len(t) + 2) for q, t in pairs]
    owner, slot = deal_units(work, 1)
    mine = local_units(owner, slot, 0)                      # rank 0 of 1 owns everything, longest pair first
    assert sorted(mine.tolist()) == list(range(9)) and (np.diff(np.array(work)[mine]) <= 0).all()
    ctx = gpu_util.ctx()
    b = aln_amd.Batch(ctx, [pairs[k][0] for k in mine], [pairs[k][1] for k in mine])
    b.dp_submatrix(alpha, table, aln_amd.LOCAL, 11, 1)
    scores, _, status = b.optimal(want_pairs=False)
    assert (status == 0).all()
    comm = Comm(ctx, 1, 0)
    for rep in range(3):                                    # buffers are reused
        out = comm.gather(scores, mine, len(mine), len(pairs))
        assert np.array_equal(out[mine].view(np.uint32), scores.view(np.uint32))
    # a rank that contributes fewer than n_max scores: the padding must not reach the output
    out = comm.gather(scores[:4], mine[:4], 9, len(pairs), out=np.full(len(pairs), -7, np.float32))
    assert np.array_equal(out[mine[:4]], scores[:4]) and (out[mine[4:]] == -7).all()
    comm.close()
    b.close()


def test_ctx_create_multi_single_device():
    """SURVEY 8(b)'s aln_ctx_create(device_ids, n): contexts + communicator for the devices of one process (n = 1 here)."""
    L = aln_amd.lib()
    dev = (C.c_int32 * 1)(0)
    ctxs = (C.c_void_p * 1)()
    comm = C.c_void_p()
    rc = L.aln_ctx_create_multi(dev, 1, ctxs, C.byref(comm))
    assert rc == 0, L.aln_comm_last_error(None)
    assert L.aln_comm_n_ranks(comm) == 1
    sc = np.array([3.5, 1.25, 9.0], np.float32)
    gi = np.array([2, 0, 1], np.int32)
    out = np.zeros(3, np.float32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    rc = L.aln_gather_scores(comm, (fp * 1)(sc.ctypes.data_as(fp)), (ip * 1)(gi.ctypes.data_as(ip)), (C.c_int32 * 1)(3), 3,
                             out.ctypes.data_as(fp), 3)
    assert rc == 0, L.aln_comm_last_error(comm)
    assert out.tolist() == [1.25, 9.0, 3.5]
    bad = np.array([5], np.int32)                           # an index outside the global list is an argument error
    rc = L.aln_gather_scores(comm, (fp * 1)(sc.ctypes.data_as(fp)), (ip * 1)(bad.ctypes.data_as(ip)), (C.c_int32 * 1)(1), 3,
                             out.ctypes.data_as(fp), 3)
    assert rc == aln_amd.E_ARG
    L.aln_comm_destroy(comm)
    L.aln_ctx_destroy(ctxs[0])
