"""Multi-GPU plumbing for the pair-sharded path — a thin binding of the C ABI's aln_deal_units /
aln_comm_* / aln_gather_scores (csrc/aln_comm.hip: length-sorted deal, ONE RCCL all-gather of (index, score) records).

Pairs are independent, so rank r owns the units the deal gives it and the only collective is the gather of the fp32
scores.  The 128-byte RCCL id reaches the other processes through the job's own transport; here that is torch.distributed
(whatever backend the job initialised).  `GlooComm` is the same interface over torch.distributed alone, for the CPU tests of
the partition logic (no RCCL without a GPU)."""
import ctypes as C

import numpy as np


def owned_range(n_total, world, rank):
    """Contiguous block partition: the first (n_total % world) ranks get one extra unit."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def deal_units(work, world):
    """aln_deal_units: units sorted by work descending, dealt boustrophedon over `world` ranks.
    -> owner[n] (rank of each unit), slot[n] (position in that rank's local list, long units first)."""
    import aln_amd
    w = np.ascontiguousarray(work, dtype=np.int64)
    owner = np.zeros(len(w), dtype=np.int32)
    slot = np.zeros(len(w), dtype=np.int32)
    rc = aln_amd.lib().aln_deal_units(w.ctypes.data_as(C.POINTER(C.c_int64)), len(w), int(world),
                                      owner.ctypes.data_as(C.POINTER(C.c_int32)), slot.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise aln_amd.AlnError(rc, "aln_deal_units")
    return owner, slot


def local_units(owner, slot, rank):
    """Global indices of rank's units in local-list order."""
    idx = np.nonzero(owner == rank)[0]
    return idx[np.argsort(slot[idx], kind="stable")].astype(np.int32)


class Comm:
    """aln_comm over RCCL: one process per GPU.  `ctx` is this rank's aln_amd.Context; the id is broadcast with torch.distributed."""

    def __init__(self, ctx, world, rank):
        import aln_amd
        self.world, self.rank, self.ctx = world, rank, ctx
        L = aln_amd.lib()
        idbuf = (C.c_uint8 * aln_amd.COMM_ID_BYTES)()
        if world > 1:
            import torch
            import torch.distributed as dist
            if rank == 0:
                rc = L.aln_comm_unique_id(idbuf)
                if rc != 0:
                    raise aln_amd.AlnError(rc, "aln_comm_unique_id: " + L.aln_comm_last_error(None).decode())
            obj = [bytes(idbuf) if rank == 0 else None]
            dist.broadcast_object_list(obj, src=0)
            idbuf = (C.c_uint8 * aln_amd.COMM_ID_BYTES).from_buffer_copy(obj[0])
            idp = C.cast(idbuf, C.c_void_p)
        else:
            idp = None
        self.h = C.c_void_p()
        arr = (C.c_void_p * 1)(ctx.h)
        rc = L.aln_comm_create(arr, 1, idp, world, rank, C.byref(self.h))
        if rc != 0:
            raise aln_amd.AlnError(rc, "aln_comm_create: " + L.aln_last_error(ctx.h).decode() + " " + L.aln_comm_last_error(None).decode())

    def gather(self, local_scores, global_index, n_max, n_total, out=None):
        """aln_gather_scores: -> float32[n_total] in global order, on every rank."""
        import aln_amd
        sc = np.ascontiguousarray(local_scores, dtype=np.float32)
        gi = np.ascontiguousarray(global_index, dtype=np.int32)
        if out is None:
            out = np.zeros(n_total, dtype=np.float32)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        scp = (fp * 1)(sc.ctypes.data_as(fp))
        gip = (ip * 1)(gi.ctypes.data_as(ip))
        nl = (C.c_int32 * 1)(len(sc))
        rc = aln_amd.lib().aln_gather_scores(self.h, scp, gip, nl, int(n_max), out.ctypes.data_as(fp), int(n_total))
        if rc != 0:
            raise aln_amd.AlnError(rc, "aln_gather_scores: " + aln_amd.lib().aln_comm_last_error(self.h).decode())
        return out

    def close(self):
        import aln_amd
        if self.h:
            aln_amd.lib().aln_comm_destroy(self.h)
            self.h = C.c_void_p()


class GlooComm:
    """The same gather over torch.distributed alone (CPU tests of the partition logic; bench.py's one-GPU rehearsal)."""

    def __init__(self, world, rank):
        self.world, self.rank = world, rank

    def gather(self, local_scores, global_index, n_max, n_total, out=None):
        import torch
        import torch.distributed as dist
        if out is None:
            out = np.zeros(n_total, dtype=np.float32)
        rec = np.zeros((n_max, 2), dtype=np.int32)                  # the C ABI's record: int32 index (-1 = padding), fp32 bits
        rec[:, 0] = -1
        rec[:len(local_scores), 0] = np.asarray(global_index, dtype=np.int32)
        rec[:len(local_scores), 1] = np.ascontiguousarray(local_scores, dtype=np.float32).view(np.int32)
        if self.world == 1:
            allr = rec[None]
        else:
            buf = torch.from_numpy(rec.reshape(-1))
            got = torch.empty(self.world * buf.numel(), dtype=torch.int32)
            dist.all_gather_into_tensor(got, buf)
            allr = got.numpy().reshape(self.world, n_max, 2)
        for r in range(allr.shape[0]):
            m = allr[r, :, 0] >= 0
            out[allr[r, m, 0]] = np.ascontiguousarray(allr[r, m, 1]).view(np.float32)
        return out

    def close(self):
        pass
