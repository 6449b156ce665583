"""Multi-GPU plumbing for the pair-sharded path: pairs are independent, so rank r owns a contiguous
block of the global pair list and the only collective is one gather of the fp32 scores.  Backend-agnostic
(RCCL = "nccl" on the GPUs, "gloo" in the CPU tests)."""
import numpy as np


def owned_range(n_total, world, rank):
    """Contiguous block partition: the first (n_total % world) ranks get one extra pair."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_scores(local_scores, n_total, world, rank, device=None, stream=None):
    """All-gather the per-rank score blocks into the global order.  Blocks may differ by one element: pad to the
    largest block, gather once, strip the padding.  `stream`: a torch.cuda.Stream to run the (tiny) copies and the collective
    on, so that waiting for the gathered scores does not wait for compute kernels queued on the main stream."""
    if world == 1:
        return np.asarray(local_scores, dtype=np.float32)
    import contextlib
    import torch
    import torch.distributed as dist
    sizes = [owned_range(n_total, world, r)[1] - owned_range(n_total, world, r)[0] for r in range(world)]
    mx = max(sizes)
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        buf = torch.zeros(mx, dtype=torch.float32, device=device)
        buf[:len(local_scores)] = torch.as_tensor(np.asarray(local_scores, dtype=np.float32), device=device)
        out = torch.empty(mx * world, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(out, buf)
        out = out.cpu().numpy().reshape(world, mx)
    return np.concatenate([out[r, :sizes[r]] for r in range(world)])
