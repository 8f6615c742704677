"""Multi-GPU plumbing: independent proofs shard by index across ranks -- no data-path collective.

One process per GPU (torch.distributed; backend "nccl" = RCCL on the GPU box, "gloo" in CPU tests).  The only
collectives are control-plane: a barrier around the timed region, a MAX over ranks of the elapsed time, and an
optional gather of per-rank proof digests on rank 0."""
import hashlib
import os


def shard_range(total, rank, world):
    """Contiguous, balanced partition of proof indices [0, total) -- rank r proves [lo, hi)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        kw = {"device_id": device} if device is not None else {}
        dist.init_process_group(backend=backend, **kw)
    return dist


def max_over_ranks(dist, value, device="cpu"):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_digests(dist, proofs):
    """rank 0 receives [(proof_index, sha256 hex)] from every rank: the 'host gathers proof buffers' step, digest-sized."""
    mine = [(i, hashlib.sha256(p).hexdigest()) for i, p in proofs]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(mine, out, dst=0)
    if out is None:
        return None
    return sorted(x for part in out for x in part)
