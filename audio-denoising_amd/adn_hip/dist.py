"""Multi-GPU plumbing: one process per GPU, torch.distributed (RCCL on ROCm, gloo on CPU).

The hot path shards by file / clip / block - units that are independent in the reference
too - so there is no collective on the data path.  The only exchange is the small
noise-magnitude profile (n_fft/2+1 values) when every rank denoises with one shared
noise sample: the owner rank learns it and broadcasts it.
"""
import numpy as np


def shard_indices(n_items, rank, world):
    """Round-robin assignment of item indices to ranks (files of similar length balance)."""
    return np.arange(rank, n_items, world, dtype=np.int64)


def is_distributed():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def broadcast_profile(profile, n_bins, src=0, device=None, dtype=None, group=None):
    """Share the learnt noise profile: `profile` is a tensor of n_bins values on rank `src`
    and is ignored (may be None) elsewhere.  Returns the profile on every rank."""
    import torch
    import torch.distributed as dist
    if not is_distributed():
        return profile
    if dist.get_rank(group) == src:
        buf = profile.detach().clone().reshape(-1)
        if buf.numel() != n_bins:
            raise ValueError("profile has %d bins, expected %d" % (buf.numel(), n_bins))
    else:
        buf = torch.empty(n_bins, dtype=dtype, device=device)
    dist.broadcast(buf, src=src, group=group)
    return buf


def max_over_ranks(value, device=None, group=None):
    """Maximum of a host scalar over all ranks (elapsed-time reduction of the benchmark)."""
    import torch
    import torch.distributed as dist
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(value, device=None, group=None):
    import torch
    import torch.distributed as dist
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())
