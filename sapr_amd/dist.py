"""Multi-GPU plumbing: one process per GPU (torchrun), utterances sharded across ranks.

MFCC and Viterbi need no communication.  Baum-Welch needs exactly one collective per EM
iteration: an all-reduce (sum, float64) of the packed per-word sufficient statistics
(``stats[W, width]`` of ``sapr_estep_diag``, ~30 KB for 11 words) over RCCL/xGMI
(``torch.distributed`` backend "nccl"); every rank then runs the identical tiny M-step, so no
broadcast is needed.  Flat-start global statistics use the same pattern.  On CPU-only test
runs the same code path runs over the gloo backend.
"""
from __future__ import annotations

import numpy as np


def is_distributed() -> bool:
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False


def world():
    if not is_distributed():
        return 0, 1
    import torch.distributed as dist
    return dist.get_rank(), dist.get_world_size()


def allreduce_sum_(t):
    """In-place sum over ranks of a torch tensor (no-op when not distributed)."""
    if is_distributed():
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_sum_numpy(a: np.ndarray, device=None) -> np.ndarray:
    """Sum a float64 numpy array over ranks (goes through a tensor on `device` for nccl)."""
    if not is_distributed():
        return a
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
    if dist.get_backend() == "nccl":
        t = t.to(device or torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def shard_range(n_items: int, rank: int = None, world_size: int = None):
    """Contiguous block [lo, hi) of `n_items` owned by `rank`."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
