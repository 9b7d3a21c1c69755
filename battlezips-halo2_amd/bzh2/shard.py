"""Multi-GPU partition of independent proofs.

Every proof depends only on the shared read-only (params, pk) and its own witness,
so rank r of R proves a contiguous slice of the batch on its own GPU with no
data-path collective; the single collective is the gather of the finished,
fixed-stride proof/commitment records to every rank (RCCL all_gather over xGMI on
the GPU box, gloo in the CPU tests).
"""
from __future__ import annotations

import torch


def shard_range(total: int, rank: int, world: int) -> range:
    """Contiguous, balanced slice [lo, hi) of `total` units for `rank` (sizes differ by at most 1)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def gather_records(local: torch.Tensor, counts, dist=None) -> torch.Tensor:
    """all_gather fixed-stride records: `local` is (count_r, stride) on this rank, `counts[r]` the
    record count of every rank.  Returns the (sum(counts), stride) tensor in rank order on every rank."""
    if dist is None or not dist.is_initialized():
        return local          # (a group of ONE rank -- bench.py --force-dist -- still runs the collective below)
    world = dist.get_world_size()
    stride = local.shape[1]
    cap = max(counts)
    padded = torch.zeros((cap, stride), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return torch.cat([out[r][: counts[r]] for r in range(world)], dim=0)


def combine_msm_partials(curve: int, local_jacobian, dist=None, form: int = 0):
    """One MSM whose points are split over the ranks (BASELINE.json configs[4], "8 GPUs"): every rank
    has computed sum_{i in shard_range(N, rank, world)} s_i G_i as one Jacobian point (12 limbs = 96 bytes).  The partials are
    all-gathered (world x 96 B -- there is no elliptic-curve reduction op to all_reduce with) and added locally by every rank.
    `local_jacobian`: (12,) uint64-as-int64 tensor or array in `form`.  Returns the (12,) uint64 numpy sum."""
    import numpy as np
    from . import jacobian_sum
    t = local_jacobian if isinstance(local_jacobian, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local_jacobian).view(np.int64))
    t = t.reshape(1, 12)
    parts = gather_records(t, [1] * (dist.get_world_size() if dist is not None and dist.is_initialized() else 1), dist)
    return jacobian_sum(curve, parts.cpu().numpy().view(np.uint64), form)
