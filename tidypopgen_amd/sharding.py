"""SNP-block sharding across the GPUs of one node.

The locus axis is the reference's own block axis (R/snp_ibs.R:59-82 cuts colInd into blocks and sums the
per-block N x N increments), so shards are contiguous locus ranges and every quantity on the hot path is
either a disjoint per-locus slice (no collective) or additive over loci:

    pairwise cross-products V, D, H, A      int32, exact and order independent   -> one all-reduce (sum)
    Fst numerator / denominator sums        2P doubles per method                -> one all-reduce (sum)
    PCA Gram matrix, squared Frobenius norm FP64                                  -> one all-reduce (sum)

One process per GPU; collectives go through torch.distributed ("nccl" = RCCL over xGMI on the GPU
node; "gloo" moves the same buffers through host memory and is what the CPU tests and single-GPU
rehearsals use).  This module is host logic only.
"""
from __future__ import annotations

import numpy as np


def shard_loci(m_total: int, world: int, rank: int, align: int = 128):
    """Contiguous [begin, end) locus range of `rank`; boundaries are multiples of `align` (the K-group
    width of the packed layouts) except the last end, sizes differ by at most `align`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    groups = -(-m_total // align)
    g0 = groups * rank // world
    g1 = groups * (rank + 1) // world
    return min(g0 * align, m_total), min(g1 * align, m_total)


def all_reduce_numpy(a: np.ndarray, op: str = "sum") -> np.ndarray:
    """In-place all-reduce of a host array over the default process group (no-op without one)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return a
    t = torch.from_numpy(a)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    if t.is_cuda:
        a[...] = t.cpu().numpy()
    return a


def fst_from_sums(sum_num: np.ndarray, sum_den: np.ndarray) -> np.ndarray:
    """Fst = sum of numerators / sum of denominators over ALL loci (src/pairwise_fst_hudson_loop.cpp:43-52):
    shards add their sums first, then divide once."""
    sn = all_reduce_numpy(np.array(sum_num, dtype=float, copy=True))
    sd = all_reduce_numpy(np.array(sum_den, dtype=float, copy=True))
    with np.errstate(invalid="ignore", divide="ignore"):
        return sn / sd
