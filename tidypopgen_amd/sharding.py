"""SNP-block sharding across the GPUs of one node: host-side helpers.

The locus axis is the reference's own block axis (R/snp_ibs.R:59-82 cuts colInd into blocks and sums the per-block
N x N increments), so shards are contiguous locus ranges and every quantity on the hot path is either a disjoint
per-locus slice (no exchange) or additive over loci.  The exchanges themselves are done by the library, over RCCL
(csrc/comm.hip; include/tpg.h "SNP-block shards"):

    pairwise cross-products V, D, H, A      int32, exact and order independent   -> one reduce-scatter; rank r then
                                                                                     finishes band r of the tiles
    Fst numerator / denominator sums        2P doubles per method                -> all-reduce
    PCA Gram matrix, squared Frobenius norm FP64                                  -> all-reduce

This module only restates the two partitions in Python (which loci a rank owns, which rows of the N x N outputs a
rank finishes) for callers and tests that have no GPU, and keeps the gloo all-reduce used by rehearsals.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib


def shard_loci(m_total: int, world: int, rank: int, align: int = 128):
    """Contiguous [begin, end) locus range of `rank`; boundaries are multiples of 128 (the K-group width of the
    packed layouts) except the last end, sizes differ by at most 128 (tpg_shard_loci)."""
    if world < 1 or not (0 <= rank < world) or align != 128:
        raise ValueError("bad world/rank")
    b, e = C.c_int64(), C.c_int64()
    check(lib.tpg_shard_loci(C.c_int64(m_total), C.c_int(world), C.c_int(rank), C.byref(b), C.byref(e)))
    return b.value, e.value


def band_rows(n: int, world: int, rank: int):
    """Rows [row0, row1) of the band rank `rank` finishes after the reduce-scatter of the pairwise slabs: its outputs
    cover rows [row0, row1) x columns [row0, n) and the mirror image rows [row0, n) x columns [row0, row1)
    (tpg_pairwise_band_of).  The bands of all ranks tile the N x N matrices."""
    a, b = C.c_int64(), C.c_int64()
    check(lib.tpg_pairwise_band_of(C.c_int64(n), C.c_int(world), C.c_int(rank), C.byref(a), C.byref(b)))
    return a.value, b.value


def band_mask(n: int, world: int, rank: int) -> np.ndarray:
    """boolean (n, n): the elements of the outputs rank `rank` writes"""
    r0, r1 = band_rows(n, world, rank)
    mask = np.zeros((n, n), dtype=bool)
    mask[r0:r1, r0:] = True
    mask[r0:, r0:r1] = True
    return mask


def all_reduce_numpy(a: np.ndarray, op: str = "sum") -> np.ndarray:
    """In-place all-reduce of a host array over the default torch.distributed process group (no-op without one):
    the control plane of bench.py (timing) and the host transport of rehearsals (api.Comm.host)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return a
    t = torch.from_numpy(a)
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return a
