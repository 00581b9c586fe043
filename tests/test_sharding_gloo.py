"""CPU, world_size 2 over gloo: the N > 1 path.  Each rank computes its SNP shard's partial results
with the oracle as the compute stand-in, the package's sharding helpers reduce them, and the result must
equal the whole-panel answer (exactly for the integer matrices, to rounding for Fst / Gram)."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from tidypopgen_amd import sharding

    n, m, G = 60, 1000, 4
    fbm = orc.synth_fbm(5, n, m, npop=G, miss=0.05, imputed_bytes=True)
    gid = (np.arange(n) % G).astype(np.int32)
    b, e = sharding.shard_loci(m, world, rank)
    cols = np.arange(b + 1, e + 1, dtype=np.int32)
    out = {}
    # integer pairwise partials
    parts = []
    for inc in (orc.increment_ibs_counts, orc.increment_king_numerator, orc.increment_as_counts):
        A = np.zeros((n, n), order="F"); B = np.zeros((n, n), order="F")
        inc(A, B, fbm, None, cols)
        parts += [A, B]
    stack = np.stack(parts).astype(np.int64)
    sharding.all_reduce_numpy(stack)
    out["pairwise"] = stack
    # Fst sums
    for method in ("Hudson", "WC84"):
        with np.errstate(invalid="ignore", divide="ignore"):
            nd = orc.pairwise_pop_fst(fbm, None, cols, gid, G, method=method, return_num_dem=True)
        num, den = nd["Fst_by_locus_num"], nd["Fst_by_locus_den"]
        ok = ~np.isnan(num) & ~np.isnan(den)
        sn = np.where(ok, num, 0).sum(axis=0); sd = np.where(ok, den, 0).sum(axis=0)
        out["fst_" + method] = sharding.fst_from_sums(sn, sd)
    # PCA Gram (additive over loci; center/scale are per locus, hence local)
    dec = np.where(fbm > 3, fbm - 4, fbm)
    poly = (dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n)
    pc = cols[poly[b:e]]
    _, _, K = orc.pca_gram(fbm, None, pc)
    sharding.all_reduce_numpy(K)
    out["gram"] = K
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_loci_partition():
    from tidypopgen_amd import sharding

    for m, w in ((1000, 2), (1_000_000, 8), (130, 4), (127, 3)):
        ranges = [sharding.shard_loci(m, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == m
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 == b0 and a0 % 128 == 0 and b0 % 128 == 0
    with pytest.raises(ValueError):
        sharding.shard_loci(10, 2, 2)


@pytest.mark.timeout(120)
def test_two_rank_reduction_equals_whole_panel():
    from oracle import oracle as orc

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=100)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    n, m, G = 60, 1000, 4
    fbm = orc.synth_fbm(5, n, m, npop=G, miss=0.05, imputed_bytes=True)
    gid = (np.arange(n) % G).astype(np.int32)
    whole = []
    for inc in (orc.increment_ibs_counts, orc.increment_king_numerator, orc.increment_as_counts):
        A = np.zeros((n, n), order="F"); B = np.zeros((n, n), order="F")
        inc(A, B, fbm, None, None)
        whole += [A, B]
    assert np.array_equal(out["pairwise"], np.stack(whole).astype(np.int64))
    for method in ("Hudson", "WC84"):
        with np.errstate(invalid="ignore", divide="ignore"):
            t = orc.pairwise_pop_fst(fbm, None, None, gid, G, method=method)["fst_tot"]
        assert np.allclose(out["fst_" + method], t, rtol=1e-12)
    dec = np.where(fbm > 3, fbm - 4, fbm)
    cols = (np.where((dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n))[0] + 1).astype(np.int32)
    _, _, K = orc.pca_gram(fbm, None, cols)
    assert np.allclose(out["gram"], K, rtol=1e-12, atol=1e-9)
