"""CPU, world_size 2 over gloo: the N > 1 path without a GPU.  Each rank computes its SNP shard's partial results
with the oracle as the compute stand-in, the partials are summed over gloo, every rank keeps what the library would
leave it with -- its band of the N x N matrices after the reduce-scatter, the all-reduced Fst sums and Gram matrix --
and the pieces must assemble to the whole-panel answer (exactly for the integer matrices, to rounding for Fst / Gram).
The partitions themselves (which loci, which band) come from the library (tpg_shard_loci, tpg_pairwise_band_of)."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M, G = 200, 1000, 4


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from tidypopgen_amd import sharding

    n, m = N, M
    fbm = orc.synth_fbm(5, n, m, npop=G, miss=0.05, imputed_bytes=True)
    gid = (np.arange(n) % G).astype(np.int32)
    b, e = sharding.shard_loci(m, world, rank)
    cols = np.arange(b + 1, e + 1, dtype=np.int32)
    out = {"rank": rank, "band": sharding.band_rows(n, world, rank)}
    # integer pairwise partials of this rank's loci, summed over the ranks; the rank keeps its band
    parts = []
    for inc in (orc.increment_ibs_counts, orc.increment_king_numerator, orc.increment_as_counts):
        A = np.zeros((n, n), order="F"); B = np.zeros((n, n), order="F")
        inc(A, B, fbm, None, cols)
        parts += [A, B]
    stack = np.ascontiguousarray(np.stack(parts))
    sharding.all_reduce_numpy(stack)
    mask = sharding.band_mask(n, world, rank)
    out["pairwise_band"] = np.where(mask, stack, np.nan)
    # Fst sums
    for method in ("Hudson", "WC84"):
        with np.errstate(invalid="ignore", divide="ignore"):
            nd = orc.pairwise_pop_fst(fbm, None, cols, gid, G, method=method, return_num_dem=True)
        num, den = nd["Fst_by_locus_num"], nd["Fst_by_locus_den"]
        ok = ~np.isnan(num) & ~np.isnan(den)
        sums = np.ascontiguousarray(np.stack([np.where(ok, num, 0).sum(axis=0), np.where(ok, den, 0).sum(axis=0)]))
        sharding.all_reduce_numpy(sums)
        out["fst_" + method] = sums[0] / sums[1]
    # PCA Gram (additive over loci; center/scale are per locus, hence local)
    dec = np.where(fbm > 3, fbm - 4, fbm)
    poly = (dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n)
    pc = cols[poly[b:e]]
    _, _, K = orc.pca_gram(fbm, None, pc)
    K = np.ascontiguousarray(K)
    sharding.all_reduce_numpy(K)
    out["gram"] = K
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_partitions():
    from tidypopgen_amd import sharding

    for m, w in ((1000, 2), (1_000_000, 8), (130, 4), (127, 3)):
        ranges = [sharding.shard_loci(m, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == m
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 == b0 and a0 % 128 == 0 and b0 % 128 == 0
    with pytest.raises(Exception):
        sharding.shard_loci(10, 2, 2)
    # the bands of all ranks tile the N x N matrices exactly once, for awkward sizes too
    for n, w in ((5000, 8), (1000, 8), (60, 2), (12, 4), (130, 3), (64, 5), (1, 2)):
        bands = [sharding.band_rows(n, w, r) for r in range(w)]
        assert bands[0][0] == 0 and bands[-1][1] == n
        for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
            assert a1 == b0 and a0 <= a1
        if n <= 1000:
            cover = sum(sharding.band_mask(n, w, r).astype(int) for r in range(w))
            assert cover.min() == 1 and cover.max() == 1
    # 8 ranks at the bench size: the bands are padded to equal chunks for the reduce-scatter; the padding (= the
    # imbalance of the bands) stays under 12 % of the unsharded buffer
    from tidypopgen_amd._lib import lib
    import ctypes as C

    one = lib.tpg_pairwise_buffer_bytes(C.c_int64(5000))
    for w in (2, 4, 8):
        assert one <= lib.tpg_pairwise_buffer_bytes_sharded(C.c_int64(5000), C.c_int(w)) <= 1.12 * one, w


@pytest.mark.timeout(120)
def test_two_rank_reduction_equals_whole_panel():
    from oracle import oracle as orc

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=100), q.get(timeout=100)], key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    n, m = N, M
    fbm = orc.synth_fbm(5, n, m, npop=G, miss=0.05, imputed_bytes=True)
    gid = (np.arange(n) % G).astype(np.int32)
    whole = []
    for inc in (orc.increment_ibs_counts, orc.increment_king_numerator, orc.increment_as_counts):
        A = np.zeros((n, n), order="F"); B = np.zeros((n, n), order="F")
        inc(A, B, fbm, None, None)
        whole += [A, B]
    whole = np.stack(whole)
    # every element is delivered by exactly one rank, and it is the whole-panel value
    have = np.stack([~np.isnan(o["pairwise_band"]) for o in outs]).sum(axis=0)
    assert have.min() == 1 and have.max() == 1
    assembled = np.nansum(np.stack([o["pairwise_band"] for o in outs]), axis=0)
    assert np.array_equal(assembled, whole)
    assert outs[0]["band"][1] == outs[1]["band"][0] and 0 < outs[0]["band"][1] < n
    for o in outs:
        for method in ("Hudson", "WC84"):
            with np.errstate(invalid="ignore", divide="ignore"):
                t = orc.pairwise_pop_fst(fbm, None, None, gid, G, method=method)["fst_tot"]
            assert np.allclose(o["fst_" + method], t, rtol=1e-12)
    dec = np.where(fbm > 3, fbm - 4, fbm)
    cols = (np.where((dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n))[0] + 1).astype(np.int32)
    _, _, K = orc.pca_gram(fbm, None, cols)
    assert np.allclose(outs[0]["gram"], K, rtol=1e-12, atol=1e-9) and np.array_equal(outs[0]["gram"], outs[1]["gram"])
