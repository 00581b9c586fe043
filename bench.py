#!/usr/bin/env python3
"""bench.py -- throughput of the tidypopgen genotype-matrix hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one synthetic SNP panel that is already
resident in HBM as FBM bytes (5 000 individuals x 1 000 000 SNPs per GPU, 51 populations, 2 %
missing kept as imputed bytes -- BASELINE.json configs[2..4]):

    pack (raw view)  -> loci_alt_freq / missingness counts -> grouped_alt_freq (51 pops)
    -> pairwise_pop_fst Hudson + WC84 -> IBS + KING + allele-sharing/GRM cross-products (int8 MFMA)
    -> [N > 1: all-reduce of the integer N x N partials and of the Fst numerator/denominator sums]
    -> IBS / KING / GRM epilogues
    -> pack (imputed view) -> gt_pca_partialSVD (k = 20): center/scale, Gram, eigen, loadings

Multi-GPU: SNP blocks shard across ranks (weak scaling: every rank owns 1 000 000 loci of a panel
that is world_size times longer); one process per GPU, torch.distributed (RCCL).
Rank 0 prints ONE JSON line.  `value` = N*M_total genotypes / wall second of the whole step.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--indiv", "--n", dest="n", type=int, default=5000, help="individuals")
    ap.add_argument("--snps", "--m", dest="m", type=int, default=1000000, help="SNPs per GPU")
    ap.add_argument("--pops", type=int, default=51)
    ap.add_argument("--k", type=int, default=20, help="principal components")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-loci", type=int, default=12288)
    ap.add_argument("--digest", default=None, help="rank 0 writes a small JSON digest of the results (tests)")
    return ap.parse_args()


class Step:
    """Holds the resident panel and runs one pass of the hot path through the C ABI."""

    def __init__(self, args, rank, world, local_rank):
        import tidypopgen_amd as tpg
        from tidypopgen_amd import api

        self.tpg, self.api, self.lib = tpg, api, tpg._lib.lib
        self.args, self.rank, self.world = args, rank, world
        self.torch = None
        self.backend = os.environ.get("TPG_BENCH_BACKEND", "nccl")
        if os.environ.get("TPG_BENCH_SHARE_GPU") == "1":
            local_rank = 0  # rehearsal: several ranks on one GPU (gloo only; RCCL refuses duplicate devices)
        if world > 1 or os.environ.get("TPG_BENCH_FORCE_DIST") == "1":  # FORCE_DIST: rehearse the collective path on 1 rank
            import torch
            import torch.distributed as dist

            self.torch, self.dist = torch, dist
            if self.backend == "nccl":
                torch.cuda.set_device(local_rank)
        from tidypopgen_amd import sharding

        self.sharding = sharding
        self.ctx = tpg.Context(local_rank)
        n, m, G = args.n, args.m, args.pops
        # FBM bytes resident in HBM before the timed region (the "input"); imputed bytes 4..6 where missing
        self.X = tpg.FBM.synth(3, n, m, j0=rank * m, npop=G, miss=0.02, imputed_bytes=True, ctx=self.ctx,
                               code256=tpg.CODE_012)
        self.gid = (np.arange(n) % G).astype(np.int32)
        self.ploidy = np.full(n, 2.0)
        self.pairs = np.ascontiguousarray(api.combn2(G).T)
        self.P = self.pairs.shape[0]
        self.code_imp = np.ascontiguousarray(tpg.CODE_IMPUTE_PRED)
        self.code_012 = np.ascontiguousarray(tpg.CODE_012)
        # device-resident outputs (PCIe excluded from `value`; see DESIGN.md for the inclusive rate)
        self.d_freq = self._dalloc(8 * 2 * m)
        self.d_gfreq = self._dalloc(8 * 2 * G * m)
        self.d_nn = [self._dalloc(8 * n * n) for _ in range(3)]  # IBS, KING, GRM
        self.pw_bytes = tpg.Pairwise.buffer_bytes(n)
        self.pw_tensor = self.K_tensor = None
        if self.torch is not None and self.backend == "nccl":
            # collectives need torch tensors: the library accumulates straight into them
            self.pw_tensor = self.torch.zeros(self.pw_bytes // 4, dtype=self.torch.int32, device="cuda")
            self.pw = tpg.Pairwise(self.ctx, n, ext_buffer=self.pw_tensor.data_ptr())
            self.K_tensor = self.torch.zeros(n * n, dtype=self.torch.float64, device="cuda")
            self.d_K = C.c_void_p(self.K_tensor.data_ptr())
        else:
            if self.torch is not None:  # gloo rehearsal: an external buffer whose address we know
                self.pw_dev = self._dalloc(self.pw_bytes)
                self.pw = tpg.Pairwise(self.ctx, n, ext_buffer=self.pw_dev.value)
            else:
                self.pw_dev = None
                self.pw = tpg.Pairwise(self.ctx, n)
            self.d_K = self._dalloc(8 * n * n)
        self.fst = {}
        self.has_pca = True
        # PCA setup (untimed): big_SVD stops on a zero scale, so monomorphic loci are dropped beforehand,
        # as a MAF filter does in the reference workflow (vignettes/articles/benchmark_hgdp.Rmd)
        vi = api.View(self.X, None, None, code256=self.code_imp)
        cnt = api.loci_counts(vi)
        vi.free()
        alt = cnt[:, 1] + 2 * cnt[:, 2]
        poly = (alt > 0) & (alt < 2 * n)
        self.pca_cols = None if poly.all() else (np.where(poly)[0] + 1).astype(np.int32)
        self.m_pca = int(poly.sum())
        k = args.k
        self.d_pca = {"u": self._dalloc(8 * n * k), "v": self._dalloc(8 * self.m_pca * k),
                      "center": self._dalloc(8 * self.m_pca), "scale": self._dalloc(8 * self.m_pca)}
        self.pca_d = np.zeros(k)
        self.pca_fro = C.c_double()

    def _dalloc(self, nbytes):
        p = C.c_void_p()
        self.tpg._lib.check(self.lib.tpg_dev_alloc(self.ctx.h, C.c_size_t(nbytes), C.byref(p)))
        return p

    def barrier_sync(self):
        self.ctx.sync()
        if self.torch is not None:
            self.dist.barrier()
            if self.backend == "nccl":
                self.torch.cuda.synchronize()

    def _all_reduce_dev(self, dptr, tensor, nbytes, dtype):
        """sum a device buffer over ranks: RCCL on the aliasing torch tensor, or through host memory (gloo)"""
        self.ctx.sync()
        if tensor is not None:
            self.dist.all_reduce(tensor)
            self.torch.cuda.synchronize()
            return
        host = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        chk = self.tpg._lib.check
        chk(self.lib.tpg_dev_to_host(self.ctx.h, self.api._ptr(host), dptr, C.c_size_t(nbytes)))
        self.sharding.all_reduce_numpy(host)
        chk(self.lib.tpg_dev_from_host(self.ctx.h, dptr, self.api._ptr(host), C.c_size_t(nbytes)))

    def run(self):
        tpg, api, lib, ctx, a = self.tpg, self.api, self.lib, self.ctx, self.args
        chk = tpg._lib.check
        n, m, G = a.n, a.m, a.pops
        P = self.P
        # ---- raw view (bytes 0/1/2 valid, the rest missing) ----
        v = api.View(self.X, None, None, code256=self.code_012)
        chk(lib.tpg_alt_freq_dip_pseudo(ctx.h, v.h, api._ptr(self.ploidy), C.c_int(0), self.d_freq))
        chk(lib.tpg_grouped_alt_freq_dip_pseudo(ctx.h, v.h, api._ptr(self.gid), C.c_int(G), api._ptr(self.ploidy),
                                                C.c_int(0), self.d_gfreq))
        sums = {}
        for name, code in (("Hudson", 0), ("WC84", 2)):
            sn, sd = np.zeros(P), np.zeros(P)
            chk(lib.tpg_pairwise_pop_fst_sums(ctx.h, v.h, api._ptr(self.gid), C.c_int(G), api._ptr(self.ploidy),
                                              C.c_int(code), api._ptr(self.pairs), C.c_int(P), api._ptr(sn),
                                              api._ptr(sd)))
            sums[name] = (sn, sd)
        self.pw.zero()
        self.pw.accumulate(v)
        if self.torch is not None:
            # data-path exchanges: integer N x N partials (exact, order independent) ...
            pw_ptr = C.c_void_p(self.pw_tensor.data_ptr()) if self.pw_tensor is not None else self._pw_ptr()
            self._all_reduce_dev(pw_ptr, self.pw_tensor, self.pw_bytes, np.int32)
        for name in sums:  # ... and 2P doubles per Fst method (no-op on one rank)
            self.fst[name] = self.sharding.fst_from_sums(sums[name][0], sums[name][1])
        chk(lib.tpg_pairwise_epilogues(ctx.h, self.pw.h, C.c_int(0), C.c_int64(m * self.world), self.d_nn[0],
                                       self.d_nn[1], C.c_void_p(None), self.d_nn[2]))
        v.free()
        # ---- imputed view + PCA ----
        if self.has_pca:
            try:
                self.pca = self._pca()
            except tpg._lib.TpgError as e:
                if e.code != 3:
                    raise
                self.has_pca = False
        ctx.sync()

    def _pw_ptr(self):
        if self.pw_dev is None:
            # the library-owned accumulator: fetch its address once through the buffer-bytes contract
            raise RuntimeError("gloo rehearsal needs an external pairwise buffer")
        return self.pw_dev

    def _pca(self):
        api, lib, ctx, k = self.api, self.lib, self.ctx, self.args.k
        chk = self.tpg._lib.check
        v = api.View(self.X, None, self.pca_cols, code256=self.code_imp)
        if self.torch is None:
            chk(lib.tpg_pca_partial_svd(ctx.h, v.h, C.c_int(k), api._ptr(self.pca_d), self.d_pca["u"], self.d_pca["v"],
                                        self.d_pca["center"], self.d_pca["scale"], C.byref(self.pca_fro)))
        else:
            # SNP shards: local center/scale and Gram, one N x N all-reduce, replicated eigen step, local loadings
            n = self.args.n
            chk(lib.tpg_pca_center_scale(ctx.h, v.h, self.d_pca["center"], self.d_pca["scale"]))
            chk(lib.tpg_pca_gram(ctx.h, v.h, self.d_pca["center"], self.d_pca["scale"], self.d_K))
            self._all_reduce_dev(self.d_K, self.K_tensor, 8 * n * n, np.float64)
            lam = np.zeros(k)
            chk(lib.tpg_sym_eig_topk(ctx.h, self.d_K, C.c_int64(n), C.c_int(k), api._ptr(lam), self.d_pca["u"]))
            self.pca_d[:] = np.sqrt(np.maximum(lam, 0))
            chk(lib.tpg_pca_loadings(ctx.h, v.h, self.d_pca["center"], self.d_pca["scale"], self.d_pca["u"],
                                     api._ptr(self.pca_d), C.c_int(k), self.d_pca["v"]))
            fro = C.c_double()
            chk(lib.tpg_square_frobenius(ctx.h, v.h, self.d_pca["center"], self.d_pca["scale"], C.byref(fro)))
            self.pca_fro.value = float(self.sharding.all_reduce_numpy(np.array([fro.value]))[0])
        v.free()
        return self.pca_d


def digest(st):
    """A few numbers that pin every output of the step (used to compare sharded and unsharded runs)."""
    n, k = st.args.n, st.args.k
    chk = st.tpg._lib.check
    out = {"fst_hudson": st.fst["Hudson"].tolist(), "fst_wc84": st.fst["WC84"].tolist(),
           "pca_d": st.pca_d.tolist(), "pca_fro": st.pca_fro.value}
    for name, dptr in zip(("ibs", "king", "grm"), st.d_nn):
        a = np.zeros((n, n), order="F")
        chk(st.lib.tpg_dev_to_host(st.ctx.h, st.api._ptr(a), dptr, C.c_size_t(8 * n * n)))
        out[name + "_sum"] = float(np.nansum(a))
        out[name + "_corner"] = a[:6, :6].tolist()
        out[name + "_nan"] = int(np.isnan(a).sum())
    u = np.zeros((n, k), order="F")
    chk(st.lib.tpg_dev_to_host(st.ctx.h, st.api._ptr(u), st.d_pca["u"], C.c_size_t(8 * n * k)))
    out["pca_u_abs_colsum"] = np.abs(u).sum(axis=0).tolist()
    return out


def cpu_baseline(args):
    """The oracle's "as the reference does it" path (oracle/oracle.py: dense FP64 one-hot blocks + BLAS
    products for IBS/KING/AS, C loops for the per-locus statistics and Fst, numpy Gram for PCA), timed on
    this host's cores over a bounded sample of the same workload."""
    from oracle import oracle as orc

    n, G, B = args.n, args.pops, args.cpu_sample_loci
    fbm = orc.synth_fbm(3, n, B, npop=G, miss=0.02, imputed_bytes=True)
    r = np.arange(1, n + 1, dtype=np.int32)
    c = np.arange(1, B + 1, dtype=np.int32)
    gid = (np.arange(n) % G).astype(np.int32)
    ploidy = np.full(n, 2.0)
    t0 = time.time()
    K = [np.zeros((n, n)) for _ in range(6)]
    orc.blas_increment_ibs(K[0], K[1], fbm, r, c)
    orc.blas_increment_king(K[2], K[3], fbm, r, c)
    orc.blas_increment_as(K[4], K[5], fbm, r, c)
    orc.king_epilogue(K[2], K[3])
    orc.pairwise_grm(orc.as_epilogue(K[4], K[5]))
    orc.alt_freq_dip_pseudo_cpp(fbm, r, c, ploidy)
    orc.grouped_alt_freq_dip_pseudo_cpp(fbm, r, c, gid, G, ploidy)
    with np.errstate(invalid="ignore", divide="ignore"):
        orc.pairwise_pop_fst(fbm, r, c, gid, G, method="Hudson")
        orc.pairwise_pop_fst(fbm, r, c, gid, G, method="WC84")
    # PCA Gram of the sample (K = Z Z' through BLAS), as bigstatsr::big_SVD accumulates it per block
    X = orc.CODE_IMPUTE_PRED[fbm]
    center = X.mean(axis=0)
    p = center / 2
    scale = np.sqrt(2 * p * (1 - p))
    keep = scale > 0
    Z = (X[:, keep] - center[keep]) / scale[keep]
    Z @ Z.T
    dt = time.time() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    try:  # the threads the BLAS products actually ran on (OpenBLAS caps them at its build maximum)
        from threadpoolctl import threadpool_info

        blas = [int(t["num_threads"]) for t in threadpool_info() if t.get("user_api") == "blas"]
        if blas:
            cores = min(cores, max(blas))
    except ImportError:
        pass
    return {"value": n * B / dt, "unit": "SNP-genotypes/s", "cores": cores, "kind": "port",
            "sample": f"{n} x {B} loci of the same synthetic panel, IBS+KING+AS via numpy/BLAS FP64 one-hot products "
                      f"(reference's 6/4/2 products per block, multi-threaded), per-locus + Fst (Hudson, WC84) C loops (one thread), PCA Gram via BLAS; "
                      f"{dt:.1f} s; eigen step excluded"}


def main():
    args = parse()
    # stdout carries exactly one JSON line: libraries that print to fd 1 (RCCL's version banner on communicator
    # creation, for one) are sent to stderr for the whole run, the result goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("TPG_BENCH_FORCE_DIST") == "1"
    if dist_on:
        import torch.distributed as dist

        dist.init_process_group(os.environ.get("TPG_BENCH_BACKEND", "nccl"))
    st = Step(args, rank, world, local_rank)
    for _ in range(args.warmup):
        st.run()
    st.ctx.prof_enable(True)
    st.ctx.prof_reset()
    st.barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.run()
    st.barrier_sync()
    dt = time.perf_counter() - t0
    if dist_on:
        dt = float(st.sharding.all_reduce_numpy(np.array([dt]), op="max")[0])
    prof = st.ctx.prof_dump()
    if rank == 0 and args.digest:
        with open(args.digest, "w") as f:
            json.dump(digest(st), f)
    if rank == 0:
        n, m = args.n, args.m
        total_genotypes = n * m * world
        ms_per_step = dt / args.steps * 1e3
        # dominant kernel: the int8 MFMA cross-product pass.  Algorithmic ops per launch: the fused
        # IBS+KING+AS pass needs 3 symmetric + 1 general product = 2.5 N^2 M MACs = 5 N^2 M ops.
        def mfma_roof(key, kernel, ops, note):
            cnt, ms = prof.get(key, (0, 0.0))
            if not cnt:
                return None
            avg_s = ms / cnt * 1e-3
            achieved = ops / avg_s / 1e12
            return {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": 5000.0, "unit": "TOP/s",
                    "frac": achieved / 5000.0, "traffic": TRAFFIC.get(key), "avg_launch_ms": ms / cnt,
                    "algorithmic_ops_per_launch": ops, "note": note}

        # HBM bytes per launch from the rocprofv3 --pmc passes committed under profiles/ (FETCH_SIZE doubled per the
        # gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE); null when the workload is not the profiled one
        TRAFFIC = {}
        if (n, m, args.pops, args.k) == (5000, 1000000, 51, 20):
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")) as f:
                    TRAFFIC = json.load(f)["hbm_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                TRAFFIC = {}
        m_pca = st.m_pca if st.has_pca else m
        roofs = [
            mfma_roof("pairwise_mfma", "tpg_pairwise_kernel (v_mfma_i32_32x32x32_i8)", 5.0 * n * n * m,
                      "fused IBS+KING+AS/GRM: 3 symmetric + 1 general int8 product = 2.5 N^2 M MACs"),
            mfma_roof("pca_gram_mfma", "tpg_pca_gram_kernel (v_mfma_i32_32x32x32_i8)", 4.0 * n * n * m_pca,
                      "PCA Gram: 4 weight digits x symmetric int8 product = 4 * N^2 M / 2 MACs"),
        ]
        roofs = [r for r in roofs if r]
        roofs.sort(key=lambda r: -r["avg_launch_ms"])
        roof = roofs[0] if roofs else None

        # the HBM-bound kernels of the path, priced on their algorithmic bytes
        def hbm_roof(key, kernel, nbytes, note):
            cnt, ms = prof.get(key, (0, 0.0))
            if not cnt:
                return None
            achieved = nbytes / (ms / cnt * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                    "frac": achieved / 8000.0, "traffic": None, "avg_launch_ms": ms / cnt,
                    "algorithmic_bytes_per_launch": nbytes, "note": note}

        hbm_roofs = [
            hbm_roof("pack", "tpg_pack_fast_kernel", 1.5 * n * m, "FBM bytes -> two 2-bit layouts: N M read + N M / 2 written"),
            hbm_roof("loci_counts", "tpg_loci_counts_kernel", 0.25 * n * m + 16.0 * m,
                     "per-locus genotype counts: N M / 4 read + 16 B per locus written"),
        ]
        analyses = ["pack", "loci_alt_freq", "grouped_alt_freq", "fst_hudson", "fst_wc84", "ibs", "king", "grm"]
        if st.has_pca:
            analyses.append(f"pca_partialSVD_k{args.k}")
        out = {
            "metric": "SNP-genotypes/s (N x M) for IBS+KING+GRM+Fst+PCA",
            "value": total_genotypes / (dt / args.steps),
            "unit": "SNP-genotypes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8 (int32 accumulate) for counts and cross-products, f64 for statistics",
            "data": "synthetic",
            "config": {"workload": f"{n} individuals x {m} SNPs per GPU ({world} GPU), 51 populations, 2% missing "
                                   f"(imputed bytes), seed 3 [BASELINE configs 2-4]",
                       "analyses": analyses, "pca_included": bool(st.has_pca)},
            "roofline": roof,
            "roofline_other_kernels": roofs[1:] + [r for r in hbm_roofs if r],
            "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())},
            "kernel_launches_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items())},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        st.dist.destroy_process_group()


if __name__ == "__main__":
    main()
