"""Host-side mirror of tidypopgen's interface for the genotype-matrix hot path.

The reference's host language is R; R is not available in this image, so the
host side above the C ABI is Python, with the reference's function names,
argument meaning and error behaviour (an error is raised where the reference
raises an R error):

    snp_ibs / snp_king / snp_allele_sharing   R/snp_ibs.R:42, R/snp_king.R:32, R/snp_allele_sharing.R:33
    pairwise_grm                              R/pairwise_grm.R:30
    loci_alt_freq / loci_missingness          R/loci_alt_freq.R:328, R/loci_missingness.R:97
    grouped_* kernels                         R/RcppExports.R:16-26
    pairwise_pop_fst                          R/pairwise_pop_fst.R:71
    gt_pca_partialSVD, fbm256_prod_and_rowSumsSq   R/gt_pca_partialSVD.R:67, R/predict_gt_pca.R:248

Index vectors are 1-based (as in R) and matrices come back as Fortran-ordered
numpy arrays (as R stores them).  All arithmetic runs on the GPU through
libtpg_hip.so; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, lib

CODE_012 = np.full(256, np.nan)
CODE_012[:3] = [0.0, 1.0, 2.0]
CODE_IMPUTE_PRED = np.full(256, np.nan)
CODE_IMPUTE_PRED[:3] = [0.0, 1.0, 2.0]
CODE_IMPUTE_PRED[4:7] = [0.0, 1.0, 2.0]

FST_METHODS = {"Hudson": 0, "Nei87": 1, "WC84": 2}
# cross-products of the pairwise accumulators (include/tpg.h: TPG_PW_*)
PW_V, PW_D, PW_H, PW_A = 1, 2, 4, 8
PW_FOR_AS, PW_FOR_IBS, PW_FOR_KING, PW_ALL = PW_V | PW_D, PW_V | PW_D | PW_H, PW_V | PW_D | PW_A, 15
PW_DH = 16  # D and H added up in one sum: all snp_ibs needs beside V (include/tpg.h)
PW_FOR_IBS_ALONE = PW_V | PW_DH


def _ptr(x):
    """numpy array -> its data pointer; int -> raw (device) pointer; None -> NULL.
    The caller must keep the array referenced until the C call returns (never pass a temporary)."""
    if x is None:
        return C.c_void_p(None)
    if isinstance(x, (int, np.integer)):
        return C.c_void_p(int(x))
    return C.c_void_p(x.ctypes.data)


def _i32(x):
    return None if x is None else np.ascontiguousarray(x, dtype=np.int32)


def _f64(x):
    return None if x is None else np.ascontiguousarray(x, dtype=np.float64)


class Context:
    """One GPU + one stream (tpg_ctx)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        check(lib.tpg_ctx_create(C.c_int(device), C.byref(h)))
        self.h = h
        self.device = device

    def set_stream(self, hip_stream: Optional[int]):
        check(lib.tpg_ctx_set_stream(self.h, C.c_void_p(hip_stream)))

    def dev_alloc(self, nbytes: int) -> "C.c_void_p":
        """raw device memory for outputs that stay in HBM between two library calls; release with dev_free"""
        p = C.c_void_p()
        check(lib.tpg_dev_alloc(self.h, C.c_size_t(int(nbytes)), C.byref(p)))
        return p

    def dev_free(self, p):
        lib.tpg_dev_free(p)

    def sync(self):
        check(lib.tpg_ctx_sync(self.h))

    def prof_enable(self, on: bool = True):
        check(lib.tpg_prof_enable(self.h, C.c_int(int(on))))

    def prof_reset(self):
        check(lib.tpg_prof_reset(self.h))

    def prof_only(self, names=None):
        """time only these launches (None: all); see include/tpg.h"""
        check(lib.tpg_prof_only(self.h, ",".join(names).encode() if names else None))

    def prof_get(self, prefix: str):
        ms = C.c_double()
        n = C.c_int64()
        check(lib.tpg_prof_get(self.h, prefix.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def prof_dump(self) -> dict:
        buf = C.create_string_buffer(1 << 16)
        check(lib.tpg_prof_dump(self.h, buf, C.c_size_t(len(buf))))
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split("\t")
            out[name] = (int(n), float(ms))
        return out

    def close(self):
        if self.h:
            lib.tpg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def device_count() -> int:
    """HIP devices visible to this process"""
    c = C.c_int()
    check(lib.tpg_device_count(C.byref(c)))
    return c.value


def bind_host_near_device(device: int = 0) -> int:
    """Keep this thread, and the threads started from it afterwards, on the host NUMA node of GPU `device`
    (tpg_host_bind_near_device: what `numactl --cpunodebind` does for a one-process-per-GPU launcher).  Returns the node,
    or -1 when nothing was done (one node, node unknown, too few of this process's CPUs on it)."""
    node = C.c_int(-1)
    check(lib.tpg_host_bind_near_device(C.c_int(device), C.byref(node)))
    return node.value


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class FBM:
    """Genotype bytes resident in HBM (the role bigstatsr's mmapped FBM.code256 plays)."""

    def __init__(self, ctx: Context, handle, nrow: int, ncol: int, code256=None):
        self.ctx, self.h, self.nrow, self.ncol = ctx, handle, nrow, ncol
        self.code256 = CODE_012 if code256 is None else np.asarray(code256, dtype=float)

    @classmethod
    def from_numpy(cls, bytes_2d, ctx: Optional[Context] = None, code256=None) -> "FBM":
        ctx = ctx or default_context()
        a = np.asarray(bytes_2d)
        if a.dtype != np.uint8 or a.ndim != 2:
            raise TypeError("FBM bytes must be a 2-D uint8 array (individuals x loci)")
        a = np.asfortranarray(a)
        h = C.c_void_p()
        check(lib.tpg_fbm_from_host(ctx.h, _ptr(a), C.c_int64(a.shape[0]), C.c_int64(a.shape[1]), C.byref(h)))
        return cls(ctx, h, a.shape[0], a.shape[1], code256)

    @classmethod
    def alloc(cls, nrow: int, ncol: int, ctx: Optional[Context] = None, code256=None) -> "FBM":
        """HBM for an FBM whose columns arrive block by block (upload_cols)"""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib.tpg_fbm_alloc(ctx.h, C.c_int64(nrow), C.c_int64(ncol), C.byref(h)))
        return cls(ctx, h, nrow, ncol, code256)

    def upload_cols(self, host_cols, col0: int, ctx: Optional[Context] = None):
        """columns [col0, col0 + k) <- host bytes (nrow x k, Fortran order, e.g. a slice of a numpy memmap of the .bk);
        pass another Context (another stream) to run the upload beside kernels of this FBM's own context"""
        a = np.asarray(host_cols)
        assert a.dtype == np.uint8 and a.ndim == 2 and a.flags.f_contiguous and a.shape[0] == self.nrow
        check(lib.tpg_fbm_upload_cols((ctx or self.ctx).h, self.h, _ptr(a), C.c_int64(col0), C.c_int64(a.shape[1])))

    @classmethod
    def open_bk(cls, path: str, nrow: int, ncol: int, ctx: Optional[Context] = None, code256=None) -> "FBM":
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib.tpg_fbm_open_bk(ctx.h, path.encode(), C.c_int64(nrow), C.c_int64(ncol), C.byref(h)))
        return cls(ctx, h, nrow, ncol, code256)

    @classmethod
    def open_bed(cls, path: str, n: int, m: int, ctx: Optional[Context] = None, code256=None) -> "FBM":
        """A PLINK .bed file as the genotype store (n, m = line counts of the .fam / .bim files)"""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib.tpg_fbm_open_bed(ctx.h, path.encode(), C.c_int64(n), C.c_int64(m), C.byref(h)))
        return cls(ctx, h, n, m, code256)

    @classmethod
    def alloc_bed(cls, n: int, m: int, ctx: Optional[Context] = None, code256=None) -> "FBM":
        """HBM for a .bed store whose SNPs arrive block by block (upload_bed_snps)"""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib.tpg_fbm_alloc_bed(ctx.h, C.c_int64(n), C.c_int64(m), C.byref(h)))
        return cls(ctx, h, n, m, code256)

    def upload_bed_snps(self, host_bytes, snp0: int, nsnps: int, ctx: Optional[Context] = None):
        """SNPs [snp0, snp0 + nsnps) <- nsnps * ceil(n / 4) payload bytes (e.g. a slice of a numpy memmap of the .bed behind
        its 3-byte magic); another Context (another stream) runs the upload beside kernels of this store's own context"""
        a = np.asarray(host_bytes)
        assert a.dtype == np.uint8 and a.flags.c_contiguous and a.size == nsnps * ((self.nrow + 3) // 4)
        check(lib.tpg_fbm_upload_bed_snps((ctx or self.ctx).h, self.h, _ptr(a), C.c_int64(snp0), C.c_int64(nsnps)))

    @classmethod
    def synth(cls, seed: int, nrow: int, ncol: int, j0: int = 0, npop: int = 51, miss: float = 0.02,
              imputed_bytes: bool = False, ctx: Optional[Context] = None, code256=None) -> "FBM":
        ctx = ctx or default_context()
        thr = min(int(round(miss * 2 ** 32)), 2 ** 32 - 1)
        h = C.c_void_p()
        check(lib.tpg_fbm_synth(ctx.h, C.c_uint64(seed), C.c_int64(nrow), C.c_int64(ncol), C.c_int64(j0),
                                C.c_int(npop), C.c_uint32(thr), C.c_int(int(imputed_bytes)), C.byref(h)))
        return cls(ctx, h, nrow, ncol, code256)

    def to_numpy(self) -> np.ndarray:
        out = np.zeros((self.nrow, self.ncol), dtype=np.uint8, order="F")
        check(lib.tpg_fbm_to_host(self.ctx.h, self.h, _ptr(out)))
        return out

    def free(self):
        if self.h:
            lib.tpg_fbm_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class View:
    """(FBM, rowInd, colInd, code256) packed to 2 bits in HBM (tpg_view)."""

    def __init__(self, X: FBM, ind_row=None, ind_col=None, code256="fbm"):
        self.X = X
        self.ctx = X.ctx
        r, c = _i32(ind_row), _i32(ind_col)
        if isinstance(code256, str):
            code = _f64(X.code256)
        else:
            code = _f64(code256)  # None = raw bytes
        h = C.c_void_p()
        check(lib.tpg_view_create(self.ctx.h, X.h, _ptr(r), C.c_int64(0 if r is None else len(r)), _ptr(c),
                                  C.c_int64(0 if c is None else len(c)), _ptr(code), C.byref(h)))
        self.h = h
        self.n = int(lib.tpg_view_n(h))
        self.m = int(lib.tpg_view_m(h))

    @classmethod
    def pair(cls, X: FBM, ind_row=None, ind_col=None, code256_a=None, code256_b=CODE_IMPUTE_PRED):
        """two views of the same rows / columns through two code tables from one read of the FBM bytes
        (tpg_view_create_pair): by default the raw view of the pairwise statistics and the imputed view of the PCA"""
        r, c = _i32(ind_row), _i32(ind_col)
        ca, cb = _f64(code256_a), _f64(code256_b)
        ha, hb = C.c_void_p(), C.c_void_p()
        check(lib.tpg_view_create_pair(X.ctx.h, X.h, _ptr(r), C.c_int64(0 if r is None else len(r)), _ptr(c),
                                       C.c_int64(0 if c is None else len(c)), _ptr(ca), _ptr(cb), C.byref(ha), C.byref(hb)))
        out = []
        for h in (ha, hb):
            v = cls.__new__(cls)
            v.X, v.ctx, v.h = X, X.ctx, h
            v.n, v.m = int(lib.tpg_view_n(h)), int(lib.tpg_view_m(h))
            out.append(v)
        return out[0], out[1]

    def unpack(self) -> np.ndarray:
        out = np.zeros((self.n, self.m), dtype=np.uint8, order="F")
        check(lib.tpg_view_unpack(self.ctx.h, self.h, _ptr(out)))
        return out

    def free(self):
        if self.h:
            lib.tpg_view_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Pairwise:
    """Integer N x N cross-product accumulators (tpg_pairwise)."""

    def __init__(self, ctx: Context, n: int, ext_buffer: Optional[int] = None):
        self.ctx, self.n = ctx, n
        h = C.c_void_p()
        check(lib.tpg_pairwise_create(ctx.h, C.c_int64(n), C.c_void_p(ext_buffer), C.byref(h)))
        self.h = h

    @staticmethod
    def buffer_bytes(n: int) -> int:
        return int(lib.tpg_pairwise_buffer_bytes(C.c_int64(n)))

    def zero(self):
        check(lib.tpg_pairwise_zero(self.ctx.h, self.h))

    def accumulate(self, view: View, col_begin: int = 0, col_end: int = -1, products=None):
        """products: None = all five cross-products; else an OR of PW_V / PW_D / PW_H / PW_A or one of the sets
        PW_FOR_AS / PW_FOR_IBS / PW_FOR_KING (the kernel specialised for that set runs: include/tpg.h)"""
        if products is None:
            check(lib.tpg_pairwise_accumulate(self.ctx.h, self.h, view.h, C.c_int64(col_begin), C.c_int64(col_end)))
        else:
            check(lib.tpg_pairwise_accumulate_products(self.ctx.h, self.h, view.h, C.c_int64(col_begin), C.c_int64(col_end),
                                                       C.c_int(int(products))))

    def products(self) -> int:
        """the products whose sums are complete since the last zero()"""
        return int(lib.tpg_pairwise_products(self.h))

    def set_as_pad_quirk(self, narrow_blocks: int):
        """opt-in emulation of reference quirk Q1 (include/tpg.h): +narrow_blocks on every allele-sharing numerator"""
        check(lib.tpg_pairwise_set_as_pad_quirk(self.h, C.c_int64(int(narrow_blocks))))

    def _mat(self):
        return np.zeros((self.n, self.n), order="F")

    def counts(self, which=("ibs", "ibs_valid", "king_num", "n_Aa_i", "as_num", "as_den")) -> dict:
        names = ("ibs", "ibs_valid", "king_num", "n_Aa_i", "as_num", "as_den")
        outs = {k: self._mat() for k in which}
        check(lib.tpg_pairwise_counts(self.ctx.h, self.h, *[_ptr(outs.get(k)) for k in names]))
        return outs

    def ibs(self, type: str = "proportion", m: int = 0) -> np.ndarray:
        out = self._mat()
        check(lib.tpg_pairwise_ibs(self.ctx.h, self.h, C.c_int(0 if type == "proportion" else 1), C.c_int64(m), _ptr(out)))
        return out

    def king(self) -> np.ndarray:
        out = self._mat()
        check(lib.tpg_pairwise_king(self.ctx.h, self.h, _ptr(out)))
        return out

    def allele_sharing(self) -> np.ndarray:
        out = self._mat()
        check(lib.tpg_pairwise_allele_sharing(self.ctx.h, self.h, _ptr(out)))
        return out

    def grm(self) -> np.ndarray:
        out = self._mat()
        check(lib.tpg_pairwise_grm(self.ctx.h, self.h, _ptr(out)))
        return out

    def epilogues(self, which=("ibs", "king", "allele_sharing", "grm"), ibs_type: str = "proportion", m: int = 0) -> dict:
        """IBS, KING, allele sharing and GRM from one pass over the accumulators"""
        names = ("ibs", "king", "allele_sharing", "grm")
        outs = {k: self._mat() for k in which}
        check(lib.tpg_pairwise_epilogues(self.ctx.h, self.h, C.c_int(0 if ibs_type == "proportion" else 1), C.c_int64(m),
                                         *[_ptr(outs.get(k)) for k in names]))
        return outs

    def free(self):
        if self.h:
            lib.tpg_pairwise_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Comm:
    """One rank of a group of GPUs that share an analysis sharded by loci (tpg_comm; include/tpg.h "SNP-block
    shards").  The library owns the collectives (RCCL); the launcher only has to deliver the 128-byte id."""

    def __init__(self, ctx: Context, handle, nranks: int, rank: int, keep=None):
        self.ctx, self.h, self.nranks, self.rank, self._keep = ctx, handle, nranks, rank, keep

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        check(lib.tpg_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def init_rank(cls, ctx: Context, nranks: int, rank: int, unique_id: Optional[bytes]) -> "Comm":
        h = C.c_void_p()
        idbuf = (C.c_uint8 * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        check(lib.tpg_comm_init_rank(ctx.h, C.c_int(nranks), C.c_int(rank), idbuf, C.byref(h)))
        return cls(ctx, h, nranks, rank)

    @classmethod
    def from_torch_distributed(cls, ctx: Context) -> "Comm":
        """Under torchrun: the process group (any backend -- gloo is enough) is only the control plane that carries
        the RCCL id from rank 0 to the others; the data path never goes through torch."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return cls.init_rank(ctx, 1, 0, None)
        # rank 0 always reaches the broadcast: a failure to make the id (no librccl.so ...) travels as a marker, so that
        # no rank is left waiting in the broadcast and every rank raises together
        box = [None]
        if dist.get_rank() == 0:
            try:
                box[0] = cls.unique_id()
            except Exception as e:  # noqa: BLE001
                box[0] = f"{type(e).__name__}: {e}"
        dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError(f"rank 0 could not create the RCCL id: {box[0]}")
        return cls.init_rank(ctx, dist.get_world_size(), dist.get_rank(), box[0])

    @classmethod
    def host(cls, ctx: Context, nranks: int, rank: int, allreduce) -> "Comm":
        """Rehearsal transport (tests): allreduce(numpy array) sums the array in place over the ranks through host
        memory -- e.g. torch.distributed over gloo with several ranks on one GPU."""
        def _cb(user, buf, count, dtype):
            try:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_int32 if dtype == 0 else C.c_double)), shape=(count,))
                allreduce(a)
                return 0
            except Exception:  # never let an exception cross the C boundary
                return 1

        cb = _lib.HOST_ALLREDUCE(_cb)
        h = C.c_void_p()
        check(lib.tpg_comm_init_host(ctx.h, C.c_int(nranks), C.c_int(rank), cb, None, C.byref(h)))
        return cls(ctx, h, nranks, rank, keep=cb)

    def shard_loci(self, m_total: int):
        b, e = C.c_int64(), C.c_int64()
        check(lib.tpg_shard_loci(C.c_int64(m_total), C.c_int(self.nranks), C.c_int(self.rank), C.byref(b), C.byref(e)))
        return b.value, e.value

    def allreduce_f64(self, buf, count: Optional[int] = None):
        """in-place sum over the ranks of a float64 numpy array, or of `count` doubles at a device pointer"""
        if isinstance(buf, np.ndarray):
            assert buf.dtype == np.float64 and buf.flags.c_contiguous or buf.flags.f_contiguous
            check(lib.tpg_comm_allreduce_f64(self.ctx.h, self.h, _ptr(buf), C.c_int64(buf.size)))
            return buf
        check(lib.tpg_comm_allreduce_f64(self.ctx.h, self.h, _ptr(buf), C.c_int64(int(count))))
        return buf

    def transport(self) -> str:
        """'none', 'host callback' or 'rccl: <library as loaded>'"""
        return lib.tpg_comm_transport(self.h).decode()

    def close(self):
        if self.h:
            lib.tpg_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedPairwise(Pairwise):
    """Pairwise accumulators of one rank: accumulate this rank's loci, reduce() (one reduce-scatter over the ranks),
    then counts() / epilogues() give this rank's band of the N x N outputs (band() tells which rows)."""

    def __init__(self, comm: Comm, n: int):
        self.ctx, self.n, self.comm = comm.ctx, n, comm
        h = C.c_void_p()
        check(lib.tpg_pairwise_create_sharded(comm.ctx.h, comm.h, C.c_int64(n), C.byref(h)))
        self.h = h

    def reduce(self):
        check(lib.tpg_pairwise_reduce(self.ctx.h, self.comm.h, self.h))

    def reduce_begin(self, side_comm: "Comm"):
        """the reduce-scatter on a second communicator (one made on another context of the same device): runs beside what this
        context enqueues next; nothing may read the accumulators until reduce_end (include/tpg.h; opt-in)"""
        check(lib.tpg_pairwise_reduce_begin(self.ctx.h, side_comm.h, self.h))

    def reduce_end(self, side_comm: "Comm"):
        check(lib.tpg_pairwise_reduce_end(self.ctx.h, side_comm.h, self.h))

    def band(self):
        a, b = C.c_int64(), C.c_int64()
        check(lib.tpg_pairwise_band(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def epilogues(self, which=("ibs", "king", "allele_sharing", "grm"), ibs_type: str = "proportion", m: int = 0) -> dict:
        names = ("ibs", "king", "allele_sharing", "grm")
        outs = {k: np.full((self.n, self.n), np.nan, order="F") for k in which}  # outside the band: left as NaN
        check(lib.tpg_pairwise_epilogues_sharded(self.ctx.h, self.comm.h, self.h, C.c_int(0 if ibs_type == "proportion" else 1),
                                                 C.c_int64(m), *[_ptr(outs.get(k)) for k in names]))
        return outs


class Multi:
    """One process, several GPUs (tpg_multi): what an R session uses."""

    def __init__(self, ndev: int, devices=None):
        h = C.c_void_p()
        dv = _i32(devices)
        check(lib.tpg_multi_create(C.c_int(ndev), _ptr(dv), C.byref(h)))
        self.h, self.ndev = h, ndev

    def transport(self) -> str:
        """transport of the device threads' communicators: 'none', 'host callback' or 'rccl: <library as loaded>'"""
        return lib.tpg_comm_transport(C.c_void_p(lib.tpg_multi_comm(self.h, C.c_int(0)))).decode()

    def pairwise(self, X_bytes, ind_row=None, ind_col=None, which=("ibs", "king", "allele_sharing", "grm"),
                 ibs_type: str = "proportion") -> dict:
        """snp_ibs / snp_king / snp_allele_sharing / pairwise_grm of a host FBM (uint8, Fortran order) on all devices"""
        X_bytes = np.asarray(X_bytes)
        assert X_bytes.dtype == np.uint8 and X_bytes.flags.f_contiguous
        r, c = _i32(ind_row), _i32(ind_col)
        n = X_bytes.shape[0] if r is None else len(r)
        m = X_bytes.shape[1] if c is None else len(c)
        names = ("ibs", "king", "allele_sharing", "grm")
        outs = {k: np.full((n, n), np.nan, order="F") for k in which}
        check(lib.tpg_multi_pairwise(self.h, _ptr(X_bytes), C.c_int64(X_bytes.shape[0]), C.c_int64(X_bytes.shape[1]),
                                     _ptr(r), C.c_int64(n), _ptr(c), C.c_int64(m), C.c_int(0 if ibs_type == "proportion" else 1),
                                     *[_ptr(outs.get(k)) for k in names]))
        return outs

    @staticmethod
    def _fbm_args(X_bytes, ind_row, ind_col):
        X_bytes = np.asarray(X_bytes)
        assert X_bytes.dtype == np.uint8 and X_bytes.flags.f_contiguous
        r, c = _i32(ind_row), _i32(ind_col)
        n = X_bytes.shape[0] if r is None else len(r)
        m = X_bytes.shape[1] if c is None else len(c)
        args = (_ptr(X_bytes), C.c_int64(X_bytes.shape[0]), C.c_int64(X_bytes.shape[1]), _ptr(r), C.c_int64(n), _ptr(c),
                C.c_int64(m))
        return args, n, m, (X_bytes, r, c)

    def loci_alt_freq(self, X_bytes, ind_row=None, ind_col=None, groupIds=None, ngroups: int = 0, ploidy=None,
                      as_counts: bool = False, code256=CODE_012) -> np.ndarray:
        """loci_alt_freq of a host FBM on all devices: m x 2G (grouped) or m x 2 ({n_alt | freq, n_valid})"""
        args, n, m, _keep = self._fbm_args(X_bytes, ind_row, ind_col)
        gid, code = _i32(groupIds), _f64(code256)
        pl = np.full(n, 2.0) if ploidy is None else _f64(ploidy)
        out = np.zeros((m, 2 * ngroups if gid is not None else 2), order="F")
        check(lib.tpg_multi_grouped_alt_freq(self.h, *args, _ptr(code), _ptr(gid), C.c_int(ngroups), _ptr(pl),
                                             C.c_int(int(as_counts)), _ptr(out)))
        return out

    def pairwise_pop_fst(self, X_bytes, ind_row, ind_col, groupIds, ngroups: int, ploidy=None, method: str = "Hudson",
                         by_locus: bool = False, return_num_dem: bool = False, pairwise_combn=None, code256=CODE_012):
        """pairwise_pop_fst of a host FBM on all devices (same result layout as api.pairwise_pop_fst)"""
        args, n, m, _keep = self._fbm_args(X_bytes, ind_row, ind_col)
        if return_num_dem:
            by_locus = True
        pairs = combn2(ngroups) if pairwise_combn is None else np.asarray(pairwise_combn, dtype=np.int32)
        pairs_c = np.ascontiguousarray(pairs.T)
        P = pairs_c.shape[0]
        gid, code = _i32(groupIds), _f64(code256)
        pl = np.full(n, 2.0) if ploidy is None else _f64(ploidy)
        tot, a, b = _fst_outputs(m, P, by_locus, return_num_dem)
        check(lib.tpg_multi_pop_fst(self.h, *args, _ptr(code), _ptr(gid), C.c_int(ngroups), _ptr(pl),
                                    C.c_int(FST_METHODS[method]), _ptr(pairs_c), C.c_int(P), C.c_int(int(by_locus)),
                                    C.c_int(int(return_num_dem)), _ptr(tot), _ptr(a), _ptr(b)))
        return _fst_result(tot, a, b, by_locus, return_num_dem)

    def gt_pca_partialSVD(self, X_bytes, ind_row=None, ind_col=None, k: int = 10, total_var: bool = True,
                          code256=CODE_IMPUTE_PRED) -> dict:
        """gt_pca_partialSVD of a host FBM on all devices (same result as api.gt_pca_partialSVD)"""
        args, n, m, _keep = self._fbm_args(X_bytes, ind_row, ind_col)
        code = _f64(code256)
        d = np.zeros(k)
        u = np.zeros((n, k), order="F")
        vl = np.zeros((m, k), order="F")
        center, scale = np.zeros(m), np.zeros(m)
        fro = C.c_double()
        check(lib.tpg_multi_pca_partial_svd(self.h, *args, _ptr(code), C.c_int(k), _ptr(d), _ptr(u), _ptr(vl), _ptr(center),
                                            _ptr(scale), C.byref(fro) if total_var else None))
        out = dict(d=d, u=u, v=vl, center=center, scale=scale, method="partialSVD")
        if total_var:
            out["square_frobenius"] = fro.value
        return out

    def close(self):
        if self.h:
            lib.tpg_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """A genotype store that stays on the HOST and is swept in blocks of loci (tpg_stream, include/tpg.h): the
    reference's own block loop (R/snp_ibs.R:59-82, R/loci_alt_freq.R:351-359, big_SVD's two sweeps behind
    R/gt_pca_partialSVD.R:82-89) inside the library.  budget_bytes bounds the HBM taken by store bytes + packed views +
    per-block scratch (0 = no bound)."""

    def __init__(self, ctx: Context, handle, nrow: int, ncol: int, keep=None):
        self.ctx, self.h, self.nrow, self.ncol, self._keep = ctx, handle, nrow, ncol, keep
        self.report = None

    @classmethod
    def from_numpy(cls, bytes_2d, budget_bytes: int = 0, ctx: Optional[Context] = None) -> "Stream":
        """host FBM bytes (uint8, Fortran order: a numpy memmap of the .bk works) -- not copied, kept referenced"""
        ctx = ctx or default_context()
        a = np.asarray(bytes_2d) if not isinstance(bytes_2d, np.memmap) else bytes_2d
        if a.dtype != np.uint8 or a.ndim != 2 or not a.flags.f_contiguous:
            raise TypeError("FBM bytes must be a 2-D uint8 array in Fortran order (individuals x loci)")
        h = C.c_void_p()
        check(lib.tpg_stream_open_host(ctx.h, _ptr(a), C.c_int64(a.shape[0]), C.c_int64(a.shape[1]),
                                       C.c_size_t(int(budget_bytes)), C.byref(h)))
        return cls(ctx, h, a.shape[0], a.shape[1], keep=a)

    @classmethod
    def open_bk(cls, path: str, nrow: int, ncol: int, budget_bytes: int = 0, ctx: Optional[Context] = None) -> "Stream":
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib.tpg_stream_open_bk(ctx.h, path.encode(), C.c_int64(nrow), C.c_int64(ncol), C.c_size_t(int(budget_bytes)),
                                     C.byref(h)))
        return cls(ctx, h, nrow, ncol)

    @classmethod
    def open_bed(cls, path: str, n: int, m: int, budget_bytes: int = 0, ctx: Optional[Context] = None) -> "Stream":
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(lib.tpg_stream_open_bed(ctx.h, path.encode(), C.c_int64(n), C.c_int64(m), C.c_size_t(int(budget_bytes)),
                                      C.byref(h)))
        return cls(ctx, h, n, m)

    @classmethod
    def from_bed_payload(cls, payload, n: int, m: int, budget_bytes: int = 0, ctx: Optional[Context] = None) -> "Stream":
        """the bytes of a PLINK .bed behind its 3-byte magic (uint8, m * ceil(n / 4)) -- not copied, kept referenced"""
        ctx = ctx or default_context()
        a = np.ascontiguousarray(payload, dtype=np.uint8)
        assert a.size == m * ((n + 3) // 4)
        h = C.c_void_p()
        check(lib.tpg_stream_open_bed_host(ctx.h, _ptr(a), C.c_int64(n), C.c_int64(m), C.c_size_t(int(budget_bytes)),
                                           C.byref(h)))
        return cls(ctx, h, n, m, keep=a)

    @classmethod
    def synth(cls, seed: int, nrow: int, ncol: int, npop: int = 51, miss: float = 0.02, imputed_bytes: bool = False,
              budget_bytes: int = 0, ctx: Optional[Context] = None) -> "Stream":
        """the synthetic panel of FBM.synth generated block by block on the device (panels larger than host memory)"""
        ctx = ctx or default_context()
        thr = min(int(round(miss * 2 ** 32)), 2 ** 32 - 1)
        h = C.c_void_p()
        check(lib.tpg_stream_open_synth(ctx.h, C.c_uint64(seed), C.c_int64(nrow), C.c_int64(ncol), C.c_int(npop),
                                        C.c_uint32(thr), C.c_int(int(imputed_bytes)), C.c_size_t(int(budget_bytes)), C.byref(h)))
        return cls(ctx, h, nrow, ncol)

    def run(self, ind_row=None, ind_col=None, pairwise=(), ibs_type: str = "proportion", code256=CODE_012, ploidy=None,
            groupIds=None, ngroups: int = 0, as_counts: bool = False, alt_freq: bool = False, grouped_alt_freq: bool = False,
            grouped_missingness: bool = False, loci_counts: bool = False, fst=(), fst_by_locus: bool = False,
            fst_return_num_dem: bool = False, pairwise_combn=None, k: int = 0, pca_tol: float = 0.0, code256_pca=CODE_IMPUTE_PRED, total_var: bool = True,
            multi: Optional["Multi"] = None) -> dict:
        """One streamed pass for everything asked for (tpg_stream_run; with `multi`, tpg_multi_stream_run: the loci
        sharded over its devices).  pairwise: any of "ibs", "king", "allele_sharing", "grm"; fst: up to three of
        "Hudson", "Nei87", "WC84"; k > 0: gt_pca_partialSVD (pca_tol > 0: gt_pca_randomSVD's tolerance).  Returns the
        results under the names of the resident functions, plus "report" (blocks, bytes moved, peak HBM)."""
        r, c = _i32(ind_row), _i32(ind_col)
        n = self.nrow if r is None else len(r)
        m = self.ncol if c is None else len(c)
        job = _lib.StreamJob()
        job.struct_size = C.sizeof(_lib.StreamJob)
        keep = [r, c]
        job.rowInd1, job.n, job.colInd1, job.m = _ptr(r), n, _ptr(c), m
        out = {}
        job.ibs_type = 0 if ibs_type == "proportion" else 1
        for name in pairwise:
            if name not in ("ibs", "king", "allele_sharing", "grm"):
                raise ValueError(f"unknown pairwise output {name!r}")
            out[name] = np.empty((n, n), order="F")  # (every element is written: rows and their mirror images)
            setattr(job, name, _ptr(out[name]))
        code = _f64(code256)
        pl = None if ploidy is None else _f64(ploidy)
        gid = _i32(groupIds)
        keep += [code, pl, gid]
        job.code256, job.ploidy, job.groupIds0, job.ngroups, job.as_counts = _ptr(code), _ptr(pl), _ptr(gid), int(ngroups), int(as_counts)
        if alt_freq:
            out["alt_freq"] = np.empty((m, 2), order="F")
            job.alt_freq = _ptr(out["alt_freq"])
        if grouped_alt_freq:
            out["grouped_alt_freq"] = np.empty((m, 2 * ngroups), order="F")
            job.grouped_alt_freq = _ptr(out["grouped_alt_freq"])
        if grouped_missingness:
            out["grouped_missingness"] = np.empty((m, ngroups), order="F")
            job.grouped_missingness = _ptr(out["grouped_missingness"])
        if loci_counts:
            out["loci_counts"] = np.empty((m, 4), dtype=np.int32)
            job.loci_counts = _ptr(out["loci_counts"])
        if fst:
            pairs = combn2(ngroups) if pairwise_combn is None else np.asarray(pairwise_combn, dtype=np.int32)
            pairs_c = np.ascontiguousarray(pairs.T)
            keep.append(pairs_c)
            P = pairs_c.shape[0]
            job.nfst, job.pairs1, job.P = len(fst), _ptr(pairs_c), P
            if fst_return_num_dem:
                fst_by_locus = True  # R/pairwise_pop_fst.R:103-106
            job.fst_return_num_dem = int(fst_return_num_dem)
            out["fst_tot"], out["fst_locus"], out["fst_locus_den"] = {}, {}, {}
            for i, method in enumerate(fst):
                job.fst_method[i] = FST_METHODS[method]
                out["fst_tot"][method] = np.zeros(P)
                job.fst_tot[i] = out["fst_tot"][method].ctypes.data
                if fst_by_locus:
                    out["fst_locus"][method] = np.empty((m, P), order="F")  # (the numerators under fst_return_num_dem)
                    job.fst_by_locus[i] = out["fst_locus"][method].ctypes.data
                if fst_return_num_dem:
                    out["fst_locus_den"][method] = np.empty((m, P), order="F")
                    job.fst_by_locus_den[i] = out["fst_locus_den"][method].ctypes.data
            if not fst_by_locus:
                del out["fst_locus"]
            if not fst_return_num_dem:
                del out["fst_locus_den"]
        fro = C.c_double()
        if k > 0:
            cp = _f64(code256_pca)
            keep.append(cp)
            out.update(d=np.zeros(k), u=np.empty((n, k), order="F"), v=np.empty((m, k), order="F"), center=np.empty(m),
                       scale=np.empty(m), method="partialSVD" if pca_tol == 0 else "randomSVD")
            job.code256_pca, job.k, job.pca_tol = _ptr(cp), int(k), float(pca_tol)
            for name in ("d", "u", "v", "center", "scale"):
                setattr(job, name, _ptr(out[name]))
            if total_var:
                job.square_frobenius = C.cast(C.pointer(fro), C.c_void_p)
        rep = _lib.StreamReport()
        if multi is not None:
            check(lib.tpg_multi_stream_run(multi.h, self.h, C.byref(job), C.byref(rep)))
        else:
            check(lib.tpg_stream_run(self.ctx.h, self.h, C.byref(job), C.byref(rep)))
        if k > 0 and total_var:
            out["square_frobenius"] = fro.value
        self.report = {f: getattr(rep, f) for f, _ in _lib.StreamReport._fields_}
        out["report"] = self.report
        return out

    def close(self):
        if self.h:
            lib.tpg_stream_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# R-level functions

def _raw_view(X: FBM, ind_row, ind_col) -> View:
    # increment_{ibs,king,as}_counts compare the RAW bytes with 0/1/2 (src/snp_ibs.cpp:47-54)
    return View(X, ind_row, ind_col, code256=None)


def _pairwise_pass(X: FBM, ind_row, ind_col, products=None):
    v = _raw_view(X, ind_row, ind_col)
    pw = Pairwise(X.ctx, v.n)
    pw.accumulate(v, products=products)
    return v, pw


def snp_ibs(X: FBM, ind_row=None, ind_col=None, type: str = "proportion", block_size=None):
    """R/snp_ibs.R:42-104.  block_size is accepted for signature compatibility; the whole locus
    range is swept in one device pass (results do not depend on it)."""
    if type not in ("proportion", "adjusted_counts", "raw_counts"):
        raise ValueError("'arg' should be one of 'proportion', 'adjusted_counts', 'raw_counts'")
    v, pw = _pairwise_pass(X, ind_row, ind_col, PW_FOR_IBS_ALONE)  # V and D + H: three MFMAs into two sums per tile pair
    if type == "raw_counts":
        c = pw.counts(("ibs", "ibs_valid"))
        return dict(ibs=c["ibs"], valid_n=c["ibs_valid"])
    return pw.ibs(type, v.m)


def snp_king(X: FBM, ind_row=None, ind_col=None, block_size=None):
    """R/snp_king.R:32-103"""
    _, pw = _pairwise_pass(X, ind_row, ind_col, PW_FOR_KING)  # V, D, A, A': 4 of 5
    return pw.king()


def as_pad_quirk_blocks(m: int, block_size: int) -> int:
    """blocks narrower than the widest when CutBySize(m, block_size) cuts m loci (R/local_reimplementations.R:13-15)"""
    return int(lib.tpg_as_pad_quirk_blocks(C.c_int64(int(m)), C.c_int64(int(block_size))))


def _as_pass(X, ind_row, ind_col, block_size, emulate_as_pad_quirk):
    v, pw = _pairwise_pass(X, ind_row, ind_col, PW_FOR_AS)  # V, D: 2 of 5
    if emulate_as_pad_quirk:
        # what the reference BINARY returns: +1 on every numerator per narrower block (src/snp_as.cpp:57-63)
        pw.set_as_pad_quirk(as_pad_quirk_blocks(v.m, block_size or block_size_default(X.nrow)))
    return pw


def snp_allele_sharing(X: FBM, ind_row=None, ind_col=None, block_size=None, emulate_as_pad_quirk: bool = False):
    """R/snp_allele_sharing.R:33-82.  Default: the mathematically intended value (what the reference's test asserts
    through hierfstat::matching).  emulate_as_pad_quirk = True reproduces what the reference binary computes when
    its blocks are unequal (quirk Q1 of SURVEY.md 8a) for the given block_size (default bigstatsr::block_size)."""
    return _as_pass(X, ind_row, ind_col, block_size, emulate_as_pad_quirk).allele_sharing()


def pairwise_grm(X: FBM, ind_row=None, ind_col=None, block_size=None, emulate_as_pad_quirk: bool = False):
    """R/pairwise_grm.R:30-51 on top of snp_allele_sharing"""
    return _as_pass(X, ind_row, ind_col, block_size, emulate_as_pad_quirk).grm()


def block_means(A, groupIds, ngroups: int, skip_diag: bool = True, ctx: Optional[Context] = None):
    """mean(A[p1, p2], na.rm = TRUE) for every pair of groups (R/pop_fst.R:47-62) -> (G, G) means, (G, G) counts"""
    ctx = ctx or default_context()
    A = np.asfortranarray(A, dtype=np.float64)
    gid = _i32(groupIds)
    mean, cnt = np.zeros((ngroups, ngroups), order="F"), np.zeros((ngroups, ngroups), order="F")
    check(lib.tpg_block_means(ctx.h, _ptr(A), C.c_int64(A.shape[0]), _ptr(gid), C.c_int(ngroups),
                              C.c_int(int(skip_diag)), _ptr(mean), _ptr(cnt)))
    return mean, cnt


def _as_block_stats(X, ind_row, ind_col, groupIds, ngroups, allele_sharing_mat):
    if ngroups < 1:
        raise ValueError(".x should be a grouped gen_tibble")
    if allele_sharing_mat is None:
        allele_sharing_mat = snp_allele_sharing(X, ind_row, ind_col)
    mMij, _ = block_means(allele_sharing_mat, groupIds, ngroups, skip_diag=True, ctx=X.ctx if X is not None else None)
    Fsts = np.diag(mMij).copy()
    # Mb: sum of the strictly lower triangle in the reference's loop order (i = 2..n_pop, j = 1..i-1), :53-63
    Mb = 0.0
    for i in range(1, ngroups):
        for j in range(i):
            Mb = Mb + mMij[i, j]
    with np.errstate(invalid="ignore", divide="ignore"):
        Mb = Mb * 2 / (ngroups * (ngroups - 1))
    return allele_sharing_mat, Fsts, Mb


def pop_fst(X: FBM, ind_row, ind_col, groupIds, ngroups: int, include_global: bool = False, allele_sharing_mat=None):
    """R/pop_fst.R:31-76 (Weir & Goudet 2017 population-specific Fst from the allele-sharing matrix)"""
    _, Fsts, Mb = _as_block_stats(X, ind_row, ind_col, groupIds, ngroups, allele_sharing_mat)
    with np.errstate(invalid="ignore", divide="ignore"):
        fst = (Fsts - Mb) / (1 - Mb)
        if include_global:
            fst = np.append(fst, np.nanmean(fst) if np.any(~np.isnan(fst)) else np.nan)
    return fst


def pop_fis_wg17(X: FBM, ind_row, ind_col, groupIds, ngroups: int, include_global: bool = False, allele_sharing_mat=None):
    """R/pop_fis.R:136-197 (method = "WG17")"""
    A, Fsts, _ = _as_block_stats(X, ind_row, ind_col, groupIds, ngroups, allele_sharing_mat)
    Mii = np.diag(np.asarray(A)) * 2 - 1
    gid = np.asarray(groupIds)
    out = np.full(ngroups, np.nan)
    with np.errstate(invalid="ignore", divide="ignore"):
        for g in range(ngroups):
            Fi = (Mii[gid == g] - Fsts[g]) / (1 - Fsts[g])
            if np.any(~np.isnan(Fi)):
                out[g] = np.nanmean(Fi)
        if include_global:
            out = np.append(out, np.nanmean(out) if np.any(~np.isnan(out)) else np.nan)
    return out


def filter_high_relatedness(matrix, kings_threshold, ids=None, ctx: Optional[Context] = None):
    """R/filter_high_relatedness.R:26-145 -> [ids that pass (in the order of decreasing mean relatedness), ids to
    remove, logical keep vector in the original order].  `matrix`: (n, n) numpy array, or a raw device pointer together
    with ids (its length gives n) -- e.g. the KING matrix left in HBM by Pairwise.king into a dev_alloc'ed buffer.
    ids default to "1" .. "n", as the reference names an unnamed matrix."""
    ctx = ctx or default_context()
    if kings_threshold is None:
        raise ValueError("argument \"kings_threshold\" is missing")
    if isinstance(matrix, (int, np.integer, C.c_void_p)):
        if ids is None:
            raise ValueError("ids are needed with a device matrix")
        n, mp = len(ids), (matrix if isinstance(matrix, C.c_void_p) else C.c_void_p(int(matrix)))
    else:
        A = np.asfortranarray(matrix, dtype=np.float64)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("matrix should be a square matrix")
        n, mp = A.shape[0], _ptr(A)
    ids = np.array([str(k) for k in range(1, n + 1)]) if ids is None else np.asarray(ids)
    keep, order = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.int32)
    check(lib.tpg_filter_high_relatedness(ctx.h, mp, C.c_int64(n), C.c_double(float(kings_threshold)), _ptr(keep),
                                          _ptr(order)))
    keepb = keep.astype(bool)
    passed = ids[order][keepb[order]]
    return [passed, ids[~keepb], keepb]


def increment_ibs_counts(k, k2, X_bytes, rowInd, colInd, ctx: Optional[Context] = None, flush: bool = True):
    """Literal mirror of src/snp_ibs.cpp:22-74 (X_bytes is the host FBM; every call uploads the columns of its own
    block, nothing of the FBM is kept).  flush = True (default): k, k2 are incremented when the call returns, as in the
    reference.  flush = False (tpg_increment_defer): the sums stay in device accumulators across the calls of a block loop
    and reach k, k2 at increment_flush()."""
    return _increment(lib.tpg_increment_ibs_counts, k, k2, X_bytes, rowInd, colInd, ctx, flush)


def increment_king_numerator(k, n_Aa_i, X_bytes, rowInd, colInd, ctx: Optional[Context] = None, flush: bool = True):
    """Literal mirror of src/snp_king.cpp:21-74"""
    return _increment(lib.tpg_increment_king_numerator, k, n_Aa_i, X_bytes, rowInd, colInd, ctx, flush)


def increment_as_counts(k, k2, X_bytes, rowInd, colInd, ctx: Optional[Context] = None, flush: bool = True,
                        scratch_cols: Optional[int] = None, emulate_as_pad_quirk: bool = False):
    """Literal mirror of src/snp_as.cpp:22-67.  scratch_cols = number of columns of the scratch matrices the R driver
    passes; with emulate_as_pad_quirk a block one column narrower than that adds +1 to every numerator (quirk Q1)."""
    ctx = ctx or default_context()
    _increment(lib.tpg_increment_as_counts, k, k2, X_bytes, rowInd, colInd, ctx, flush)
    if emulate_as_pad_quirk and scratch_cols is not None and scratch_cols == len(colInd) + 1:
        check(lib.tpg_increment_as_note_narrow_block(ctx.h, _ptr(k), C.c_int64(k.shape[0])))


def increment_flush(ctx: Optional[Context] = None):
    """add the sums the increment_* mirrors hold on the device to their host matrices"""
    ctx = ctx or default_context()
    check(lib.tpg_increment_flush(ctx.h))


def resident_drop(ctx: Optional[Context] = None):
    """release the device scratch the increment_* mirrors keep between calls"""
    ctx = ctx or default_context()
    check(lib.tpg_resident_drop(ctx.h))


def _increment(fn, a, b, X_bytes, rowInd, colInd, ctx, flush):
    ctx = ctx or default_context()
    X_bytes = np.asarray(X_bytes)
    assert X_bytes.dtype == np.uint8 and X_bytes.flags.f_contiguous
    assert a.flags.f_contiguous and b.flags.f_contiguous and a.dtype == np.float64 and b.dtype == np.float64
    r, c = _i32(rowInd), _i32(colInd)
    check(lib.tpg_increment_defer(ctx.h, C.c_int(int(not flush))))  # switching it off flushes what is pending
    check(fn(ctx.h, _ptr(a), _ptr(b), _ptr(X_bytes), C.c_int64(X_bytes.shape[0]), C.c_int64(X_bytes.shape[1]),
             _ptr(r), C.c_int64(len(r)), _ptr(c), C.c_int64(len(c))))


def _ploidy(v: View, ploidy):
    return np.full(v.n, 2.0) if ploidy is None else _f64(ploidy)


def loci_counts(v: View) -> np.ndarray:
    out = np.zeros((v.m, 4), dtype=np.int32)
    check(lib.tpg_loci_counts(v.ctx.h, v.h, _ptr(out)))
    return out


def indiv_counts(v: View) -> np.ndarray:
    out = np.zeros((v.n, 4), dtype=np.int32)
    check(lib.tpg_indiv_counts(v.ctx.h, v.h, _ptr(out)))
    return out


def gt_ind_hetero(v: View) -> np.ndarray:
    """src/gt_ind_hetero.cpp:11-42 -> (2, n) integer matrix: row 0 heterozygous loci, row 1 missing loci"""
    out = np.zeros((2, v.n), dtype=np.int32, order="F")
    check(lib.tpg_gt_ind_hetero(v.ctx.h, v.h, _ptr(out)))
    return out


def gt_pi_diploid(v: View) -> np.ndarray:
    """src/gt_pi_diploid.cpp:7-38"""
    out = np.zeros(v.m)
    check(lib.tpg_gt_pi_diploid(v.ctx.h, v.h, _ptr(out)))
    return out


def gt_grouped_pi_diploid(v: View, groupIds, ngroups: int) -> dict:
    """src/gt_grouped_pi_diploid.cpp:7-42"""
    pi, n = np.zeros((v.m, ngroups), order="F"), np.zeros((v.m, ngroups), order="F")
    gid = _i32(groupIds)
    check(lib.tpg_gt_grouped_pi_diploid(v.ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(pi), _ptr(n)))
    return dict(pi=pi, n=n)


def grouped_genotype_counts(v: View, groupIds, ngroups: int) -> np.ndarray:
    """the genotype table of gt_grouped_hwe (src/hwe.cpp:238-250) -> (3, m, G) int32: [k] = individuals with k alternate alleles"""
    out = np.zeros((3, ngroups, v.m), dtype=np.int32)  # three column-major m x G matrices
    gid = _i32(groupIds)
    check(lib.tpg_grouped_genotype_counts(v.ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(out)))
    return np.ascontiguousarray(out.transpose(0, 2, 1))


GLOBAL_STATS_COLUMNS = ("Ho", "Hs", "Ht", "Dst", "Htp", "Dstp", "Fst", "Fstp", "Fis", "Dest")


def pop_global_stats(X: FBM, ind_row, ind_col, groupIds, ngroups: int, ploidy=None, by_locus: bool = False):
    """R/pop_global_stats.R:113-212 -> (m, 10) array (by_locus) or the 10 overall values; columns GLOBAL_STATS_COLUMNS"""
    v = View(X, ind_row, ind_col)
    gid = _i32(groupIds)
    pl = _ploidy(v, ploidy)
    loc = np.zeros((v.m, 10), order="F") if by_locus else None
    ov = np.zeros(10)
    check(lib.tpg_pop_global_stats(v.ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(pl),
                                   _ptr(loc) if by_locus else None, _ptr(ov)))
    return loc if by_locus else ov


def _pop_basic(X, ind_row, ind_col, groupIds, ngroups, ploidy, which, by_locus, include_global, global_col):
    v = View(X, ind_row, ind_col)
    gid, pl = _i32(groupIds), _ploidy(v, ploidy)
    loc = np.zeros((v.m, ngroups), order="F") if by_locus else None
    cm = np.zeros(ngroups)
    check(lib.tpg_pop_basic_stats(v.ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(pl), C.c_int(which),
                                  _ptr(loc) if by_locus else None, _ptr(cm)))
    if not include_global:
        return loc if by_locus else cm
    g_loc = pop_global_stats(X, ind_row, ind_col, groupIds, ngroups, ploidy, by_locus=True)[:, global_col]
    if by_locus:
        return np.column_stack([loc, g_loc])
    with np.errstate(invalid="ignore"):
        return np.append(cm, np.nanmean(g_loc) if np.any(~np.isnan(g_loc)) else np.nan)


def pop_het_obs(X: FBM, ind_row, ind_col, groupIds, ngroups: int, ploidy=None, by_locus=False, include_global=False):
    """R/pop_het_obs.R:52-92"""
    return _pop_basic(X, ind_row, ind_col, groupIds, ngroups, ploidy, 0, by_locus, include_global, 0)


def pop_het_exp(X: FBM, ind_row, ind_col, groupIds, ngroups: int, ploidy=None, by_locus=False, include_global=False):
    """R/pop_het_exp.R:53-104 (alias pop_gene_div)"""
    return _pop_basic(X, ind_row, ind_col, groupIds, ngroups, ploidy, 1, by_locus, include_global, 1)


pop_gene_div = pop_het_exp


def pop_fis(X: FBM, ind_row, ind_col, groupIds, ngroups: int, ploidy=None, method: str = "Nei87", by_locus=False,
            include_global=False, allele_sharing_mat=None):
    """R/pop_fis.R:56-133"""
    if method not in ("Nei87", "WG17"):
        raise ValueError("'arg' should be one of 'Nei87', 'WG17'")
    if method == "WG17":
        if by_locus:
            raise ValueError("by_locus not implemented for WG17")
        return pop_fis_wg17(X, ind_row, ind_col, groupIds, ngroups, include_global, allele_sharing_mat)
    if allele_sharing_mat is not None:
        raise ValueError("allele_sharing_mat not relevant for Nei87")
    if by_locus or not include_global:
        return _pop_basic(X, ind_row, ind_col, groupIds, ngroups, ploidy, 2, by_locus, include_global, 8)
    # by_locus = FALSE with the global value: the reference takes pop_global_stats(by_locus = FALSE)["Fis"], :126-130
    cm = _pop_basic(X, ind_row, ind_col, groupIds, ngroups, ploidy, 2, False, False, 8)
    return np.append(cm, pop_global_stats(X, ind_row, ind_col, groupIds, ngroups, ploidy, by_locus=False)[8])


def alt_freq_dip_pseudo_cpp(v: View, ploidy=None, as_counts: bool = False) -> np.ndarray:
    """src/alt_freq_dip_pseudo_cpp.cpp:8-58 -> (m, 2)"""
    out = np.zeros((v.m, 2), order="F")
    pl = _ploidy(v, ploidy)
    check(lib.tpg_alt_freq_dip_pseudo(v.ctx.h, v.h, _ptr(pl), C.c_int(int(as_counts)), _ptr(out)))
    return out


def loci_alt_freq(X: FBM, ind_row=None, ind_col=None, ploidy=None, as_counts: bool = False, block_size=None):
    """R/loci_alt_freq.R:328-379"""
    v = View(X, ind_row, ind_col)
    freq = alt_freq_dip_pseudo_cpp(v, ploidy, as_counts)
    return freq if as_counts else freq[:, 0]


def loci_missingness(X: FBM, ind_row=None, ind_col=None, as_counts: bool = False, block_size=None):
    """R/loci_missingness.R:97-134"""
    v = View(X, ind_row, ind_col)
    n_na = loci_counts(v)[:, 3].astype(float)
    return n_na if as_counts else n_na / v.n


def grouped_alt_freq_dip_pseudo_cpp(v: View, groupIds, ngroups: int, ploidy=None, as_counts: bool = False):
    """src/grouped_alt_freq_dip_pseudo_cpp.cpp:8-58 -> (m, 2G)"""
    out = np.zeros((v.m, 2 * ngroups), order="F")
    gid, pl = _i32(groupIds), _ploidy(v, ploidy)
    check(lib.tpg_grouped_alt_freq_dip_pseudo(v.ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(pl),
                                              C.c_int(int(as_counts)), _ptr(out)))
    return out


def grouped_missingness_cpp(v: View, groupIds, ngroups: int):
    """src/grouped_missingness_cpp.cpp:8-33 -> (m, G)"""
    out = np.zeros((v.m, ngroups), order="F")
    gid = _i32(groupIds)
    check(lib.tpg_grouped_missingness(v.ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(out)))
    return out


def grouped_summaries_dip_pseudo_cpp(v: View, groupIds, ngroups: int, ploidy=None) -> dict:
    """src/grouped_summaries_dip_pseudo_cpp.cpp:11-63"""
    outs = [np.zeros((v.m, ngroups), order="F") for _ in range(4)]
    gid, pl = _i32(groupIds), _ploidy(v, ploidy)
    check(lib.tpg_grouped_summaries_dip_pseudo(v.ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(pl),
                                               *[_ptr(o) for o in outs]))
    return dict(freq_alt=outs[0], freq_ref=outs[1], n=outs[2], het_obs=outs[3])


def combn2(G: int) -> np.ndarray:
    """utils::combn(G, 2) (R/pairwise_pop_fst.R:119): 2 x P, 1-based"""
    cols = [(a, b) for a in range(1, G + 1) for b in range(a + 1, G + 1)]
    return np.array(cols, dtype=np.int32).T.reshape(2, -1)


def _fst_outputs(m, P, by_locus, return_num_dem):
    tot = np.zeros(P)
    a = np.zeros((m, P), order="F") if by_locus else None
    b = np.zeros((m, P), order="F") if return_num_dem else None
    return tot, a, b


def _fst_result(tot, a, b, by_locus, return_num_dem):
    if return_num_dem:
        return dict(Fst_by_locus_num=a, Fst_by_locus_den=b)
    return dict(fst_locus=a if by_locus else np.zeros((0, 0)), fst_tot=tot)


def pairwise_pop_fst(X: FBM, ind_row, ind_col, groupIds, ngroups: int, ploidy=None, method: str = "Hudson",
                     by_locus: bool = False, return_num_dem: bool = False, pairwise_combn=None, sums: bool = False):
    """R/pairwise_pop_fst.R:71-161 (numeric part; the tidy / matrix formatting is out of scope).  sums = True
    (not in the reference): also return the sums of numerators and denominators over the loci, the additive
    quantities a run sharded by loci exchanges."""
    if method not in FST_METHODS:
        raise ValueError("'arg' should be one of 'Hudson', 'Nei87', 'WC84'")
    if not isinstance(return_num_dem, (bool, np.bool_)):
        raise ValueError("return_num_dem must be a logical value (TRUE or FALSE)")
    if return_num_dem:
        by_locus = True
    v = View(X, ind_row, ind_col)
    pairs = combn2(ngroups) if pairwise_combn is None else np.asarray(pairwise_combn, dtype=np.int32)
    pairs_c = np.ascontiguousarray(pairs.T)  # (P, 2) row-major == 2 x P column-major
    P = pairs_c.shape[0]
    gid, pl = _i32(groupIds), _ploidy(v, ploidy)
    if sums:
        if by_locus:
            raise ValueError("sums = True returns totals only")
        sn, sd = np.zeros(P), np.zeros(P)
        check(lib.tpg_pairwise_pop_fst_sums(v.ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(pl),
                                            C.c_int(FST_METHODS[method]), _ptr(pairs_c), C.c_int(P), _ptr(sn), _ptr(sd)))
        with np.errstate(invalid="ignore", divide="ignore"):
            return dict(fst_tot=sn / sd, sum_num=sn, sum_den=sd)
    tot, a, b = _fst_outputs(v.m, P, by_locus, return_num_dem)
    check(lib.tpg_pairwise_pop_fst(v.ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(pl),
                                   C.c_int(FST_METHODS[method]), _ptr(pairs_c), C.c_int(P), C.c_int(int(by_locus)),
                                   C.c_int(int(return_num_dem)), _ptr(tot), _ptr(a), _ptr(b)))
    return _fst_result(tot, a, b, by_locus, return_num_dem)


def window_index_ranges(chromosome, position, window_size, step_size, size_unit="snp", complete=False):
    """Host part of windows_stats_generic (R/windows_stats_generic.R:113-141 and runner's window rule, recalled:
    the window ending at `at` holds the indices in (at - k, at]; with na_pad = TRUE a window that reaches outside
    the index range is NA).  Loci must be ordered by position inside a chromosome, chromosomes in blocks, as in a
    gen_tibble.  -> dict(chromosome, start, end, lo, hi, pad_na) with lo/hi 0-based half-open locus ranges."""
    if size_unit not in ("snp", "bp"):
        raise ValueError("'arg' should be one of 'snp', 'bp'")
    if not isinstance(complete, (bool, np.bool_)):
        raise ValueError("complete must be a boolean (logical).")
    if window_size <= 0:
        raise ValueError("window_size must be positive.")
    if step_size <= 0:
        raise ValueError("step_size must be positive.")
    chromosome = np.asarray(chromosome)
    if size_unit == "bp":
        if position is None:
            raise ValueError("loci_table must contain columns 'chromosome' and 'position' when size_unit is 'bp'.")
        position = np.asarray(position, dtype=np.float64)
    chroms, starts_, ends_, lo, hi, pad = [], [], [], [], [], []
    seen = []
    for ch in chromosome:  # unique(), order of first appearance
        if ch not in seen:
            seen.append(ch)
    for ch in seen:
        idx = np.where(chromosome == ch)[0]
        first = int(idx[0])
        if not np.array_equal(idx, np.arange(first, first + len(idx))):
            raise ValueError("the loci of a chromosome must be contiguous")
        pos = position[idx] if size_unit == "bp" else np.arange(1, len(idx) + 1, dtype=np.float64)
        if np.any(np.diff(pos) < 0):
            raise ValueError("positions must be sorted inside a chromosome")
        r0, r1 = math.ceil(pos.min() / window_size), math.ceil(pos.max() / window_size)
        at = np.arange(r0 * window_size, r1 * window_size + 1e-9 * step_size, step_size, dtype=np.float64)
        for a in at:
            w_lo = int(np.searchsorted(pos, a - window_size, side="right"))  # first index with pos > at - k
            w_hi = int(np.searchsorted(pos, a, side="right"))                # one past the last with pos <= at
            chroms.append(ch); starts_.append(a - window_size + 1); ends_.append(a)
            lo.append(first + w_lo); hi.append(first + w_hi)
            pad.append(bool(complete) and (a - window_size + 1 < pos[0] or a > pos[-1]))
    return dict(chromosome=np.array(chroms), start=np.array(starts_), end=np.array(ends_),
                lo=np.array(lo, dtype=np.int64), hi=np.array(hi, dtype=np.int64), pad_na=np.array(pad, dtype=np.uint8))


def _window_stats(ctx, x_ptr, m, ncol, wr, op, min_loci):
    nw = len(wr["lo"])
    stat = np.zeros((nw, ncol), order="F")
    nl = np.zeros((nw, ncol), dtype=np.int32, order="F")
    lo, hi, pad = wr["lo"], wr["hi"], wr["pad_na"]
    check(lib.tpg_window_stats(ctx.h, x_ptr, C.c_int64(m), C.c_int(ncol), _ptr(lo), _ptr(hi), _ptr(pad), C.c_int64(nw),
                               C.c_int(op), C.c_int(int(min_loci)), _ptr(stat), _ptr(nl)))
    return stat, nl


def windows_stats_generic(x, chromosome, position=None, operator="mean", window_size=None, step_size=None,
                          size_unit="snp", min_loci=1, complete=False, ctx: Optional[Context] = None):
    """R/windows_stats_generic.R:47-184 for operator "mean" / "sum" -> dict(chromosome, start, end, stat, n_loci)"""
    if operator not in ("mean", "sum"):
        raise ValueError("'arg' should be one of 'mean', 'sum' (custom functions run on the host in the reference)")
    x = np.ascontiguousarray(x, dtype=np.float64)
    if len(chromosome) != len(x):
        raise ValueError("loci_table must have the same number of rows as x.")
    if min_loci is not None and min_loci <= 0:
        raise ValueError("min_loci must be positive.")
    wr = window_index_ranges(chromosome, position, window_size, step_size, size_unit, complete)
    if min_loci > window_size:
        raise ValueError("min_loci must be less than window_size.")
    ctx = ctx or default_context()
    stat, nl = _window_stats(ctx, _ptr(x), len(x), 1, wr, 0 if operator == "mean" else 1, min_loci)
    n_loci = nl[:, 0].astype(float)
    n_loci[nl[:, 0] < 0] = np.nan
    return dict(chromosome=wr["chromosome"], start=wr["start"], end=wr["end"], stat=stat[:, 0], n_loci=n_loci)


def windows_pairwise_pop_fst(X: FBM, ind_row, ind_col, groupIds, ngroups: int, chromosome, position=None, ploidy=None,
                             window_size=None, step_size=None, size_unit="snp", min_loci=1, complete=False):
    """R/windows_pairwise_pop_fst.R:49-118 (type = "matrix"): window means of the by-locus Hudson numerators and
    denominators, then their ratio (the reference always takes Hudson here, whatever `method` says, :62-65).  The two
    m x P matrices stay in HBM.  -> dict(chromosome, start, end, fst (nw, P))"""
    v = View(X, ind_row, ind_col)
    if len(chromosome) != v.m:
        raise ValueError("loci_table must have the same number of rows as x.")
    wr = window_index_ranges(chromosome, position, window_size, step_size, size_unit, complete)
    if min_loci <= 0:
        raise ValueError("min_loci must be positive.")
    if min_loci > window_size:
        raise ValueError("min_loci must be less than window_size.")
    pairs_c = np.ascontiguousarray(combn2(ngroups).T)
    P = pairs_c.shape[0]
    gid, pl = _i32(groupIds), _ploidy(v, ploidy)
    ctx = v.ctx
    nbytes = 8 * v.m * P
    d_num, d_den = ctx.dev_alloc(nbytes), ctx.dev_alloc(nbytes)
    try:
        check(lib.tpg_pairwise_pop_fst(ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(pl), C.c_int(FST_METHODS["Hudson"]),
                                       _ptr(pairs_c), C.c_int(P), C.c_int(1), C.c_int(1), None, d_num, d_den))
        num, _ = _window_stats(ctx, d_num, v.m, P, wr, 0, min_loci)
        den, _ = _window_stats(ctx, d_den, v.m, P, wr, 0, min_loci)
    finally:
        ctx.dev_free(d_num)
        ctx.dev_free(d_den)
    with np.errstate(invalid="ignore", divide="ignore"):
        fst = num / den
    return dict(chromosome=wr["chromosome"], start=wr["start"], end=wr["end"], fst=fst)


def _pbs_triplets(ngroups):
    """utils::combn(levels, 3) order, with the Fst columns of (p1.p2, p1.p3, p2.p3) in combn(levels, 2) order"""
    pairs = combn2(ngroups)  # (2, P), 1-based
    col = {(int(a), int(b)): k for k, (a, b) in enumerate(pairs.T)}
    trips, cols = [], []
    for a in range(1, ngroups + 1):
        for b in range(a + 1, ngroups + 1):
            for c in range(b + 1, ngroups + 1):
                trips.append((a, b, c))
                cols.append((col[(a, b)], col[(a, c)], col[(b, c)]))
    return trips, np.ascontiguousarray(cols, dtype=np.int32)


def nwise_pop_pbs(X: FBM, ind_row, ind_col, groupIds, ngroups: int, ploidy=None, fst_method: str = "Hudson",
                  return_fst: bool = False):
    """R/nwise_pop_pbs.R:36-156, type = "matrix": by-locus PBS and normalised PBS for every triplet of populations.
    -> dict(pbs (m, 6 * n_triplets), names, [fst (m, P)]); the by-locus Fst matrix stays in HBM in between."""
    if ngroups < 3:
        raise ValueError("At least 3 populations are required to compute PBS.")
    if not isinstance(return_fst, (bool, np.bool_)):
        raise ValueError("return_fst must be a logical value (TRUE or FALSE)")
    if fst_method not in FST_METHODS:
        raise ValueError("'arg' should be one of 'Hudson', 'Nei87', 'WC84'")
    v = View(X, ind_row, ind_col)
    pairs_c = np.ascontiguousarray(combn2(ngroups).T)
    P = pairs_c.shape[0]
    trips, tcols = _pbs_triplets(ngroups)
    gid, pl = _i32(groupIds), _ploidy(v, ploidy)
    ctx = v.ctx
    tot = np.zeros(P)
    d_fst = ctx.dev_alloc(8 * v.m * P)
    try:
        check(lib.tpg_pairwise_pop_fst(ctx.h, v.h, _ptr(gid), C.c_int(ngroups), _ptr(pl), C.c_int(FST_METHODS[fst_method]),
                                       _ptr(pairs_c), C.c_int(P), C.c_int(1), C.c_int(0), _ptr(tot), d_fst, None))
        out = np.zeros((v.m, 6 * len(trips)), order="F")
        check(lib.tpg_pbs_from_fst(ctx.h, d_fst, C.c_int64(v.m), C.c_int(P), _ptr(tcols), C.c_int(len(trips)), _ptr(out)))
        res = dict(pbs=out)
        if return_fst:
            f = np.zeros((v.m, P), order="F")
            check(lib.tpg_dev_to_host(ctx.h, _ptr(f), d_fst, C.c_size_t(f.nbytes)))
            res["fst"] = f
    finally:
        ctx.dev_free(d_fst)
    names = []
    for (a, b, c) in trips:
        for stat in ("pbs", "pbsn1"):
            names += [f"{stat}_{a}.{b}.{c}", f"{stat}_{b}.{a}.{c}", f"{stat}_{c}.{a}.{b}"]
    res["names"] = names
    return res


def _fst_loop(method, pairwise_combn, n, freq_alt, freq_ref, het_obs, by_locus, return_num_dem, ctx):
    ctx = ctx or default_context()
    pairs_c = np.ascontiguousarray(np.asarray(pairwise_combn, dtype=np.int32).T)
    P = pairs_c.shape[0]
    n = np.asfortranarray(n, dtype=float)
    m, G = n.shape
    mats = [None if x is None else np.asfortranarray(x, dtype=float) for x in (freq_alt, freq_ref, het_obs)]
    tot, a, b = _fst_outputs(m, P, by_locus or return_num_dem, return_num_dem)
    check(lib.tpg_pairwise_fst_loop(ctx.h, C.c_int(FST_METHODS[method]), _ptr(pairs_c), C.c_int(P), C.c_int64(m),
                                    C.c_int(G), _ptr(n), *[_ptr(x) for x in mats], C.c_int(int(by_locus)),
                                    C.c_int(int(return_num_dem)), _ptr(tot), _ptr(a), _ptr(b)))
    return _fst_result(tot, a, b, by_locus or return_num_dem, return_num_dem)


def pairwise_fst_hudson_loop(pairwise_combn, n, freq_alt, freq_ref, by_locus=False, return_num_dem=False, ctx=None):
    """src/pairwise_fst_hudson_loop.cpp:5-63"""
    return _fst_loop("Hudson", pairwise_combn, n, freq_alt, freq_ref, None, by_locus, return_num_dem, ctx)


def pairwise_fst_wc84_loop(pairwise_combn, n, freq_alt, het_obs, by_locus=False, return_num_dem=False, ctx=None):
    """src/pairwise_fst_wc84_loop.cpp:5-121"""
    return _fst_loop("WC84", pairwise_combn, n, freq_alt, None, het_obs, by_locus, return_num_dem, ctx)


def pairwise_fst_nei87_loop(pairwise_combn, n, het_obs, freq_alt, freq_ref, by_locus=False, return_num_dem=False,
                            ctx=None):
    """src/pairwise_fst_nei87_loop.cpp:5-115"""
    return _fst_loop("Nei87", pairwise_combn, n, freq_alt, freq_ref, het_obs, by_locus, return_num_dem, ctx)


# ---------------------------------------------------------------------------
# PCA

def pca_center_scale(v: View):
    center, scale = np.zeros(v.m), np.zeros(v.m)
    check(lib.tpg_pca_center_scale(v.ctx.h, v.h, _ptr(center), _ptr(scale)))
    return center, scale


def pca_gram(v: View, center, scale) -> np.ndarray:
    K = np.zeros((v.n, v.n), order="F")
    center, scale = _f64(center), _f64(scale)
    check(lib.tpg_pca_gram(v.ctx.h, v.h, _ptr(center), _ptr(scale), _ptr(K)))
    return K


def sym_eig_topk(K, k: int, ctx: Optional[Context] = None):
    """top-k eigenpairs (descending) of a symmetric PSD matrix: (lambda[k], U n x k)"""
    ctx = ctx or default_context()
    K = np.asfortranarray(K, dtype=float)
    n = K.shape[0]
    lam, U = np.zeros(k), np.zeros((n, k), order="F")
    check(lib.tpg_sym_eig_topk(ctx.h, _ptr(K), C.c_int64(n), C.c_int(k), _ptr(lam), _ptr(U)))
    return lam, U


def pca_loadings(v: View, center, scale, U, d) -> np.ndarray:
    """v = Z'u/d for the loci of the view (second sweep of big_SVD)"""
    center, scale, d = _f64(center), _f64(scale), _f64(d)
    U = np.asfortranarray(U, dtype=float)
    k = U.shape[1]
    out = np.zeros((v.m, k), order="F")
    check(lib.tpg_pca_loadings(v.ctx.h, v.h, _ptr(center), _ptr(scale), _ptr(U), _ptr(d), C.c_int(k), _ptr(out)))
    return out


def gt_pca_partialSVD(X: FBM, ind_row=None, ind_col=None, k: int = 10, total_var: bool = True,
                      code256=CODE_IMPUTE_PRED) -> dict:
    """R/gt_pca_partialSVD.R:67-108: the imputed code table is switched on (:74-77), then
    bigstatsr::big_SVD with bigsnpr::snp_scaleBinom."""
    v = View(X, ind_row, ind_col, code256=code256)
    d = np.zeros(k)
    u = np.zeros((v.n, k), order="F")
    vl = np.zeros((v.m, k), order="F")
    center, scale = np.zeros(v.m), np.zeros(v.m)
    fro = C.c_double()
    check(lib.tpg_pca_partial_svd(v.ctx.h, v.h, C.c_int(k), _ptr(d), _ptr(u), _ptr(vl), _ptr(center), _ptr(scale),
                                  C.byref(fro) if total_var else None))
    out = dict(d=d, u=u, v=vl, center=center, scale=scale, method="partialSVD")
    if total_var:
        out["square_frobenius"] = fro.value
    return out


def gt_pca_randomSVD(X: FBM, ind_row=None, ind_col=None, k: int = 10, tol: float = 1e-4, total_var: bool = True,
                     code256=CODE_IMPUTE_PRED) -> dict:
    """R/gt_pca_randomSVD.R:77-135.  The reference reaches the truncated SVD of the scaled matrix through
    bigstatsr::big_randomSVD (RSpectra::svds on the implicit operator), which accepts a singular triplet at the
    relative residual `tol`; the device path runs the same Gram + subspace iteration as gt_pca_partialSVD and stops
    at that tolerance: |K u_j - d_j^2 u_j| <= tol * d_1^2."""
    if not (0 < tol < 1):
        raise ValueError("tol must be in (0, 1)")
    v = View(X, ind_row, ind_col, code256=code256)
    d = np.zeros(k)
    u = np.zeros((v.n, k), order="F")
    vl = np.zeros((v.m, k), order="F")
    center, scale = np.zeros(v.m), np.zeros(v.m)
    fro = C.c_double()
    check(lib.tpg_pca_random_svd(v.ctx.h, v.h, C.c_int(k), C.c_double(tol), _ptr(d), _ptr(u), _ptr(vl), _ptr(center),
                                 _ptr(scale), C.byref(fro) if total_var else None))
    out = dict(d=d, u=u, v=vl, center=center, scale=scale, method="randomSVD")
    if total_var:
        out["square_frobenius"] = fro.value
    return out


def predict_gt_pca(pca: dict, X: Optional[FBM] = None, ind_row=None, ind_col=None, project_method: str = "none",
                   lsq_pcs=(1, 2), code256=CODE_IMPUTE_PRED):
    """predict.gt_pca (R/predict_gt_pca.R:73-236), numeric part: `pca` = the dict gt_pca_partialSVD returns; ind_col =
    the FBM columns of new_data that match the loci of the PCA, in the PCA's order (the reference derives them by
    matching locus names, :113-127).  X = None -> the scores U D of the data the PCA was built on (:101-106).
      "none"           X V on the imputed code table (bigstatsr::big_prodMat, :141-149; missing values are not allowed)
      "simple"         fbm256_prod_and_rowSumsSq, missing genotypes count as 0 (:157-175)
      "least_squares"  per individual, regress its non-missing scaled genotypes on v[, lsq_pcs] (:187-232)
      "OADP"           XV and the squared norms come from the same device sweep, but the final K x K transform is
                       bigsnpr's OADP_proj, which is not in the reference checkout: use oadp_inputs() and bigsnpr."""
    if project_method not in ("none", "simple", "OADP", "least_squares"):
        raise ValueError("'arg' should be one of 'none', 'simple', 'OADP', 'least_squares'")
    if X is None:
        return np.asfortranarray(pca["u"] * pca["d"])
    if project_method == "OADP":
        raise NotImplementedError("OADP projection calls bigsnpr::OADP_proj (third-party, not in the reference checkout); "
                                  "oadp_inputs() returns the XV and X_norm it takes")
    Vl = np.asfortranarray(pca["v"], dtype=float)
    if project_method in ("none", "simple"):
        v = View(X, ind_row, ind_col, code256=code256)
        if project_method == "none" and loci_counts(v)[:, 3].any():
            raise ValueError("You can't have missing values in 'X'.")  # bigstatsr's check on the code table
        XV, _ = _prod_and_rss(v, pca["center"], pca["scale"], Vl)
        return XV
    lsq = np.asarray(lsq_pcs)
    if (len(lsq) == 0 or np.any(np.isnan(lsq.astype(float))) or np.any(lsq < 1) or np.any(lsq > Vl.shape[1])
            or np.any(lsq != lsq.astype(int))):
        raise ValueError(f"lsq_pcs should be a vector of valid component indices (positive integers between 1 and "
                         f"{Vl.shape[1]}), e.g., c(1, 2) or c(1, 2, 3)")
    if len(set(lsq.tolist())) != len(lsq):
        raise ValueError("lsq_pcs should not contain duplicate values")
    lsq = lsq.astype(int) - 1
    L = len(lsq)
    v = View(X, ind_row, ind_col, code256=code256)
    Vs = np.asfortranarray(Vl[:, lsq])
    # right-hand sides crossprod(v_sub, g_scaled): the missing -> 0 product; matrices crossprod(v_sub): masked sums of
    # the pairwise products of the chosen columns of v
    rhs, _ = _prod_and_rss(v, pca["center"], pca["scale"], Vs)
    pairs = [(a, b) for a in range(L) for b in range(a, L)]
    tab = np.asfortranarray(np.stack([Vs[:, a] * Vs[:, b] for a, b in pairs], axis=1))
    masked = np.zeros((v.n, len(pairs)), order="F")
    check(lib.tpg_fbm256_valid_prod(v.ctx.h, v.h, _ptr(tab), C.c_int(len(pairs)), _ptr(masked)))
    out = np.zeros((v.n, L), order="F")
    for i in range(v.n):
        A = np.zeros((L, L))
        for (a, b), val in zip(pairs, masked[i]):
            A[a, b] = A[b, a] = val
        out[i] = np.linalg.solve(A, rhs[i])  # solve(crossprod(v_sub), crossprod(v_sub, genotypes_scaled)), :228
    return out


def oadp_inputs(pca: dict, X: FBM, ind_row=None, ind_col=None, code256=CODE_IMPUTE_PRED):
    """(XV, X_norm) that predict(project_method = "OADP") hands to bigsnpr::OADP_proj (R/predict_gt_pca.R:157-181)"""
    v = View(X, ind_row, ind_col, code256=code256)
    return _prod_and_rss(v, pca["center"], pca["scale"], np.asfortranarray(pca["v"], dtype=float))


def _prod_and_rss(v: View, center, scale, V):
    if V.shape[0] != v.m:
        raise ValueError("Incompatibility between dimensions.")  # bigstatsr myassert_size
    XV = np.zeros((v.n, V.shape[1]), order="F")
    rss = np.zeros(v.n)
    center, scale = _f64(center), _f64(scale)
    check(lib.tpg_fbm256_prod_and_rowSumsSq(v.ctx.h, v.h, _ptr(center), _ptr(scale), _ptr(V), C.c_int(V.shape[1]),
                                            _ptr(XV), _ptr(rss)))
    return XV, rss


def fbm256_prod_and_rowSumsSq(X: FBM, ind_row, ind_col, center, scale, V, code256="fbm"):
    """src/fbm_prod_and_rowSumSq.cpp:10-47 -> (XV (n, K), rowSumsSq (n,))"""
    v = View(X, ind_row, ind_col, code256=code256)
    V = np.asfortranarray(V, dtype=float)
    if V.shape[0] != v.m:
        raise ValueError("Incompatibility between dimensions.")  # bigstatsr myassert_size
    XV = np.zeros((v.n, V.shape[1]), order="F")
    rss = np.zeros(v.n)
    center, scale = _f64(center), _f64(scale)
    check(lib.tpg_fbm256_prod_and_rowSumsSq(v.ctx.h, v.h, _ptr(center), _ptr(scale), _ptr(V),
                                            C.c_int(V.shape[1]), _ptr(XV), _ptr(rss)))
    return XV, rss


def square_frobenius(X: FBM, ind_row, ind_col, center, scale, code256=CODE_IMPUTE_PRED) -> float:
    """R/square_frobenius.R:19-35"""
    v = View(X, ind_row, ind_col, code256=code256)
    center, scale = _f64(center), _f64(scale)
    if len(center) != v.m or len(scale) != v.m:
        raise ValueError("center and scale must be the same length as the number of columns in the matrix")
    out = C.c_double()
    check(lib.tpg_square_frobenius(v.ctx.h, v.h, _ptr(center), _ptr(scale), C.byref(out)))
    return out.value


def block_size(n: int, ncores: int = 1) -> int:
    """bigstatsr::block_size (recalled)"""
    return max(1, int(math.floor(1024.0 ** 3 / (8.0 * n * ncores))))


block_size_default = block_size


def cut_by_size(m: int, block_size: int):
    """CutBySize (R/local_reimplementations.R:13-15) = bigparallelr::split_len(m, nb = ceiling(m / block.size)) (third
    party, recalled: upper_b = round(b m / nb) with R's round-half-even): (lower, upper), 1-based inclusive -- the blocks the
    R drivers loop over (R/snp_ibs.R:59-82)"""
    nb = int(math.ceil(m / block_size))
    up = np.rint(np.arange(1, nb + 1, dtype=np.float64) * (m / nb)).astype(np.int64)
    lo = np.concatenate([[1], up[:-1] + 1])
    return lo.astype(np.int32), up.astype(np.int32)
