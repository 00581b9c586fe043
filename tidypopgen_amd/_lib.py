"""ctypes binding of libtpg_hip.so (the C ABI declared in include/tpg.h).

There is no CPU fallback: if the shared library has not been built, importing
this module raises; if no HIP device is usable, creating a context raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TPG_LIB_PATH: another build of the same library (A/B timing of kernel variants inside one GPU job: tools/enc_ab.py)
LIB_PATH = os.environ.get("TPG_LIB_PATH") or os.path.join(_HERE, "libtpg_hip.so")


class TpgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[tpg error {code}] {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess

    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j8", "-s"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C tidypopgen_amd/csrc). "
        "tidypopgen_amd has no CPU fallback."
    )

lib = C.CDLL(LIB_PATH)

c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_f64p = C.POINTER(C.c_double)
vp = C.c_void_p

lib.tpg_last_error.restype = C.c_char_p
lib.tpg_version.restype = C.c_char_p
lib.tpg_comm_transport.restype = C.c_char_p
lib.tpg_comm_transport.argtypes = [C.c_void_p]
lib.tpg_pairwise_buffer_bytes.restype = C.c_size_t
lib.tpg_pairwise_buffer_bytes.argtypes = [C.c_int64]
lib.tpg_view_n.restype = C.c_int64
lib.tpg_view_m.restype = C.c_int64
lib.tpg_view_n.argtypes = [vp]
lib.tpg_view_m.argtypes = [vp]
lib.tpg_as_pad_quirk_blocks.restype = C.c_int64
lib.tpg_as_pad_quirk_blocks.argtypes = [C.c_int64, C.c_int64]
lib.tpg_pairwise_set_as_pad_quirk.argtypes = [vp, C.c_int64]
lib.tpg_pairwise_products.argtypes = [vp]
lib.tpg_filter_high_relatedness.argtypes = [vp, vp, C.c_int64, C.c_double, vp, vp]
lib.tpg_multi_ctx.restype = vp
lib.tpg_multi_comm.restype = vp
lib.tpg_multi_ctx.argtypes = [vp, C.c_int]
lib.tpg_multi_comm.argtypes = [vp, C.c_int]
lib.tpg_multi_ndev.argtypes = [vp]
lib.tpg_comm_rank.argtypes = [vp]
lib.tpg_comm_size.argtypes = [vp]
lib.tpg_pairwise_buffer_bytes_sharded.restype = C.c_size_t
lib.tpg_pairwise_buffer_bytes_sharded.argtypes = [C.c_int64, C.c_int]
HOST_ALLREDUCE = C.CFUNCTYPE(C.c_int, vp, vp, C.c_int64, C.c_int)
for _name in ("tpg_ctx_destroy", "tpg_fbm_free", "tpg_view_free", "tpg_pairwise_free", "tpg_dev_free",
              "tpg_comm_destroy", "tpg_multi_destroy"):
    getattr(lib, _name).restype = None
    getattr(lib, _name).argtypes = [vp]

# every symbol include/tpg.h declares (checked by tests/test_abi.py against the header)
SYMBOLS = [
    "tpg_last_error", "tpg_version", "tpg_device_count", "tpg_host_bind_near_device", "tpg_ctx_create", "tpg_ctx_destroy", "tpg_ctx_set_stream", "tpg_ctx_sync",
    "tpg_prof_enable", "tpg_prof_reset", "tpg_prof_only", "tpg_prof_get", "tpg_prof_dump", "tpg_dev_alloc", "tpg_dev_free",
    "tpg_dev_to_host", "tpg_dev_from_host", "tpg_sym_eig_topk", "tpg_pca_loadings", "tpg_pairwise_pop_fst_sums", "tpg_fbm_from_host", "tpg_fbm_open_bk",
    "tpg_fbm_synth", "tpg_fbm_alloc", "tpg_fbm_upload_cols", "tpg_pca_gram_add", "tpg_fbm_open_bed", "tpg_fbm_from_bed_host", "tpg_fbm_alloc_bed", "tpg_fbm_upload_bed_snps", "tpg_fbm_to_host", "tpg_fbm_free", "tpg_view_create", "tpg_view_create_pair", "tpg_view_create_from_host", "tpg_view_free", "tpg_view_n",
    "tpg_view_m", "tpg_view_unpack", "tpg_loci_counts", "tpg_indiv_counts", "tpg_gt_ind_hetero", "tpg_gt_pi_diploid",
    "tpg_gt_grouped_pi_diploid", "tpg_grouped_genotype_counts", "tpg_pop_global_stats", "tpg_pop_basic_stats", "tpg_window_stats", "tpg_pbs_from_fst", "tpg_alt_freq_dip_pseudo",
    "tpg_grouped_alt_freq_dip_pseudo", "tpg_grouped_missingness", "tpg_grouped_summaries_dip_pseudo",
    "tpg_pairwise_pop_fst", "tpg_pairwise_fst_loop", "tpg_pairwise_buffer_bytes", "tpg_pairwise_create",
    "tpg_pairwise_free", "tpg_pairwise_zero", "tpg_pairwise_accumulate", "tpg_pairwise_counts", "tpg_pairwise_ibs",
    "tpg_pairwise_king", "tpg_pairwise_allele_sharing", "tpg_pairwise_grm", "tpg_pairwise_epilogues", "tpg_block_means", "tpg_increment_ibs_counts",
    "tpg_increment_king_numerator", "tpg_increment_as_counts", "tpg_pca_center_scale", "tpg_pca_gram",
    "tpg_pca_partial_svd", "tpg_fbm256_prod_and_rowSumsSq", "tpg_square_frobenius",
    "tpg_pairwise_set_as_pad_quirk", "tpg_as_pad_quirk_blocks", "tpg_increment_defer", "tpg_increment_flush", "tpg_resident_drop",
    "tpg_increment_as_note_narrow_block", "tpg_filter_high_relatedness", "tpg_pca_random_svd",
    "tpg_fbm256_valid_prod", "tpg_comm_unique_id", "tpg_comm_init_rank", "tpg_comm_init_host", "tpg_comm_destroy",
    "tpg_comm_rank", "tpg_comm_size", "tpg_comm_transport", "tpg_shard_loci", "tpg_comm_allreduce_f64", "tpg_pairwise_buffer_bytes_sharded",
    "tpg_pairwise_create_sharded", "tpg_pairwise_reduce", "tpg_pairwise_band", "tpg_pairwise_band_of", "tpg_pairwise_epilogues_sharded",
    "tpg_pca_partial_svd_sharded", "tpg_multi_create", "tpg_multi_destroy", "tpg_multi_ndev", "tpg_multi_ctx", "tpg_multi_comm", "tpg_multi_pairwise",
    "tpg_multi_grouped_alt_freq", "tpg_multi_pop_fst", "tpg_multi_pca_partial_svd",
    "tpg_pairwise_accumulate_products", "tpg_pairwise_products", "tpg_pairwise_reduce_begin", "tpg_pairwise_reduce_end",
    "tpg_stream_open_host", "tpg_stream_open_bk", "tpg_stream_open_bed", "tpg_stream_open_bed_host", "tpg_stream_open_synth",
    "tpg_stream_close", "tpg_stream_run", "tpg_multi_stream_run",
]


class StreamJob(C.Structure):
    """tpg_stream_job of include/tpg.h, field for field"""
    _fields_ = [
        ("struct_size", C.c_size_t), ("rowInd1", vp), ("n", C.c_int64), ("colInd1", vp), ("m", C.c_int64),
        ("ibs_type", C.c_int), ("ibs", vp), ("king", vp), ("allele_sharing", vp), ("grm", vp),
        ("code256", vp), ("ploidy", vp), ("groupIds0", vp), ("ngroups", C.c_int), ("as_counts", C.c_int),
        ("alt_freq", vp), ("grouped_alt_freq", vp), ("grouped_missingness", vp), ("loci_counts", vp),
        ("nfst", C.c_int), ("fst_method", C.c_int * 3), ("pairs1", vp), ("P", C.c_int), ("fst_return_num_dem", C.c_int),
        ("fst_tot", vp * 3), ("fst_by_locus", vp * 3), ("fst_by_locus_den", vp * 3),
        ("code256_pca", vp), ("k", C.c_int), ("pca_tol", C.c_double),
        ("d", vp), ("u", vp), ("v", vp), ("center", vp), ("scale", vp), ("square_frobenius", vp),
    ]


class StreamReport(C.Structure):
    """tpg_stream_report of include/tpg.h"""
    _fields_ = [
        ("blocks", C.c_int64), ("block_loci", C.c_int64), ("sweeps", C.c_int), ("views_kept", C.c_int),
        ("bytes_up", C.c_size_t), ("bytes_down", C.c_size_t), ("budget_bytes", C.c_size_t), ("planned_bytes", C.c_size_t),
        ("state_bytes", C.c_size_t), ("peak_device_bytes", C.c_size_t), ("seconds", C.c_double),
        ("seconds_first_sweep", C.c_double),
    ]


lib.tpg_stream_close.restype = None
lib.tpg_stream_close.argtypes = [vp]
lib.tpg_stream_run.argtypes = [vp, vp, C.POINTER(StreamJob), C.POINTER(StreamReport)]
lib.tpg_multi_stream_run.argtypes = [vp, vp, C.POINTER(StreamJob), C.POINTER(StreamReport)]


def check(rc: int) -> None:
    if rc != 0:
        raise TpgError(rc, lib.tpg_last_error().decode("utf-8", "replace"))
