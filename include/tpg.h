/*
 * tpg.h -- C ABI of the MI355X-native tidypopgen hot path (libtpg_hip.so).
 *
 * This is the drop-in boundary.  The reference crosses from R into native code
 * through `.Call` on Rcpp-generated shims (R/RcppExports.R:4-91, registration
 * table src/RcppExports.cpp:348-377); every entry point below states which of
 * those native functions (and which R driver loop around it) it replaces.  An R
 * shim that binds them is shown in INTEGRATION.md.
 *
 * Conventions (identical to the reference's):
 *   - the genotype store is a bigstatsr FBM.code256: uint8, column-major,
 *     element (i,j) at bytes[i + j*nrow];
 *   - rowInd / colInd are 1-based int32 (as R passes them; src/snp_ibs.cpp:35);
 *   - groupIds are 0-based int32 (R/loci_alt_freq.R:179);
 *   - code256 is double[256], NA = any NaN.  The device path packs genotypes to
 *     2 bits, so every non-NA entry of code256 that occurs must be 0, 1 or 2
 *     (CODE_012 / CODE_IMPUTE_PRED both are); anything else -> TPG_EUNSUPPORTED.
 *     code256 == NULL means "raw bytes": 0/1/2 valid, everything else missing,
 *     which is what increment_{ibs,king,as}_counts do regardless of code256
 *     (src/snp_ibs.cpp:47-54);
 *   - all matrices are column-major doubles, as R stores them;
 *   - output pointers may be host or device memory (hipMemcpyDefault).
 *
 * Every function returns 0 on success, a TPG_E* code otherwise; the message is
 * available from tpg_last_error() (thread-local).  Nothing throws across the
 * boundary.  One host thread per context at a time (R's main thread).
 * There is NO CPU fallback: without a usable HIP device every call fails.
 */
#ifndef TPG_H
#define TPG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPG_OK 0
#define TPG_EINVAL 1       /* bad argument */
#define TPG_EHIP 2         /* HIP runtime error (no device, OOM, launch failure) */
#define TPG_EUNSUPPORTED 3 /* e.g. code256 value outside {0,1,2,NA} */
#define TPG_ENUMERIC 4     /* e.g. missing value / zero scale in PCA (big_SVD stops too) */

typedef struct tpg_ctx tpg_ctx;   /* one GPU, one stream */
typedef struct tpg_fbm tpg_fbm;   /* FBM bytes resident in HBM */
typedef struct tpg_view tpg_view; /* (FBM, rowInd, colInd, code256) packed to 2 bits in HBM */
typedef struct tpg_pairwise tpg_pairwise; /* int32 N x N cross-product accumulators in HBM */
typedef struct tpg_comm tpg_comm;   /* one rank (= one context = one GPU) of a group that shares an analysis sharded by loci */
typedef struct tpg_multi tpg_multi; /* one process driving several GPUs: a context and a communicator per device */

const char* tpg_last_error(void);
const char* tpg_version(void);

/* ---- context ---------------------------------------------------------- */
/* HIP devices visible to this process (0 when there is none or the runtime fails: tpg_ctx_create then says why) */
int tpg_device_count(int* count);
int tpg_ctx_create(int device, tpg_ctx** out);
void tpg_ctx_destroy(tpg_ctx* ctx);
/* One process per GPU on a host with several NUMA nodes: keep the CALLING thread -- and every thread it starts afterwards: the
   upload, pack, download and add teams of this library are started by their caller -- on the CPUs of the node `device` hangs
   off, so that the buffers they touch first land there too.  What a launcher does with `numactl --cpunodebind`; the unmodified
   block loop of an R driver took 0.40 s spread over both sockets of the pool's hosts and 0.31 s on one.
   *node = the node bound to, or -1 when nothing was done: the host has one node, the device's node is unknown, or fewer than
   16 of the CPUs this process may use are on it.  Never an error for those; call it before the process allocates its buffers. */
int tpg_host_bind_near_device(int device, int* node);
/* use an externally owned hipStream_t (e.g. torch's current stream); NULL = own stream */
int tpg_ctx_set_stream(tpg_ctx* ctx, void* hip_stream);
int tpg_ctx_sync(tpg_ctx* ctx);
/* per-kernel HIP-event timing (on the context's stream) */
int tpg_prof_enable(tpg_ctx* ctx, int on);
int tpg_prof_reset(tpg_ctx* ctx);
/* time only the launches whose name is in the comma-separated list (NULL or "": all).  Two event records around a
   launch cost ~5 us of idle GPU on the stream: a timed run brackets the few kernels it prices, not all of them */
int tpg_prof_only(tpg_ctx* ctx, const char* names_csv);
/* total milliseconds and launch count of kernels whose name starts with `prefix` */
int tpg_prof_get(tpg_ctx* ctx, const char* prefix, double* total_ms, int64_t* launches);
/* writes "name\tlaunches\ttotal_ms\n" lines into buf (truncated to cap) */
int tpg_prof_dump(tpg_ctx* ctx, char* buf, size_t cap);

/* plain device buffers for callers without their own allocator (outputs may be device memory) */
int tpg_dev_alloc(tpg_ctx* ctx, size_t bytes, void** out);
void tpg_dev_free(void* p);
/* both copies are complete when the call returns, whatever the size (the context's stream has been waited for) */
int tpg_dev_to_host(tpg_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
int tpg_dev_from_host(tpg_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);

/* ---- genotype store (replaces the mmapped FBM) -------- */
int tpg_fbm_from_host(tpg_ctx* ctx, const uint8_t* bytes, int64_t nrow, int64_t ncol, tpg_fbm** out);
/* mmap bigstatsr's <backingfile>.bk and upload it */
int tpg_fbm_open_bk(tpg_ctx* ctx, const char* path, int64_t nrow, int64_t ncol, tpg_fbm** out);
/* An FBM that arrives block of columns by block of columns (the reference's own block loop, R/snp_ibs.R:59-82):
 * tpg_fbm_alloc reserves the HBM, tpg_fbm_upload_cols fills columns [col0, col0 + ncols) (0-based) from host memory.
 * Called from a second host thread with a context of its own, the upload of block b + 1 overlaps the pack /
 * accumulate kernels of block b (views over columns already uploaded).  bench.py's end_to_end leg is the worked example. */
int tpg_fbm_alloc(tpg_ctx* ctx, int64_t nrow, int64_t ncol, tpg_fbm** out);
int tpg_fbm_upload_cols(tpg_ctx* ctx, tpg_fbm* fbm, const uint8_t* host_cols, int64_t col0, int64_t ncols);
/* deterministic synthetic panel generated on the device (csrc/synth_common.h) */
int tpg_fbm_synth(tpg_ctx* ctx, uint64_t seed, int64_t nrow, int64_t ncol, int64_t j0, int npop,
                  uint32_t miss_thresh, int imputed_bytes, tpg_fbm** out);
/* a PLINK .bed file (SNP-major, 2 bits per genotype) as the genotype store, without
 * the 1-byte-per-genotype FBM in between (R/gen_tibble_bed.R:101-125 reads it into an FBM through bigsnpr's
 * readbina; the byte each 2-bit code would have become -- 00,01,10,11 -> 2,3,1,0 -- is what code256 is applied
 * to).  n / m are the line counts of the .fam / .bim files.  `bytes` is the payload after the 3-byte magic. */
int tpg_fbm_open_bed(tpg_ctx* ctx, const char* path, int64_t n, int64_t m, tpg_fbm** out);
int tpg_fbm_from_bed_host(tpg_ctx* ctx, const uint8_t* bytes, int64_t n, int64_t m, tpg_fbm** out);
/* the same store filled block of SNPs by block of SNPs (the .bed form of tpg_fbm_alloc / tpg_fbm_upload_cols: a block of SNPs
 * is one contiguous piece of the payload, ceil(n / 4) bytes per SNP): an uploader thread with a context of its own feeds
 * block b + 1 while views of block b are packed and analysed */
int tpg_fbm_alloc_bed(tpg_ctx* ctx, int64_t n, int64_t m, tpg_fbm** out);
int tpg_fbm_upload_bed_snps(tpg_ctx* ctx, tpg_fbm* fbm, const uint8_t* host_snps, int64_t snp0, int64_t nsnps);
int tpg_fbm_to_host(tpg_ctx* ctx, const tpg_fbm* fbm, uint8_t* bytes);
void tpg_fbm_free(tpg_fbm* fbm);

/* The (rowInd, colInd) view every reference kernel receives, packed once:
 * rowInd NULL = all rows, colInd NULL = all columns. */
int tpg_view_create(tpg_ctx* ctx, const tpg_fbm* fbm, const int32_t* rowInd1, int64_t n,
                    const int32_t* colInd1, int64_t m, const double* code256, tpg_view** out);
/* two views of the same (rowInd, colInd) through two code tables, packed from ONE read of the FBM bytes: e.g. the raw
 * view (code256_a = NULL) for the pairwise statistics and the imputed view (CODE_IMPUTE_PRED) for the PCA of the same
 * gen_tibble (R/gt_has_imputed.R:101-106 flips the FBM's code256 between exactly these two) */
int tpg_view_create_pair(tpg_ctx* ctx, const tpg_fbm* fbm, const int32_t* rowInd1, int64_t n, const int32_t* colInd1,
                         int64_t m, const double* code256_a, const double* code256_b, tpg_view** out_a,
                         tpg_view** out_b);
/* The view of HOST FBM bytes (e.g. the columns a block of an R driver loop covers) without a store that outlives the call:
 * upload, pack, release.  One code table is known, so the bytes cross PCIe as 2 bits per genotype where table and bytes
 * allow it (every table entry below 16 a code, no byte >= 16: both true for CODE_012 / CODE_IMPUTE_PRED stores), as nibbles
 * or bytes otherwise; the result is the view tpg_fbm_from_host + tpg_view_create give. */
int tpg_view_create_from_host(tpg_ctx* ctx, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1, int64_t n,
                              const int32_t* colInd1, int64_t m, const double* code256, tpg_view** out);
void tpg_view_free(tpg_view* v);
int64_t tpg_view_n(const tpg_view* v);
int64_t tpg_view_m(const tpg_view* v);
/* unpack to one byte per genotype (0,1,2,3=NA), column-major n x m (tests) */
int tpg_view_unpack(tpg_ctx* ctx, const tpg_view* v, uint8_t* codes);

/* ---- per-locus sweeps --------------------------------------------------- */
/* genotype counts per locus: out is m x 4 int32 row-major {n0,n1,n2,nNA}
 * (the counts behind a4/a6/a11; also bigstatsr::big_counts, R/loci_missingness.R:109-122) */
int tpg_loci_counts(tpg_ctx* ctx, const tpg_view* v, int32_t* out);
/* genotype counts per individual over the view's loci: out is n x 4 int32 row-major {n0,n1,n2,nNA} */
int tpg_indiv_counts(tpg_ctx* ctx, const tpg_view* v, int32_t* out);
/* SURVEY.md §8f(2) "next" rows, same sweeps on the same counts:
 * gt_ind_hetero (src/gt_ind_hetero.cpp:11-42): out is the 2 x n integer matrix {n_het; n_na} (column-major);
 * gt_pi_diploid (src/gt_pi_diploid.cpp:7-38): pi[m];
 * gt_grouped_pi_diploid (src/gt_grouped_pi_diploid.cpp:7-42): pi and n, both m x G */
int tpg_gt_ind_hetero(tpg_ctx* ctx, const tpg_view* v, int32_t* out);
int tpg_gt_pi_diploid(tpg_ctx* ctx, const tpg_view* v, double* pi);
int tpg_gt_grouped_pi_diploid(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups, double* pi,
                              double* n);
/* genotype counts per locus x group, the 3 x ngroups table gt_grouped_hwe fills before each exact test
 * (src/hwe.cpp:238-250; the test itself, PLINK's SNPHWE2, is out of scope): out = three m x G int32 matrices
 * (column-major), k = 0, 1, 2 alternate alleles */
int tpg_grouped_genotype_counts(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                                int32_t* out);
/* pop_global_stats (R/pop_global_stats.R:113-212, with compute_np_mn, src/compute_np_mn.cpp:8-34): by_locus =
 * m x 10 column-major {Ho, Hs, Ht, Dst, Htp, Dstp, Fst, Fstp, Fis, Dest} (may be NULL), overall = the 10
 * by_locus = FALSE values (may be NULL).  ploidy (may be NULL) must be all 2: the reference stops otherwise. */
int tpg_pop_global_stats(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                         const double* ploidy, double* by_locus, double* overall);
/* pop_het_obs (which = 0, R/pop_het_obs.R:78-91), pop_het_exp / pop_gene_div (1, R/pop_het_exp.R:85-103) and
 * pop_fis(method = "Nei87") (2, R/pop_fis.R:108-130): by_locus = m x G (may be NULL), colmeans = colMeans(na.rm = TRUE)
 * over the loci, G values (may be NULL).  ploidy as in tpg_pop_global_stats. */
int tpg_pop_basic_stats(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups, const double* ploidy,
                        int which, double* by_locus, double* colmeans);
/* the numeric core of windows_stats_generic (R/windows_stats_generic.R:113-176: runner::mean_run /
 * sum_run with na_rm = TRUE) for every column of the per-locus matrix x (m x ncol, column-major, host or device):
 * window w covers the loci lo[w] .. hi[w]-1 (0-based; the host side derives them from chromosome / position /
 * window_size / step_size); pad_na[w] != 0 marks a window the reference returns as NA under complete = TRUE (may be
 * NULL).  op 0 = mean, 1 = sum.  stat (nw x ncol) is NaN where the window holds no value or fewer than min_loci;
 * n_loci (nw x ncol int32, may be NULL) = values present, -1 for a pad_na window. */
int tpg_window_stats(tpg_ctx* ctx, const double* x, int64_t m, int ncol, const int64_t* lo, const int64_t* hi,
                     const uint8_t* pad_na, int64_t nw, int op, int min_loci, double* stat, int32_t* n_loci);
/* population branch statistic from a by-locus (or by-window) pairwise Fst matrix: pbs_one_triplet of
 * R/nwise_pop_pbs.R:118-156.  fst is m x P (column-major, host or device); triplet t names the columns of
 * (pop1.pop2, pop1.pop3, pop2.pop3) in trip_cols0[3t..3t+2] (0-based); out is m x 6 ntrip, six columns per triplet:
 * pbs_1, pbs_2, pbs_3, pbsn1_1, pbsn1_2, pbsn1_3 */
int tpg_pbs_from_fst(tpg_ctx* ctx, const double* fst, int64_t m, int P, const int32_t* trip_cols0, int ntrip,
                     double* out);
/* replaces alt_freq_dip_pseudo_cpp (src/alt_freq_dip_pseudo_cpp.cpp:8-58) for the whole
 * colInd at once (the big_apply block loop R/loci_alt_freq.R:351-359 collapses):
 * out m x 2 = {n_alt | freq, n_valid} */
int tpg_alt_freq_dip_pseudo(tpg_ctx* ctx, const tpg_view* v, const double* ploidy, int as_counts,
                            double* out);
/* replaces grouped_alt_freq_dip_pseudo_cpp (src/grouped_alt_freq_dip_pseudo_cpp.cpp:8-58):
 * out m x 2G */
int tpg_grouped_alt_freq_dip_pseudo(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0,
                                    int ngroups, const double* ploidy, int as_counts, double* out);
/* replaces grouped_missingness_cpp (src/grouped_missingness_cpp.cpp:8-33): out m x G */
int tpg_grouped_missingness(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                            double* out);
/* replaces grouped_summaries_dip_pseudo_cpp (src/grouped_summaries_dip_pseudo_cpp.cpp:11-63):
 * four m x G outputs (any may be NULL) */
int tpg_grouped_summaries_dip_pseudo(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0,
                                     int ngroups, const double* ploidy, double* freq_alt,
                                     double* freq_ref, double* n, double* het_obs);

/* ---- pairwise population Fst --------------------------------------------- */
#define TPG_FST_HUDSON 0
#define TPG_FST_NEI87 1
#define TPG_FST_WC84 2
/* Fused path: grouped summaries + pair loop without materialising the m x G matrices
 * (replaces R/pairwise_pop_fst.R:123-161).  pairs1 is 2 x P column-major, 1-based.
 * fst_tot[P]; out_a / out_b are m x P (by_locus ratio or numerator / denominator), may be
 * NULL when by_locus == 0. */
int tpg_pairwise_pop_fst(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                         const double* ploidy, int method, const int32_t* pairs1, int P, int by_locus,
                         int return_num_dem, double* fst_tot, double* out_a, double* out_b);
/* Same sweep, but returns the sums over this view's loci of numerator and denominator
 * (sum_num[P], sum_den[P]) instead of their ratio: SNP-block shards on different GPUs add these
 * (one all-reduce of 2P doubles) before dividing. */
int tpg_pairwise_pop_fst_sums(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                              const double* ploidy, int method, const int32_t* pairs1, int P,
                              double* sum_num, double* sum_den);
/* Literal mirrors of the three loop functions (src/pairwise_fst_hudson_loop.cpp:5-63,
 * src/pairwise_fst_wc84_loop.cpp:5-121, src/pairwise_fst_nei87_loop.cpp:5-115): inputs are
 * the m x G double matrices the reference passes; unused ones may be NULL. */
int tpg_pairwise_fst_loop(tpg_ctx* ctx, int method, const int32_t* pairs1, int P, int64_t m, int G,
                          const double* n, const double* freq_alt, const double* freq_ref,
                          const double* het_obs, int by_locus, int return_num_dem, double* fst_tot,
                          double* out_a, double* out_b);

/* ---- pairwise individual matrices (IBS / KING / allele sharing / GRM) ---- */
/* Accumulators for the four integer cross-products V=vv', D=dd', H=hh', A=hv'
 * (v valid, d = dosage-1, h heterozygous), from which every count matrix of
 * increment_ibs_counts / increment_king_numerator / increment_as_counts follows.
 * If ext_buffer != NULL it must be device memory of tpg_pairwise_buffer_bytes(n)
 * bytes (e.g. a torch tensor, so that the caller can all-reduce it over RCCL). */
size_t tpg_pairwise_buffer_bytes(int64_t n);
int tpg_pairwise_create(tpg_ctx* ctx, int64_t n, void* ext_buffer, tpg_pairwise** out);
void tpg_pairwise_free(tpg_pairwise* pw);
int tpg_pairwise_zero(tpg_ctx* ctx, tpg_pairwise* pw);
/* add loci [col_begin, col_end) of the view (0-based, end exclusive; -1 = m): all five products */
int tpg_pairwise_accumulate(tpg_ctx* ctx, tpg_pairwise* pw, const tpg_view* v, int64_t col_begin,
                            int64_t col_end);
/* The same for the products one analysis needs (the reference runs 2 / 6 / 4 dense products for allele sharing / IBS /
 * KING: src/snp_as.cpp:64-65, src/snp_ibs.cpp:67-72, src/snp_king.cpp:70-72): `products` = OR of TPG_PW_V (typed x
 * typed), TPG_PW_D (dosage-1 x dosage-1), TPG_PW_H (het x het), TPG_PW_A (het x typed, both orientations).  Each set
 * has a kernel with a wave tile of its own (fewer sums per pair leave registers for more pairs per operand byte).
 * Products that were left out stay unknown until the next tpg_pairwise_zero: the count / epilogue entry points
 * refuse (TPG_EINVAL) an output that needs one of them. */
#define TPG_PW_V 1
#define TPG_PW_D 2
#define TPG_PW_H 4
#define TPG_PW_A 8
/* D and H added up in ONE sum (IBS = V + (D + H), IBS_valid = 2 V: snp_ibs needs nothing else, src/snp_ibs.cpp:67-72).  Two
 * sums per pair instead of three leave registers for a 128 x 64 wave tile: 24 MFMAs per 6 operand fragments where the
 * {V, D, H} kernel has 12 per 4.  Goes with TPG_PW_V only; afterwards D and H are unknown on their own -- the allele-sharing
 * and KING outputs are refused -- and tpg_pairwise_products reports TPG_PW_V | TPG_PW_DH. */
#define TPG_PW_DH 16
#define TPG_PW_FOR_AS (TPG_PW_V | TPG_PW_D)               /* snp_allele_sharing, pairwise_grm */
#define TPG_PW_FOR_IBS (TPG_PW_V | TPG_PW_D | TPG_PW_H)   /* snp_ibs together with allele sharing / GRM */
#define TPG_PW_FOR_IBS_ALONE (TPG_PW_V | TPG_PW_DH)       /* snp_ibs on its own */
#define TPG_PW_FOR_KING (TPG_PW_V | TPG_PW_D | TPG_PW_A)  /* snp_king */
#define TPG_PW_ALL (TPG_PW_V | TPG_PW_D | TPG_PW_H | TPG_PW_A)
int tpg_pairwise_accumulate_products(tpg_ctx* ctx, tpg_pairwise* pw, const tpg_view* v, int64_t col_begin,
                                     int64_t col_end, int products);
/* the products whose sums are complete since the last tpg_pairwise_zero (TPG_PW_ALL when nothing was left out) */
int tpg_pairwise_products(const tpg_pairwise* pw);
/* Reference quirk Q1, opt-in.  increment_as_counts adds +1 to EVERY element of the allele-sharing
 * numerator for every block of the R driver that is one column narrower than the widest (src/snp_as.cpp:57-63 with
 * the scratch matrices of R/snp_allele_sharing.R:55-56).  The default (0 blocks) is the mathematically intended
 * value, which is what the reference's own test asserts; to reproduce a real R run bit for bit pass the number of
 * narrower blocks of that run: as_num, allele sharing and GRM then come out as R's.  tpg_as_pad_quirk_blocks gives
 * that number for m loci cut by CutBySize(m, block_size) (R/local_reimplementations.R:13-15). */
int tpg_pairwise_set_as_pad_quirk(tpg_pairwise* pw, int64_t narrow_blocks);
int64_t tpg_as_pad_quirk_blocks(int64_t m, int64_t block_size);
/* raw count matrices, n x n column-major doubles, any may be NULL:
 * ibs / ibs_valid (snp_ibs raw_counts), king_num / n_Aa_i (snp_king), as_num / as_den */
int tpg_pairwise_counts(tpg_ctx* ctx, const tpg_pairwise* pw, double* ibs, double* ibs_valid,
                        double* king_num, double* n_Aa_i, double* as_num, double* as_den);
/* epilogues of the R drivers */
#define TPG_IBS_PROPORTION 0
#define TPG_IBS_ADJUSTED_COUNTS 1
int tpg_pairwise_ibs(tpg_ctx* ctx, const tpg_pairwise* pw, int type, int64_t m, double* out); /* R/snp_ibs.R:84-103 */
int tpg_pairwise_king(tpg_ctx* ctx, const tpg_pairwise* pw, double* out);           /* R/snp_king.R:79-101 */
int tpg_pairwise_allele_sharing(tpg_ctx* ctx, const tpg_pairwise* pw, double* out); /* R/snp_allele_sharing.R:77-81 */
int tpg_pairwise_grm(tpg_ctx* ctx, const tpg_pairwise* pw, double* out);            /* R/pairwise_grm.R:42-50 */
/* all four epilogues from one pass over the accumulators; any output may be NULL */
int tpg_pairwise_epilogues(tpg_ctx* ctx, const tpg_pairwise* pw, int ibs_type, int64_t m, double* ibs,
                           double* king, double* allele_sharing, double* grm);
/* the reduction pop_fst / pop_fis(method = "WG17") make of the N x N allele-sharing matrix
 * (R/pop_fst.R:40-63, R/pop_fis.R:151-173): mean[g1 + g2 G] = mean(A[rows of g1, columns of g2], na.rm = TRUE),
 * with the diagonal of A left out when skip_diag != 0; count (may be NULL) = number of entries averaged.
 * A is n x n column-major, host or device. */
int tpg_block_means(tpg_ctx* ctx, const double* A, int64_t n, const int32_t* groupIds0, int ngroups, int skip_diag,
                    double* mean, double* count);

/* filter_high_relatedness (R/filter_high_relatedness.R:26-145) on an n x n relatedness matrix (host
 * or device memory, e.g. the KING matrix tpg_pairwise_king left in HBM): keep[i] = 1 for the individuals that pass,
 * in the ORIGINAL order (the reference's third list element); new_order0 (may be NULL) = the order of decreasing mean
 * relatedness the loop walks in (0-based), so that the ids to keep, in the reference's order, are
 * new_order0[k] for the k with keep[new_order0[k]] = 1.  An NA among the compared relatednesses is an error, as in R. */
int tpg_filter_high_relatedness(tpg_ctx* ctx, const double* matrix, int64_t n, double kings_threshold, uint8_t* keep,
                                int32_t* new_order0);

/* Literal per-block mirrors of the three increment_* entry points
 * (src/snp_ibs.cpp:22-74, src/snp_king.cpp:21-74, src/snp_as.cpp:22-67); the scratch matrices of the reference
 * are not needed.  fbm_bytes is the host (mmapped) FBM.  DEFAULT = the reference's semantics: the caller's n x n
 * doubles are incremented when the call returns (an unmodified R driver reads them right after its loop,
 * R/snp_ibs.R:84-95).  Every call uploads the columns of its own block; no copy of the caller's FBM outlives the
 * call, so an FBM that is rewritten in place between analyses (R/gt_impute_simple.R:86) is never read stale.
 * OPT-IN, tpg_increment_defer(ctx, 1): every (K, K2) pair gets device accumulators that live across the calls of the
 * R block loop (R/snp_ibs.R:69-82) and the caller's matrices are incremented by tpg_increment_flush (one line added
 * to the R driver after its loop): one N x N download per analysis instead of per block.  All
 * blocks that accumulate into the same (K, K2) must then pass the same rowInd. */
int tpg_increment_defer(tpg_ctx* ctx, int on);
int tpg_increment_ibs_counts(tpg_ctx* ctx, double* K, double* K2, const uint8_t* fbm_bytes,
                             int64_t nrow, int64_t ncol, const int32_t* rowInd1, int64_t n,
                             const int32_t* colInd1, int64_t m);
int tpg_increment_king_numerator(tpg_ctx* ctx, double* K, double* N_Aa_i, const uint8_t* fbm_bytes,
                                 int64_t nrow, int64_t ncol, const int32_t* rowInd1, int64_t n,
                                 const int32_t* colInd1, int64_t m);
int tpg_increment_as_counts(tpg_ctx* ctx, double* K, double* K2, const uint8_t* fbm_bytes,
                            int64_t nrow, int64_t ncol, const int32_t* rowInd1, int64_t n,
                            const int32_t* colInd1, int64_t m);

/* K += accumulated sums for every pending (K, K2) pair; the device accumulators are released (a no-op unless
 * tpg_increment_defer is on) */
int tpg_increment_flush(tpg_ctx* ctx);
/* release the device scratch the increment_* mirrors keep between calls (an error while increments are pending) */
int tpg_resident_drop(tpg_ctx* ctx);
/* quirk Q1 through the literal mirror (opt-in): the block just passed to tpg_increment_as_counts for the n x n
 * matrix K was one column narrower than the R driver's scratch matrices: +1 on every element, at the flush when K is
 * pending, at once otherwise */
int tpg_increment_as_note_narrow_block(tpg_ctx* ctx, double* K, int64_t n);

/* ---- SNP-block shards over the GPUs of one node -------------------------------------------------
 * The locus axis is the reference's own block axis (R/snp_ibs.R:59-82): a shard is a contiguous range of loci.
 * Per-locus outputs are disjoint slices (no exchange).  What is additive over loci is exchanged by the LIBRARY, over
 * RCCL (xGMI): the int32 pairwise slabs by ONE reduce-scatter -- rank r then finishes band r of the tiles (1 / nranks
 * of the epilogue work and of the output bytes) --, Fst sums, the PCA Gram matrix and the GRM mean by all-reduces of
 * doubles.  With one rank every exchange is the identity: the code path is the same for 1 .. 8 GPUs.
 *
 * One process per GPU (torchrun, mpirun): rank 0 calls tpg_comm_unique_id, the launcher broadcasts the 128 bytes,
 * every rank calls tpg_comm_init_rank with its own context.  One process for all GPUs (an R session): tpg_multi_*. */
int tpg_comm_unique_id(uint8_t* id128);
int tpg_comm_init_rank(tpg_ctx* ctx, int nranks, int rank, const uint8_t* id128, tpg_comm** out);
/* Rehearsal transport (tests): the caller's in-place all-reduce (sum) of `count` elements in HOST memory, dtype 0 =
 * int32, 1 = float64, returning 0 on success -- e.g. torch.distributed over gloo, so that several ranks can share
 * one GPU, which RCCL refuses. */
int tpg_comm_init_host(tpg_ctx* ctx, int nranks, int rank,
                       int (*allreduce)(void* user, void* buf, int64_t count, int dtype), void* user, tpg_comm** out);
void tpg_comm_destroy(tpg_comm* comm);
/* which transport the communicator's collectives run over: "none" (one rank), "host callback" (tpg_comm_init_host), or
 * "rccl: <library name as loaded>" (librccl.so.1 unless TPG_RCCL_LIBRARY names another) */
const char* tpg_comm_transport(const tpg_comm* comm);
int tpg_comm_rank(const tpg_comm* comm);
int tpg_comm_size(const tpg_comm* comm);
/* loci [begin, end) of `rank`: contiguous, boundaries on multiples of 128 loci, sizes differ by at most 128 */
int tpg_shard_loci(int64_t m_total, int nranks, int rank, int64_t* begin, int64_t* end);
/* in-place sum over the ranks of `count` doubles (host or device memory): Fst numerator / denominator sums
 * (tpg_pairwise_pop_fst_sums), the Gram matrix (tpg_pca_gram), the squared Frobenius norm */
int tpg_comm_allreduce_f64(tpg_ctx* ctx, tpg_comm* comm, double* buf, int64_t count);
/* pairwise accumulators laid out for the reduce-scatter; accumulate as usual, then tpg_pairwise_reduce: this rank
 * is left with the complete sums of its band of tiles, and tpg_pairwise_counts / tpg_pairwise_epilogues_sharded write
 * only the part of the N x N outputs the band covers -- rows [row0, row1) x columns [row0, n) and the mirror image
 * rows [row0, n) x columns [row0, row1) (tpg_pairwise_band); the bands of all ranks tile the matrices. */
size_t tpg_pairwise_buffer_bytes_sharded(int64_t n, int nranks);
int tpg_pairwise_create_sharded(tpg_ctx* ctx, const tpg_comm* comm, int64_t n, tpg_pairwise** out);
int tpg_pairwise_reduce(tpg_ctx* ctx, tpg_comm* comm, tpg_pairwise* pw);
/* The same reduction on a SECOND communicator -- one made on another context (= another stream) of the same device, same
 * ranks -- so that the reduce-scatter runs beside what pw's own context enqueues next (in the fused analysis: the PCA's Gram
 * kernels) instead of in front of it.  _begin orders the reduce-scatter behind the accumulate kernels already enqueued on
 * pw's context and returns at once; _end orders pw's context behind the reduce-scatter (nothing may read the accumulators in
 * between).  RCCL orders the operations of ONE communicator, hence the second one; every rank must issue _begin and the
 * collectives of its first communicator in the same order.  Rehearsed over the stream-ordered mock RCCL only:
 * opt-in, nothing in the library calls it by itself. */
int tpg_pairwise_reduce_begin(tpg_ctx* ctx, tpg_comm* side_comm, tpg_pairwise* pw);
int tpg_pairwise_reduce_end(tpg_ctx* ctx, tpg_comm* side_comm, tpg_pairwise* pw);
int tpg_pairwise_band(const tpg_pairwise* pw, int64_t* row0, int64_t* row1);
/* the band rank `rank` of `nranks` gets for n individuals (host arithmetic; no GPU needed) */
int tpg_pairwise_band_of(int64_t n, int nranks, int rank, int64_t* row0, int64_t* row1);
int tpg_pairwise_epilogues_sharded(tpg_ctx* ctx, tpg_comm* comm, const tpg_pairwise* pw, int ibs_type, int64_t m,
                                   double* ibs, double* king, double* allele_sharing, double* grm);
/* gt_pca_partialSVD with the loci sharded over the ranks: `v` holds this rank's loci; center, scale and the rows of
 * the loadings come back for those loci; d and u are the same on every rank (the Gram matrix is summed over the ranks
 * inside, the eigen step is replicated); one rank: identical to tpg_pca_partial_svd */
int tpg_pca_partial_svd_sharded(tpg_ctx* ctx, tpg_comm* comm, const tpg_view* v, int k, double* d, double* u,
                                double* vload, double* center, double* scale, double* square_frobenius);
/* one process, `ndev` GPUs (devices == NULL: 0 .. ndev-1): a context and a communicator (ncclCommInitAll) each */
int tpg_multi_create(int ndev, const int* devices, tpg_multi** out);
void tpg_multi_destroy(tpg_multi* mg);
int tpg_multi_ndev(const tpg_multi* mg);
tpg_ctx* tpg_multi_ctx(tpg_multi* mg, int i);
tpg_comm* tpg_multi_comm(tpg_multi* mg, int i);
/* snp_ibs + snp_king + snp_allele_sharing + pairwise_grm of one HOST FBM on all devices: every device uploads and
 * packs its share of colInd (raw-byte semantics, src/snp_ibs.cpp:47-54), one reduce-scatter, every device writes its
 * band of IBS / KING / allele sharing / GRM straight into the caller's n x n host matrices (any may be NULL).
 * m for TPG_IBS_ADJUSTED_COUNTS is the number of loci kept. */
int tpg_multi_pairwise(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1,
                       int64_t n, const int32_t* colInd1, int64_t m, int ibs_type, double* ibs, double* king,
                       double* allele_sharing, double* grm);
/* The other analyses of one HOST FBM on all devices of `mg`, for a caller that is ONE process (an R session): every
 * device uploads and packs its share of colInd (through code256; NULL = raw bytes) and the results land in the
 * caller's arrays (host or device memory).
 *   tpg_multi_grouped_alt_freq: loci_alt_freq of a grouped gen_tibble (R/loci_alt_freq.R:174-197), out m x 2G; with
 *     groupIds0 == NULL the ungrouped form (R/loci_alt_freq.R:328-379), out m x 2.  Per-locus outputs: no exchange.
 *   tpg_multi_pop_fst: pairwise_pop_fst (R/pairwise_pop_fst.R:116-161), arguments as tpg_pairwise_pop_fst; by-locus
 *     rows come from the device that owns the locus, totals from the numerator / denominator sums of all devices.
 *   tpg_multi_pca_partial_svd: gt_pca_partialSVD (R/gt_pca_partialSVD.R:67-108), arguments as tpg_pca_partial_svd;
 *     the Gram matrix is summed over the devices by one all-reduce (RCCL). */
int tpg_multi_grouped_alt_freq(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1,
                               int64_t n, const int32_t* colInd1, int64_t m, const double* code256,
                               const int32_t* groupIds0, int ngroups, const double* ploidy, int as_counts, double* out);
int tpg_multi_pop_fst(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1, int64_t n,
                      const int32_t* colInd1, int64_t m, const double* code256, const int32_t* groupIds0, int ngroups,
                      const double* ploidy, int method, const int32_t* pairs1, int P, int by_locus, int return_num_dem,
                      double* fst_tot, double* out_a, double* out_b);
int tpg_multi_pca_partial_svd(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1,
                              int64_t n, const int32_t* colInd1, int64_t m, const double* code256, int k, double* d,
                              double* u, double* vload, double* center, double* scale, double* square_frobenius);

/* ---- streamed whole analyses: the reference's block loop inside the library ---------------------------------------
 * What defines the reference on this path is that the genotype store is a FILE and is swept in blocks of loci
 * (R/snp_ibs.R:59-82, R/loci_alt_freq.R:351-359, R/gen_tibble_fbm.R:185-194; big_SVD behind
 * R/gt_pca_partialSVD.R:82-89 sweeps it twice).  A tpg_stream is such a store that STAYS ON THE HOST: tpg_stream_run
 * sweeps colInd in blocks of loci, an uploader thread (its own stream) filling one of two block buffers while the kernels
 * of the block before run, and keeps resident only those two blocks of FBM bytes, the packed views of the block at hand
 * and the additive state (pairwise slabs, Gram matrix, Fst sums); per-locus outputs leave block by block (a downloader
 * thread, its own stream) into the caller's arrays.  The PCA's second sweep (the loadings v = Z'u / d) reads the imputed
 * views again if the budget let them stay (n m / 4 bytes), and streams the store a second time otherwise.
 * budget_bytes bounds the HBM taken by FBM bytes + packed views + per-block scratch (0 = no bound: a few large blocks,
 * views kept -- the fastest end-to-end route for a panel that fits); the additive state is not part of it.
 * Results are those of the resident entry points: integer counts bit for bit, FP64 sums in block order. */
typedef struct tpg_stream tpg_stream;
/* the host FBM bytes (e.g. the mmap of bigstatsr's .bk; must stay valid until tpg_stream_close) */
int tpg_stream_open_host(tpg_ctx* ctx, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, size_t budget_bytes,
                         tpg_stream** out);
/* <backingfile>.bk, mapped by the library */
int tpg_stream_open_bk(tpg_ctx* ctx, const char* path, int64_t nrow, int64_t ncol, size_t budget_bytes, tpg_stream** out);
/* a PLINK .bed: path (magic checked), or the payload behind its 3-byte magic */
int tpg_stream_open_bed(tpg_ctx* ctx, const char* path, int64_t n, int64_t m, size_t budget_bytes, tpg_stream** out);
int tpg_stream_open_bed_host(tpg_ctx* ctx, const uint8_t* payload, int64_t n, int64_t m, size_t budget_bytes,
                             tpg_stream** out);
/* the synthetic panel of tpg_fbm_synth generated block by block on the device (panels larger than host memory) */
int tpg_stream_open_synth(tpg_ctx* ctx, uint64_t seed, int64_t nrow, int64_t ncol, int npop, uint32_t miss_thresh,
                          int imputed_bytes, size_t budget_bytes, tpg_stream** out);
void tpg_stream_close(tpg_stream* s);

/* What one sweep computes: every output pointer is optional (NULL = not asked for), host or device memory, laid out as
 * the resident entry point of the same name lays it out.  One (rowInd, colInd) selection serves all of them. */
#define TPG_STREAM_MAX_FST 3
typedef struct tpg_stream_job {
  size_t struct_size;     /* sizeof(tpg_stream_job) of the caller's header */
  const int32_t* rowInd1; /* NULL = all rows */
  int64_t n;
  const int32_t* colInd1; /* NULL = all columns */
  int64_t m;
  /* snp_ibs / snp_king / snp_allele_sharing / pairwise_grm (raw-byte semantics, src/snp_ibs.cpp:47-54): n x n.  Only
   * the cross-products the requested matrices are made of are accumulated (tpg_pairwise_accumulate_products) */
  int ibs_type;
  double *ibs, *king, *allele_sharing, *grm;
  /* per-locus sweeps through code256 (NULL = raw bytes) */
  const double* code256;
  const double* ploidy;      /* NULL = all diploid */
  const int32_t* groupIds0;  /* needed by the grouped outputs and by Fst */
  int ngroups;
  int as_counts;
  double* alt_freq;            /* m x 2   (tpg_alt_freq_dip_pseudo) */
  double* grouped_alt_freq;    /* m x 2G  (tpg_grouped_alt_freq_dip_pseudo) */
  double* grouped_missingness; /* m x G   (tpg_grouped_missingness) */
  int32_t* loci_counts;        /* m x 4 row-major (tpg_loci_counts) */
  /* pairwise_pop_fst: up to three estimators from one sweep; fst_tot[i] (P, may be NULL) = ratio of the sums over all
   * loci; fst_by_locus[i] (m x P, may be NULL) = the by_locus ratios -- or, with fst_return_num_dem, the numerators, and
   * fst_by_locus_den[i] (m x P) the denominators (R/pairwise_pop_fst.R:103-106) */
  int nfst;
  int fst_method[TPG_STREAM_MAX_FST];
  const int32_t* pairs1; /* 2 x P, 1-based */
  int P;
  int fst_return_num_dem;
  double* fst_tot[TPG_STREAM_MAX_FST];
  double* fst_by_locus[TPG_STREAM_MAX_FST];
  double* fst_by_locus_den[TPG_STREAM_MAX_FST];
  /* gt_pca_partialSVD through code256_pca (e.g. CODE_IMPUTE_PRED); k = 0: no PCA.  pca_tol = 0: the partial SVD's
   * 1e-12, else tpg_pca_random_svd's tolerance */
  const double* code256_pca;
  int k;
  double pca_tol;
  double *d, *u, *v, *center, *scale, *square_frobenius;
} tpg_stream_job;

typedef struct tpg_stream_report {
  int64_t blocks;          /* blocks of the first sweep */
  int64_t block_loci;      /* loci per block (the last one may be narrower) */
  int sweeps;              /* 1, or 2 when the loadings streamed the store again */
  int views_kept;          /* the imputed views stayed in HBM for the loadings */
  size_t bytes_up;         /* host -> device, all sweeps */
  size_t bytes_down;       /* device -> host */
  size_t budget_bytes;     /* as given */
  size_t planned_bytes;    /* what the block plan expects to hold: FBM blocks + views + scratch (<= budget when one was given) */
  size_t state_bytes;      /* additive state (pairwise slabs, Gram matrices, N x N outputs): not part of the budget */
  size_t peak_device_bytes; /* largest growth of the device's used memory over the run (hipMemGetInfo, sampled per block) */
  double seconds;
  double seconds_first_sweep;
} tpg_stream_report;

/* one streamed pass over the store for everything the job asks for; report may be NULL */
int tpg_stream_run(tpg_ctx* ctx, tpg_stream* s, const tpg_stream_job* job, tpg_stream_report* report);
/* The same with the loci sharded over the devices of `mg` ("then stream sub-blocks per GPU"): every device
 * streams its contiguous share of colInd in blocks under the same budget (per device), then one reduce-scatter of the
 * pairwise slabs, all-reduces of the Fst sums and of the Gram matrix, replicated eigen step, each device the loadings of its
 * share.  Outputs must be host memory (device threads write disjoint pieces).  `s` supplies the store and the budget. */
int tpg_multi_stream_run(tpg_multi* mg, tpg_stream* s, const tpg_stream_job* job, tpg_stream_report* report);

/* ---- PCA (gt_pca_partialSVD) ---------------------------------------------- */
/* center / scale of bigsnpr::snp_scaleBinom; TPG_ENUMERIC on a missing value or zero scale */
int tpg_pca_center_scale(tpg_ctx* ctx, const tpg_view* v, double* center, double* scale);
/* Gram matrix K = Z Z' (n x n) accumulated behind bigstatsr::big_SVD
 * (call site R/gt_pca_partialSVD.R:82-89) */
int tpg_pca_gram(tpg_ctx* ctx, const tpg_view* v, const double* center, const double* scale, double* K);
/* K (device memory, n x n) += Gram matrix of this view's loci: block-by-block accumulation for callers that receive
 * the genotypes in blocks of loci; tpg_sym_eig_topk and tpg_pca_loadings finish the SVD */
int tpg_pca_gram_add(tpg_ctx* ctx, const tpg_view* v, const double* center, const double* scale, double* K);
/* full partial SVD: d[k], u n x k, v m x k, center[m], scale[m]; square_frobenius may be NULL
 * (R/square_frobenius.R:19-35).  Any k <= n (the eigen solver works on a block of at most 64 vectors: beyond 52
 * components the spectrum is taken in batches of 26 with explicit deflation in between). */
int tpg_pca_partial_svd(tpg_ctx* ctx, const tpg_view* v, int k, double* d, double* u, double* vload,
                        double* center, double* scale, double* square_frobenius);
/* gt_pca_randomSVD (R/gt_pca_randomSVD.R:77-135): the same truncated SVD accepted at the
 * relative residual `tol` of the reference's big_randomSVD / RSpectra path (default there 1e-4):
 * |K u_j - d_j^2 u_j| <= tol * d_1^2 for every returned pair. */
int tpg_pca_random_svd(tpg_ctx* ctx, const tpg_view* v, int k, double tol, double* d, double* u, double* vload,
                       double* center, double* scale, double* square_frobenius);
/* The pieces of tpg_pca_partial_svd for SNP-block shards on several GPUs: every rank computes the Gram
 * matrix of its loci (tpg_pca_gram, additive over loci), the N x N partials are summed (one all-reduce),
 * then tpg_sym_eig_topk gives lambda[k] (descending) and U (n x k) of the summed matrix and
 * tpg_pca_loadings the rows of v = Z'u/d that belong to the rank's loci (d = sqrt(lambda)). */
/* K: BOTH triangles filled and bitwise symmetric (K[i + j n] == K[j + i n]; what tpg_pca_gram / tpg_pca_gram_add write): the
 * products K Q read K by rows or by columns, whichever is faster, so a matrix that is symmetric only to rounding, or has one
 * triangle filled, gives the eigenpairs of neither. */
int tpg_sym_eig_topk(tpg_ctx* ctx, const double* K, int64_t n, int k, double* lambda, double* U);
int tpg_pca_loadings(tpg_ctx* ctx, const tpg_view* v, const double* center, const double* scale,
                     const double* U, const double* d, int k, double* vload);
/* replaces fbm256_prod_and_rowSumsSq (src/fbm_prod_and_rowSumSq.cpp:10-47): V m x K,
 * XV n x K, rss[n] */
int tpg_fbm256_prod_and_rowSumsSq(tpg_ctx* ctx, const tpg_view* v, const double* center,
                                  const double* scale, const double* V, int K, double* XV, double* rss);
int tpg_square_frobenius(tpg_ctx* ctx, const tpg_view* v, const double* center, const double* scale,
                         double* out);
/* out (n x K) [i, k] = sum of Tab[j, k] (m x K) over the loci j at which individual i is typed: the masked sums
 * behind predict(project_method = "least_squares") (R/predict_gt_pca.R:221-228) */
int tpg_fbm256_valid_prod(tpg_ctx* ctx, const tpg_view* v, const double* Tab, int K, double* out);

#ifdef __cplusplus
}
#endif
#endif /* TPG_H */
