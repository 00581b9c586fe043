// common.h -- internal declarations shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/tpg.h"

// ---------------------------------------------------------------------------
// error plumbing (nothing throws across the C ABI)
void tpg_set_error(const char* fmt, ...);

#define TPG_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t _e = (call);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      tpg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e));     \
      return TPG_EHIP;                                                                       \
    }                                                                                        \
  } while (0)

#define TPG_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      tpg_set_error(__VA_ARGS__);     \
      return (code);                  \
    }                                 \
  } while (0)

#define TPG_TRY(call)          \
  do {                         \
    int _rc = (call);          \
    if (_rc != TPG_OK) return _rc; \
  } while (0)

// ---------------------------------------------------------------------------
struct ProfRec {
  std::string name;
  hipEvent_t start, stop;
};

struct tpg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool prof = false;
  std::set<std::string> prof_only;  // empty: every launch is timed
  std::vector<ProfRec> prof_pending;
  std::map<std::string, std::pair<double, int64_t>> prof_acc;  // name -> (ms, launches)
  std::vector<hipEvent_t> event_pool;
  int num_cu = 256;
  int pool_id = 0;  // this context's device-memory pool (runtime.hip): blocks are reused in the order of ITS stream
  void* resident = nullptr;  // pairwise.hip: accumulators kept across increment_* calls
  // pca.hip, digit-split Gram kernel: fractional bits the per-locus weights keep (22: within 2^-23 of the weight, what a
  // whole-panel Gram matrix averages down to 1e-10; a streamed run (stream.hip) adds up many short blocks and asks for 30)
  int pca_digit_fbits = 22;
  // small host -> device copies without a stream synchronisation: a ring of pinned slots (runtime.hip: tpg_h2d_async)
  static constexpr int H2D_SLOTS = 8;
  static constexpr size_t H2D_SLOT_BYTES = 256u << 10;
  uint8_t* h2d_pinned = nullptr;
  hipEvent_t h2d_done[H2D_SLOTS] = {};
  bool h2d_used[H2D_SLOTS] = {};
  int h2d_next = 0;
  // Small results and small inputs WITHOUT the runtime's copy engines or a stream synchronisation (runtime.hip:
  // tpg_fetch_small / tpg_push_small): a one-workgroup kernel copies between device memory and a coherent pinned "mailbox"
  // and raises a sequence number there; the host polls that word.  A hipMemcpyAsync + hipStreamSynchronize round trip
  // costs ~90 us of idle GPU on these boxes (tools/gaps.py), the mailbox ~10.
  static constexpr size_t MAIL_FETCH_BYTES = 64u << 10;   // one result at a time
  static constexpr size_t MAIL_PUSH_BYTES = 512u << 10;   // ring of inputs in flight
  static constexpr size_t MAIL_PUSH_MAX = 64u << 10;      // largest single input
  uint8_t* mail_host = nullptr;  // [flags: 64 B][fetch area][push ring]
  uint8_t* mail_dev = nullptr;   // the same memory as the device sees it
  uint32_t mail_fetch_seq = 0;
  uint32_t mail_push_seq = 0;    // sequence number of the last push
  size_t mail_push_off = 0;      // next free byte of the ring
  uint32_t mail_push_wrap_seq = 0;  // last push of the previous lap: must have been read before the ring is written again
};

struct ProfScope {
  tpg_ctx* ctx;
  ProfRec rec;
  bool on;
  ProfScope(tpg_ctx* c, const char* name);
  ~ProfScope();
};
int tpg_prof_resolve(tpg_ctx* ctx);
// gramcls.hip: S' = sum_j w_j g_i g_k' (n x n, column-major, both triangles) into d_K by weight classes; *done = false
// (nothing written) when the weights take too many distinct values for that to pay.  d_what (may be NULL): the weight
// actually used for every locus (its class representative, within 2^-47 of w_j)
// centred_ok: the caller applies the double centring H K H afterwards, so terms r_i + r_k + const may be left out of S'
// (gramcls.hip: the centred operand layout of the mixed-fold kernel)
int tpg_gram_classes(tpg_ctx* ctx, const struct tpg_view* v, const double* d_w, double* d_what, double* d_K, bool* done,
                     bool centred_ok);
// The same over the ranks of a communicator, with whole weight classes per rank: `v` holds this rank's loci (counts: its
// m x 4 genotype counts, scale: its binomial scales), the packed columns are exchanged so that rank r ends up with ALL loci
// of the allele-count range it owns, and d_K receives S' of those classes (the sum over the ranks is S' of the panel).
// *done = false (on every rank alike, nothing exchanged) when that would not pay.
int tpg_gram_classes_exchanged(tpg_ctx* ctx, struct tpg_comm* comm, const struct tpg_view* v, const int32_t* d_counts,
                               const double* d_scale, double* d_K, bool* done);

// Launch a kernel on the context's stream, bracketed by HIP events when profiling is on.
#define TPG_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                          \
  do {                                                                                  \
    ProfScope _ps((ctx), (name));                                                       \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);        \
  } while (0)

#define TPG_CHECK_LAUNCH() TPG_HIP(hipGetLastError())

// ---------------------------------------------------------------------------
// HBM layout of a packed view (see DESIGN.md "Data layout").
//
// A *fragment block* is 1 KiB = 64 lanes x 16 B.  Lane l = (r = l & 31, h = l >> 5)
// holds 4 dwords s = 0..3; dword s packs 16 two-bit codes, element e = 4k + b stored at
// bits [8b + 2k, 8b + 2k + 1], so that (w >> 2k) & 0x03030303 is the 4 code bytes of MFMA
// operand register k.  One dword is exactly the 16-deep K slice that lane (r, h) feeds to
// v_mfma_i32_32x32x32_i8 for row/column r.
//
//   T ("individual-tiled", contraction over loci: pairwise N x N, PCA Gram, Z.V):
//     block (rt, kg): lane (r,h), dword s, element e  <->  individual 32 rt + r,
//     locus 128 kg + 32 s + 16 h + e.          address: ((rt * KG + kg) * 64 + lane) * 16 B
//   L ("locus-tiled", contraction over individuals: per-locus / per-group counts, Z'u):
//     block (lt, q):  lane (r,h), dword s, element e  <->  locus 32 lt + r,
//     individual 128 q + 32 s + 16 h + e.      address: ((lt * Q + q) * 64 + lane) * 16 B
//
// Codes: 0,1,2 = alt-allele dosage, 3 = missing.  Padding (individuals >= n, loci >= m) is 3.
struct tpg_fbm {
  tpg_ctx* ctx;
  uint8_t* d_bytes;
  int64_t nrow, ncol;
  int64_t bed_bpl = 0;  // > 0: d_bytes is a PLINK .bed payload with this many bytes per SNP (4 genotypes per byte)
  bool pooled = false;  // d_bytes came from the context's pool (the per-block uploads of the increment_* mirrors)
};

// class-wise counts via MFMA: cls[n] in [0, nclass); cnt[3][Mpad][Cpad] (het, hom-alt, valid)
struct GroupedCounts {
  int32_t* cnt = nullptr;
  int64_t Mpad = 0;  // 32 * n_lt
  int Cpad = 0;      // 32 * ceil(nclass/32)
  int nclass = 0;
  bool borrowed = false;  // a view's cached counts handed out to a caller: not freed by the borrower
  ~GroupedCounts();
};

struct tpg_view {
  tpg_ctx* ctx;
  int64_t n, m;    // kept individuals / loci
  int64_t Q, KG;   // ceil(n/128), ceil(m/128)
  mutable uint4* T;  // (4Q) row tiles x KG blocks; NULL in the views of tpg_view_create_pair until somebody needs it
                     // (tpg_view_need_T builds it from L)
  uint4* L;        // (4KG) locus tiles x Q blocks
  size_t bytes_each;
  // T re-coded as FP4 (E2M1) operand nibbles for the pairwise kernel (pairwise.hip); written by the pack kernel for
  // the first view of tpg_view_create_pair, otherwise made from T on the first tpg_pairwise_accumulate; 2 x bytes_each
  mutable uint4* T4 = nullptr;
  // the last per-class counts computed on this view (grouped_alt_freq, grouped_summaries and the Fst
  // methods of one analysis all use the same grouping): reused while the class vector is unchanged
  mutable GroupedCounts gc_cache;
  mutable std::vector<int32_t> gc_cls;
  // per-locus genotype counts as the fast pack kernel left them (it has every code in registers anyway): for each chunk of
  // 256 individuals and each locus one dword {codes with bit 0 set, with bit 1 set, with both} in fields of 10 bits,
  // lc_part[chunk * lc_row + locus]; tpg_launch_loci_counts adds the chunks up instead of reading the L layout again
  uint32_t* lc_part = nullptr;
  int lc_chunks = 0;
  int64_t lc_row = 0;
};

// tile-packed int32 accumulators of the pairwise kernel: per unit (super-tile I of TPG_PW_TA row tiles, 32-column
// tile jt >= TA I) 5 products x TA sub-tiles x 16 accumulator registers x 64 lanes (MFMA C/D order)
#define TPG_PW_PRODUCTS 5  // V, D, H, HV (= A[i][j]), VH (= A[j][i])
#define TPG_PW_TA 3        // A row tiles per wave: 96 x 32 pairs, 15 accumulator tiles (240 AGPRs); see pairwise.hip
#define TPG_PW_PLANE_INTS (TPG_PW_TA * 16 * 64)
#define TPG_PW_TILE_INTS (TPG_PW_PRODUCTS * TPG_PW_PLANE_INTS)
struct tpg_pairwise {
  tpg_ctx* ctx;
  int64_t n;
  int64_t nst;  // super-tiles of 32 TPG_PW_TA individuals
  int64_t ntp;  // number of (I, jt) unit slots = TA nst (nst + 1) / 2
  int32_t* acc;
  bool owns;
  void* order;  // device int2[nun]: the units (I, jt) that hold data, in XCD patch order (pairwise.hip)
  int64_t nun;
  int64_t loci;          // loci accumulated since the last zero (all ranks' loci after a reduction): overflow guard
  int64_t as_pad_quirk;  // reference quirk Q1: added to every allele-sharing numerator (tpg_pairwise_set_as_pad_quirk)
  // Sharding over the ranks of a tpg_comm (comm.hip).  The slab of unit (I, jt) sits at tpg_pw_unit_index + rowpad[I]:
  // super-tile rows are dealt to the ranks in contiguous bands of (nearly) equal unit counts, every band padded to
  // `chunk_units` slabs, so that ONE reduce-scatter with equal counts leaves rank r with the sums of band r.  A
  // single rank has one band, rowpad = 0 and chunk_units = ntp: the unsharded layout.
  int nranks = 1, rank = 0;
  int64_t chunk_units = 0;
  std::vector<int32_t> band;   // nranks + 1 super-tile boundaries: band r = rows [band[r], band[r + 1])
  int64_t* rowpad = nullptr;   // device int64[nst]
  bool reduced = false;        // after tpg_pairwise_reduce: only this rank's band holds (complete) sums
  // tpg_pairwise_reduce_begin / _end (the reduce-scatter on a second communicator's stream)
  bool reducing = false;
  hipEvent_t ev_acc = nullptr, ev_red = nullptr;
  int64_t pending_loci = 0;
  int pending_lack = 0;
  // products (TPG_PW_V | D | H | A) every accumulate since the last zero has added: what the epilogues may read
  int have = 31;  // TPG_PW_HAVE_ALL
  // unit tables of the product-subset kernels (pairwise.hip), one per wave-tile shape: key 16 RA + RB -> (device int2[], count)
  std::map<int, std::pair<void*, int64_t>> orders;
};
#define TPG_PW_MAX_LOCI 2147483647ll
#define TPG_PW_HAVE_ALL (TPG_PW_ALL | TPG_PW_DH)  // tpg_pairwise.have after a zero: every product, and with D and H their sum
// the smallest product set that serves the outputs asked for (pairwise.hip kernels: {V, D+H}, {V, D}, {V, D, H}, {V, D, A}, all five)
static inline int tpg_pw_products_for(bool ibs, bool king, bool as_or_grm) {
  if (ibs && !king && !as_or_grm) return TPG_PW_FOR_IBS_ALONE;
  return (ibs ? TPG_PW_FOR_IBS : 0) | (king ? TPG_PW_FOR_KING : 0) | (as_or_grm ? TPG_PW_FOR_AS : 0);
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Size-bucketed cache of device allocations (hipMalloc / hipFree cost milliseconds and synchronise
// the device; a step of the hot path needs ~40 scratch buffers).  One pool PER CONTEXT: a context issues
// all its work on one stream, so a block that goes back to its pool and out again is reused in stream
// order; a block never crosses to another context (another device, or another stream of the same device).
// tpg_pmalloc() serves the context the calling thread entered last (TpgEnter, first statement of every C-ABI
// entry point, which also makes that context's device current); tpg_pfree() returns a block to the pool it
// came from, or to hipFree() when that context is gone or the pointer is not the pool's.
hipError_t tpg_pmalloc(void** p, size_t bytes);
void tpg_pfree(void* p);
void tpg_pool_trim(int pool_id);  // release the cached (free) blocks of one pool
struct TpgEnter {
  tpg_ctx* prev;
  explicit TpgEnter(tpg_ctx* ctx);
  ~TpgEnter();
};
tpg_ctx* tpg_current_ctx();

// host -> device on the context's stream.  Up to a slot's size the source is copied into pinned memory first, so the
// caller's buffer is free at return and nothing waits for the stream; larger copies are waited for.
hipError_t tpg_h2d_async(tpg_ctx* ctx, void* dst, const void* src, size_t bytes);
// device -> host for small results (a multiple of 4 bytes, at most MAIL_FETCH_BYTES; anything else takes the copy engine):
// returns when the bytes are in host_dst, which is when everything enqueued before has run -- without draining the stream
// through the runtime
hipError_t tpg_fetch_small(tpg_ctx* ctx, void* host_dst, const void* d_src, size_t bytes);
// the other direction (the caller's buffer is free at return, nothing is waited for)
hipError_t tpg_push_small(tpg_ctx* ctx, void* d_dst, const void* host_src, size_t bytes);
// device -> device on the stream by a kernel (a hipMemcpyAsync between two kernels costs 15 - 30 us of idle GPU)
hipError_t tpg_copy_dev(tpg_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
// bulk transfers (waited for): large ones are chunked through pinned slots with a team of copying threads
hipError_t tpg_upload(tpg_ctx* ctx, void* dst, const void* src, size_t bytes);
hipError_t tpg_download(tpg_ctx* ctx, void* dst, const void* src, size_t bytes);
// host FBM bytes that will be read through ONE code table: as 2 bits per genotype where table and bytes allow it (runtime.hip)
int tpg_fbm_from_host_for_table(tpg_ctx* ctx, const uint8_t* bytes, int64_t nrow, int64_t ncol, const double* code256, tpg_fbm** out,
                                const double** view_table);
// FBM bytes -> a .bed payload in HBM, packed to 2 bits per genotype on the host through lut16 (runtime.hip); *ok = false: send bytes instead
hipError_t tpg_upload_bedpacked(tpg_ctx* ctx, uint8_t* dst, const uint8_t* src, int64_t nrow, int64_t ncols, const uint8_t* lut16, bool* ok);
// `height` contiguous device pieces of `width` bytes -> host pieces `dpitch` bytes apart (rows of a column-major host matrix)
hipError_t tpg_download_rows(tpg_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t width, size_t height);
// pinned staging buffers (256 MiB each) the process keeps between transfers: at least `buffers` from now on (freeing one
// synchronises the device, so a caller that runs several transfers at once -- tpg_multi's device threads -- raises it first)
void tpg_stage_keep(int buffers);

// device allocation helpers
template <typename T>
static inline int tpg_dmalloc(T** p, size_t count) {
  TPG_HIP(hipMalloc((void**)p, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
  return TPG_OK;
}

// Output buffer that may be host or device memory: kernels write to dev(); commit() copies back
// if the user's pointer is not device memory.
bool tpg_is_device_ptr(const void* p);
struct OutBuf {
  void* user = nullptr;
  void* d = nullptr;
  size_t bytes = 0;
  bool owned = false;
  int init(void* user_ptr, size_t nbytes);
  int commit(tpg_ctx* ctx);  // async copy on ctx stream (+ sync) when user is host memory
  ~OutBuf();
  template <typename T> T* dev() { return (T*)d; }
};
// Input buffer: host or device pointer -> device pointer
struct InBuf {
  const void* d = nullptr;
  void* owned_ptr = nullptr;
  int init(tpg_ctx* ctx, const void* user_ptr, size_t nbytes);
  ~InBuf();
  template <typename T> const T* dev() { return (const T*)d; }
};

void tpg_resident_release(tpg_ctx* ctx);

// collectives over the ranks of a communicator (comm.hip).  One rank = one context.  Transport: RCCL (production:
// one process per GPU, or one process driving several GPUs), or a caller-supplied all-reduce through host memory
// (rehearsals of the sharded paths without RCCL: tests with gloo, several ranks sharing one GPU).
struct tpg_comm {
  tpg_ctx* ctx = nullptr;
  int nranks = 1, rank = 0;
  void* nccl = nullptr;  // ncclComm_t
  int (*host_fn)(void* user, void* buf, int64_t count, int dtype) = nullptr;  // in-place sum; dtype 0 int32, 1 float64
  void* host_user = nullptr;
  int32_t* d_status = nullptr;  // device int32[TPG_COMM_STATUS_INTS]: the status word of tpg_comm_agree
  int a2a_state = 0;            // tpg_comm_alltoall_usable: 0 not tried yet, 1 works, -1 does not (on every rank alike)
};
#define TPG_COMM_STATUS_INTS 8
// pca.hip: the k largest eigenpairs of the symmetric K (device memory) at relative residual tol; lambda on the host, U (n x k) in device memory
int tpg_sym_eig_topk_tol(tpg_ctx* ctx, const double* d_K, int64_t n, int k, double tol, double* lambda_host, double* d_U);
int tpg_pca_gram_allreduce(tpg_ctx* ctx, tpg_comm* comm, double* d_K, int64_t n);  // pca.hip: K <- sum over the ranks (its triangle travels)
// stream.hip: a tpg_multi_* analysis of a host FBM with every device streaming its share in blocks under `budget` bytes
int tpg_multi_stream_host(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, size_t budget, const tpg_stream_job* job);
int tpg_comm_agree(tpg_comm* comm, int rc);  // all ranks get the same status (the worst any of them passed in)
int tpg_comm_reduce_scatter_i32(tpg_comm* comm, int32_t* d_buf, int64_t chunk_count);  // in place, chunk r -> rank r
int tpg_comm_allreduce(tpg_comm* comm, void* d_buf, int64_t count, int dtype);          // in place, device memory
// 8-byte words: rank r sends scnt[d] words at soff[d] to rank d, receives rcnt[s] words from rank s at roff[s]
// a tiny all-to-all with known contents, once per communicator: true on every rank or on none.  The class exchange of the
// PCA Gram (gramcls.hip) is only taken when it is.
bool tpg_comm_alltoall_usable(tpg_comm* comm);
int tpg_comm_alltoallv64(tpg_comm* comm, const void* d_send, const size_t* scnt, const size_t* soff, void* d_recv,
                         const size_t* rcnt, const size_t* roff);

// ---- cross-TU entry points (one per .hip file) -----------------------------
int tpg_launch_pack(tpg_ctx* ctx, const tpg_fbm* fbm, const int32_t* d_rows, const int32_t* d_cols,
                    const uint8_t* d_lut, tpg_view* v, tpg_view* v2);
int tpg_launch_unpack(tpg_ctx* ctx, const tpg_view* v, uint8_t* d_codes, int from_L);
// the T layout of a view that was packed without it (from L; a no-op when it is there)
int tpg_view_need_T(tpg_ctx* ctx, const tpg_view* v);
int tpg_launch_synth(tpg_ctx* ctx, uint8_t* d_bytes, uint64_t seed, int64_t nrow, int64_t ncol, int64_t j0,
                     int npop, uint32_t miss_thresh, int imputed_bytes);

// per-locus
int tpg_launch_loci_counts(tpg_ctx* ctx, const tpg_view* v, int32_t* d_counts /* m x 4 */);
int tpg_grouped_counts(tpg_ctx* ctx, const tpg_view* v, const int32_t* h_cls, int nclass, GroupedCounts* out);
