// stream.hip -- the reference's block loop INSIDE the library: whole analyses of a genotype store that stays on the host.
//
// The reference's defining property on this path is that the FBM is a file and is swept in blocks of loci
// (R/snp_ibs.R:59-82 for the pairwise matrices, R/loci_alt_freq.R:351-359 for the per-locus statistics,
// bigstatsr::big_SVD behind R/gt_pca_partialSVD.R:82-89 for the Gram matrix and, a second time, for the loadings).
// tpg_stream_run is that loop as a three-stage pipeline on one GPU:
//
//   uploader thread (own context = own stream)   block b + 1 of the store -> one of two block buffers in HBM
//   calling thread  (the caller's context)       block b: pack views, per-locus statistics, Fst sums, pairwise
//                                                cross-products += , Gram matrix += (all additive over loci)
//   downloader thread (own context = own stream) per-locus results of block b - 1 -> rows of the caller's arrays
//
// Resident at any time: two blocks of store bytes, the views and scratch of the block at hand, two sets of per-block
// outputs, and the additive state (pairwise slabs, Gram matrix, Fst sums).  The block width follows from the caller's
// budget.  After the sweep: epilogues (N x N results go down beside the eigen step), eigen step, and the loadings -- from
// the imputed views if the budget let them stay, from a second sweep over the store otherwise.
// tpg_multi_stream_run runs the same loop on every device of a tpg_multi over its share of colInd and exchanges what is
// additive.
//
// Everything here is orchestration over the library's own entry points: no kernel of the hot path lives in this file.
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "common.h"

namespace {

enum SrcKind { SRC_BYTES = 0, SRC_BED = 1, SRC_SYNTH = 2 };

struct StreamSource {
  SrcKind kind = SRC_BYTES;
  const uint8_t* bytes = nullptr;  // column-major FBM bytes, or the .bed payload
  int64_t nrow = 0, ncol = 0;
  int64_t bpl = 0;  // .bed: bytes per SNP
  uint64_t seed = 0;
  int npop = 1;
  uint32_t miss = 0;
  int imputed = 0;
  size_t unit() const { return kind == SRC_BED ? (size_t)bpl : (size_t)nrow; }  // store bytes per locus
};

}  // namespace

struct tpg_stream {
  tpg_ctx* ctx = nullptr;
  StreamSource src;
  size_t budget = 0;
  void* map_base = nullptr;  // mapping owned by the stream (tpg_stream_open_bk / _bed)
  size_t map_len = 0;
  tpg_ctx *up_ctx = nullptr, *down_ctx = nullptr;  // the worker threads' contexts (streams), made on the first run
};

namespace {

// byte -> 2-bit code of a code256 (runtime.hip: make_lut), to find out which of the job's tables are the same view
static void lut_of(const double* code256, uint8_t* lut) {
  for (int b = 0; b < 256; b++) {
    if (!code256) { lut[b] = b < 3 ? (uint8_t)b : 3; continue; }
    const double x = code256[b];
    lut[b] = !(x > -1) ? 3 : x == 0.0 ? 0 : x == 1.0 ? 1 : x == 2.0 ? 2 : 0xFF;
  }
}

// ---- shared state of one run's threads: first error wins, everybody else stops at its next wait ----
struct Shared {
  std::mutex mu;
  std::condition_variable cv;
  bool failed = false;
  int code = TPG_OK;
  std::string msg;
  void fail(int rc, const char* what) {
    std::lock_guard<std::mutex> lk(mu);
    if (!failed) { failed = true; code = rc; msg = what ? what : ""; }
    cv.notify_all();
  }
};

struct DownTask {
  hipEvent_t ready = nullptr;  // recorded on the producer's stream behind the kernels that wrote src (may be NULL: already complete)
  const void* src = nullptr;
  void* dst = nullptr;
  size_t width = 0, height = 0, dpitch = 0;
  int slot = -1;  // per-block output slot this read belongs to (-1: none)
};

__global__ void tpg_stream_add_kernel(double* __restrict__ y, const double* __restrict__ x, int64_t count) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) y[i] += x[i];
}

struct DevBuf {  // pool block, back to its pool at scope exit (the pool is the context's the calling thread entered)
  void* p = nullptr;
  int alloc(size_t bytes) {
    free();
    TPG_HIP(tpg_pmalloc(&p, bytes ? bytes : 16));
    return TPG_OK;
  }
  void free() { if (p) tpg_pfree(p); p = nullptr; }
  ~DevBuf() { free(); }
  template <typename T> T* as() const { return (T*)p; }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

static size_t device_used() {
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return tot - fr;
}

// One device's streamed run over positions [P0, P1) of the job's colInd.
struct StreamRun {
  tpg_ctx* ctx = nullptr;
  tpg_comm* comm = nullptr;  // NULL: a single device
  const StreamSource* src = nullptr;
  const tpg_stream_job* job = nullptr;
  size_t budget = 0;
  int64_t n = 0, m = 0;      // the selection (m = all loci of the job, not only this device's)
  int64_t P0 = 0, P1 = 0;    // this device's positions of colInd
  tpg_ctx *up_ctx = nullptr, *down_ctx = nullptr;
  bool own_workers = true;  // the worker contexts are this run's (tpg_multi_stream_run) or the stream object's
  Shared sh;

  // what the job needs
  bool want_pw = false, want_loc = false, want_fst = false, want_pca = false;
  int products = 0;
  const double *code_pw = nullptr, *code_loc = nullptr, *code_pca = nullptr;
  int view_of_pw = -1, view_of_loc = -1, view_of_pca = -1;  // index into the (deduplicated) tables of a block, -1 unused
  int ntab = 0;
  const double* tab[3] = {nullptr, nullptr, nullptr};

  // plan
  int64_t B = 0, nblocks = 0;
  bool keep_views = false;
  size_t planned = 0, state = 0;
  std::vector<uint8_t> contiguous;  // per block: colInd runs up by one over the block

  // block buffers of the store
  uint8_t* d_blk[2] = {nullptr, nullptr};
  // A byte store whose blocks are wanted through ONE code table goes up as 2 bits per genotype, packed on the host into the
  // layout of a .bed payload (host_bedpack.h: a quarter of the bytes over PCIe, half of the nibble pack's), and the device
  // packs its views with the .bed front end.  bedpack_lut: byte < 16 -> .bed code (valid = false: not this table); a block in
  // which a byte >= 16 turns up goes as bytes (blk_bed[slot] tells the main thread which it got).
  uint8_t bedpack_lut[16] = {};
  bool bedpack = false;
  bool blk_bed[2] = {false, false};
  // uploader <-> main
  std::deque<int64_t> ready_q;     // blocks uploaded, in order
  int64_t released[2] = {-1, -1};  // slot k may be overwritten for block b when released[k] >= b - 2
  // per-block outputs (two slots) and the downloader
  struct OutSlot {
    DevBuf af, gaf, gm, lc, fl[TPG_STREAM_MAX_FST], fd[TPG_STREAM_MAX_FST], dc, ds, dv;
    hipEvent_t ev = nullptr, ev2 = nullptr;  // behind the per-locus kernels / behind the PCA's center and scale of the block
    int pending = 0;  // download tasks of this slot not finished yet
  } out[2];
  std::deque<DownTask> down_q;
  bool down_stop = false;
  size_t bytes_down = 0, bytes_up = 0;
  std::thread up_th, down_th;
  bool down_running = false;

  // additive state
  tpg_pairwise* pw = nullptr;
  DevBuf d_K, d_fsum, d_fpart;
  bool have_K = false;
  double fro = 0.0;
  struct Kept { tpg_view* v = nullptr; DevBuf dc, ds; int64_t q0 = 0, mb = 0; bool L_borrowed = false; };
  std::deque<Kept> kept;
  // The PCA Gram of a run that keeps its imputed views (no budget) is NOT taken block by block: the class path pays for every
  // weight class a block holds (whole 64-locus MFMA blocks + a fold), and a block of an eighth of the panel still holds nearly
  // all of them -- 8.1 ms for the panel in one piece, 29.3 in eight (profiles/r06_stream_kernels_vs_blocks.txt).  The kept
  // views' L layouts are laid end to end in ONE buffer (L is locus-tile major: appending loci appends memory) and the Gram runs
  // ONCE, after the last block -- for a store whose run is paced by the kernels (.bed payload, synthetic): 70.9 -> 64.4 ms for
  // the bench panel from a .bed, with 4 - 8 blocks instead of 2.  A byte store is paced by its uploads: its blocks' Gram
  // matrices hide behind them, and batching (everything but the last two blocks while those are on their way, then each of
  // them alone) measured 113.7 ms against 108.6 block by block -- so it keeps the Gram per block.
  bool batch_gram = false;
  bool big_views = false;  // the kept views' L layouts end to end in bigL (every run without a budget that keeps several views):
                           // the loadings are then ONE call and ONE download (eight pieces of 19 MB took 7 ms, one of 160 MB 3)
  DevBuf bigL, big_c, big_s, d_vbig;
  int64_t gram_from = 0;  // loci (positions from P0) whose Gram is in K
  void free_kept(Kept& kp) {
    if (kp.v && kp.L_borrowed) kp.v->L = nullptr;  // it points into bigL
    tpg_view_free(kp.v);
    kp.v = nullptr;
  }
  DevBuf d_u, d_nn[4];
  hipEvent_t ev_fin[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<double> dh;

  size_t base_used = 0, peak_used = 0;
  double t_start = 0, t_sweep1 = 0;
  int saved_fbits = 0;

  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  // TPG_STREAM_TRACE=1: a timeline of the run on stderr (milliseconds since its start; who waited for whom)
  bool trace = getenv("TPG_STREAM_TRACE") != nullptr;  // (read per run)
  void stamp(const char* what, long long b = -1) const {
    if (trace) fprintf(stderr, "[stream] %8.2f ms  %s %lld\n", 1e3 * (now() - t_start), what, b);
  }

  ~StreamRun() { cleanup(); }

  // ------------------------------------------------------------------ set-up
  int setup() {
    const tpg_stream_job* j = job;
    want_pw = j->ibs || j->king || j->allele_sharing || j->grm;
    want_loc = j->alt_freq || j->grouped_alt_freq || j->grouped_missingness || j->loci_counts;
    want_fst = j->nfst > 0;
    want_pca = j->k > 0;
    TPG_REQUIRE(want_pw || want_loc || want_fst || want_pca, TPG_EINVAL, "the job asks for nothing");
    products = tpg_pw_products_for(j->ibs != nullptr, j->king != nullptr, j->allele_sharing || j->grm);
    if (want_fst) {
      TPG_REQUIRE(j->nfst <= TPG_STREAM_MAX_FST && j->pairs1 && j->P > 0, TPG_EINVAL, "bad Fst request");
      for (int i = 0; i < j->nfst; i++) {
        TPG_REQUIRE(j->fst_method[i] >= 0 && j->fst_method[i] <= 2, TPG_EINVAL, "bad Fst method %d", j->fst_method[i]);
        TPG_REQUIRE(j->fst_tot[i] || j->fst_by_locus[i], TPG_EINVAL, "Fst method %d has no output", i);
        TPG_REQUIRE(!j->fst_return_num_dem || !j->fst_by_locus[i] || j->fst_by_locus_den[i], TPG_EINVAL,
                    "return_num_dem needs fst_by_locus_den for method %d", i);
      }
    }
    if (j->grouped_alt_freq || j->grouped_missingness || want_fst)
      TPG_REQUIRE(j->groupIds0 && j->ngroups > 0, TPG_EINVAL, "grouped outputs need groupIds and ngroups");
    if (want_pca) {
      TPG_REQUIRE(j->d && j->u && j->v && j->center && j->scale, TPG_EINVAL, "the PCA needs d, u, v, center and scale");
      TPG_REQUIRE(j->k <= n && j->k <= m, TPG_EINVAL, "k = %d out of range", j->k);
      TPG_REQUIRE(j->pca_tol >= 0 && j->pca_tol < 1, TPG_EINVAL, "pca_tol = %g out of [0, 1)", j->pca_tol);
    }
    // which views a block needs: tables that map every byte alike are one view
    code_pw = nullptr;
    code_loc = j->code256;
    code_pca = j->code256_pca;
    uint8_t luts[3][256];
    auto add = [&](const double* code) -> int {
      uint8_t l[256];
      lut_of(code, l);
      for (int t = 0; t < ntab; t++)
        if (memcmp(luts[t], l, 256) == 0) return t;
      memcpy(luts[ntab], l, 256);
      tab[ntab] = code;
      return ntab++;
    };
    if (want_pw) view_of_pw = add(code_pw);  // first: the view of a pair that gets the FP4 operand layout
    if (want_loc || want_fst) view_of_loc = add(code_loc);
    if (want_pca) view_of_pca = add(code_pca);
    return TPG_OK;
  }

  // ------------------------------------------------------------------ the block plan
  int plan() {
    const tpg_stream_job* j = job;
    const int64_t mloc = P1 - P0;
    const size_t npad = (size_t)ceil_div(n, 128) * 128;
    const int G = j->ngroups;
    // bytes per locus of a block in flight
    size_t per = 2 * src->unit();  // the two block buffers
    for (int t = 0; t < ntab; t++) {
      per += npad / 4;                                                   // L
      if (t == view_of_pw) per += npad / 2;                              // T4 (made by the pack or on first use)
      else if (ntab == 1 || (ntab == 3 && t == view_of_loc)) per += npad / 4;  // a single view through a table carries T as well
    }
    if (want_loc || want_fst) per += 16 + (j->groupIds0 ? 12 * 32 * (size_t)ceil_div(G + 1, 32) : 0);  // counts; class counts (cached by the view)
    size_t outp = 0;  // per-block outputs, two slots
    if (j->alt_freq) outp += 16;
    if (j->grouped_alt_freq) outp += 16 * (size_t)G;
    if (j->grouped_missingness) outp += 8 * (size_t)G;
    if (j->loci_counts) outp += 16;
    for (int i = 0; i < j->nfst; i++)
      if (j->fst_by_locus[i]) outp += 8 * (size_t)j->P * (j->fst_return_num_dem ? 2 : 1);
    if (want_pca) outp += 16 + 8 * (size_t)j->k;
    per += 2 * outp;
    if (want_pca) per += 16 + 24 + npad / 4 + (npad / 4) * 5 / 4;  // counts, weights, the class path's locus-major copy + sorted operands
    per += per / 8;                                                  // slack: row / column index vectors, padding to whole tiles, pool rounding
    const size_t keep_all = want_pca ? (size_t)mloc * (npad / 4 + 16) : 0;  // imputed L views + center / scale for the loadings
    // additive state: not part of the budget
    state = 0;
    if (want_pw) state += tpg_pairwise_buffer_bytes(n) + 8 * (size_t)n * (size_t)n * (size_t)((j->ibs ? 1 : 0) + (j->king ? 1 : 0) + (j->allele_sharing ? 1 : 0) + (j->grm ? 1 : 0));
    if (want_pca) state += 3 * 8 * (size_t)n * (size_t)n + 8 * (size_t)n * 64 * 6;  // K, a block's Gram, the class path's slabs; eigen blocks
    if (want_fst) state += 8 * 4 * (size_t)j->P * (size_t)j->nfst;
    int64_t target_blocks;
    if (budget == 0) {
      // no bound: the pipeline's own optimum -- a handful of blocks (per-block fixed costs against overlap: 8 blocks of a
      // byte store, 4 of a .bed payload measured best at 5 000 x 1 000 000), views kept
      target_blocks = src->kind == SRC_BED ? 4 : 8;
      if (const char* e = getenv("TPG_STREAM_BLOCKS")) target_blocks = std::max(1, atoi(e));
      B = ceil_div(ceil_div(mloc, target_blocks), 128) * 128;
      keep_views = want_pca;
    } else {
      keep_views = want_pca && keep_all <= budget / 2;
      const size_t avail = budget - (keep_views ? keep_all : 0);
      B = (int64_t)(avail / per) / 128 * 128;
      TPG_REQUIRE(B >= 128, TPG_EINVAL, "budget of %zu bytes is below one block of 128 loci (%zu bytes per locus in flight)", budget, per);
    }
    B = std::min<int64_t>(B, ceil_div(mloc, 128) * 128);
    if (B < 128) B = 128;
    nblocks = ceil_div(mloc, B);
    planned = (size_t)std::min<int64_t>(B, mloc) * per + (keep_views ? keep_all : 0);
    contiguous.assign((size_t)nblocks, 1);
    if (j->colInd1)
      for (int64_t b = 0; b < nblocks; b++) {
        const int64_t q0 = P0 + b * B, q1 = std::min(P1, q0 + B);
        for (int64_t q = q0 + 1; q < q1; q++)
          if (j->colInd1[q] != j->colInd1[q - 1] + 1) { contiguous[(size_t)b] = 0; break; }
      }
    if (src->kind == SRC_SYNTH)
      for (int64_t b = 0; b < nblocks; b++)
        TPG_REQUIRE(contiguous[(size_t)b], TPG_EUNSUPPORTED, "a synthetic store is generated in runs of consecutive loci: colInd must be contiguous");
    return TPG_OK;
  }

  // the 2-bit host pack serves a byte store and ONE table whose every entry below 16 is a code (TPG_STREAM_BEDPACK=0: off)
  void set_bedpack(const double* table) {
    bedpack = false;
    const char* e = getenv("TPG_STREAM_BEDPACK");  // (read per run: the tests switch it)
    if ((e && atoi(e) == 0) || src->kind != SRC_BYTES) return;
    uint8_t l[256];
    lut_of(table, l);
    static const uint8_t bedcode[4] = {3, 2, 0, 1};  // code 0, 1, 2, missing -> .bed 11, 10, 00, 01 (the bytes 0, 1, 2, 3 of bigsnpr's reading)
    for (int b = 0; b < 16; b++) {
      if (l[b] > 3) return;
      bedpack_lut[b] = bedcode[l[b]];
    }
    bedpack = true;
  }
  // the table a view of a block is made through: the job's for a block that arrived as bytes; for one that arrived packed the
  // codes are already those of the job's table, read back as bytes 0, 1, 2, 3 = NA (raw semantics where the job's were raw, so
  // that a lone pairwise view is packed with its FP4 layout)
  const double* block_table(int slot, const double* job_table) const {
    static const double* code012 = [] {
      static double t[256];
      for (int b = 0; b < 256; b++) t[b] = b < 3 ? (double)b : __builtin_nan("");
      return (const double*)t;
    }();
    return blk_bed[slot] ? (job_table ? code012 : nullptr) : job_table;
  }
  tpg_fbm block_fbm(int slot, int64_t mb) const {
    tpg_fbm f{ctx, d_blk[slot], src->nrow, mb};
    f.bed_bpl = src->kind == SRC_BED ? src->bpl : blk_bed[slot] ? (src->nrow + 3) / 4 : 0;
    return f;
  }

  void block_range(int64_t b, int64_t* q0, int64_t* q1) const {
    *q0 = P0 + b * B;
    *q1 = std::min(P1, *q0 + B);
  }
  int64_t first_col(int64_t q) const { return job->colInd1 ? (int64_t)job->colInd1[q] - 1 : q; }  // 0-based store column of position q

  // ------------------------------------------------------------------ uploader thread
  int fill(int slot, int64_t b, std::vector<uint8_t>& stage) {
    int64_t q0, q1;
    block_range(b, &q0, &q1);
    const int64_t nb = q1 - q0;
    const size_t unit = src->unit();
    tpg_fbm f{ctx, d_blk[slot], src->nrow, nb};
    f.bed_bpl = src->kind == SRC_BED ? src->bpl : 0;
    if (src->kind == SRC_SYNTH) {
      TPG_TRY(tpg_launch_synth(up_ctx, f.d_bytes, src->seed, src->nrow, nb, first_col(q0), src->npop, src->miss, src->imputed));
      TPG_HIP(hipStreamSynchronize(up_ctx->stream));
      return TPG_OK;
    }
    blk_bed[slot] = false;
    const uint8_t* host = nullptr;
    if (bedpack && contiguous[(size_t)b]) {
      bool ok = false;
      TPG_HIP(tpg_upload_bedpacked(up_ctx, d_blk[slot], src->bytes + (size_t)first_col(q0) * unit, src->nrow, nb, bedpack_lut, &ok));
      if (ok) {
        blk_bed[slot] = true;
        bytes_up += (size_t)nb * (size_t)((src->nrow + 3) / 4);
        return TPG_OK;
      }
    }
    if (contiguous[(size_t)b]) {
      host = src->bytes + (size_t)first_col(q0) * unit;
    } else {  // scattered columns: gathered on the host first (a column / a SNP is one contiguous piece of the store)
      stage.resize((size_t)nb * unit);
      const int nth = (int)std::min<int64_t>(8, std::max<int64_t>(1, (int64_t)(stage.size() >> 22)));
      std::vector<std::thread> th;
      for (int t = 0; t < nth; t++)
        th.emplace_back([&, t]() {
          for (int64_t q = q0 + nb * t / nth; q < q0 + nb * (t + 1) / nth; q++)
            memcpy(stage.data() + (size_t)(q - q0) * unit, src->bytes + (size_t)first_col(q) * unit, unit);
        });
      for (auto& t : th) t.join();
      host = stage.data();
    }
    bytes_up += (size_t)nb * unit;
    if (src->kind == SRC_BED) return tpg_fbm_upload_bed_snps(up_ctx, &f, host, 0, nb);
    return tpg_fbm_upload_cols(up_ctx, &f, host, 0, nb);
  }

  void uploader() {
    TpgEnter _enter(up_ctx);
    std::vector<uint8_t> stage;
    for (int64_t b = 0; b < nblocks; b++) {
      const int slot = (int)(b & 1);
      {
        std::unique_lock<std::mutex> lk(sh.mu);
        sh.cv.wait(lk, [&] { return sh.failed || released[slot] >= b - 2; });
        if (sh.failed) return;
      }
      stamp("  upload starts", b);
      const int rc = fill(slot, b, stage);
      if (rc != TPG_OK) { sh.fail(rc, tpg_last_error()); return; }
      stamp("  upload done", b);
      {
        std::lock_guard<std::mutex> lk(sh.mu);
        ready_q.push_back(b);
      }
      sh.cv.notify_all();
    }
  }

  int start_uploader() {
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      ready_q.clear();
      released[0] = released[1] = -1;  // both buffers are free (blocks 0 and 1 need released >= -2 / -1)
    }
    up_th = std::thread([this] { uploader(); });
    return TPG_OK;
  }
  int wait_block(int64_t b) {
    std::unique_lock<std::mutex> lk(sh.mu);
    sh.cv.wait(lk, [&] { return sh.failed || (!ready_q.empty() && ready_q.front() == b); });
    if (sh.failed) { tpg_set_error("%s", sh.msg.c_str()); return sh.code; }
    ready_q.pop_front();
    return TPG_OK;
  }
  void release_block(int64_t b) {
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      released[b & 1] = b;
    }
    sh.cv.notify_all();
  }

  // ------------------------------------------------------------------ downloader thread
  void downloader() {
    TpgEnter _enter(down_ctx);
    for (;;) {
      DownTask t;
      {
        std::unique_lock<std::mutex> lk(sh.mu);
        sh.cv.wait(lk, [&] { return sh.failed || down_stop || !down_q.empty(); });
        if (sh.failed) return;
        if (down_q.empty()) return;  // stop requested and nothing left
        t = down_q.front();
        down_q.pop_front();
      }
      hipError_t e = hipSuccess;
      if (t.ready) e = hipStreamWaitEvent(down_ctx->stream, t.ready, 0);
      stamp("    download starts, MB:", (long long)(t.width * t.height >> 20));
      if (e == hipSuccess) e = tpg_download_rows(down_ctx, t.dst, t.dpitch, t.src, t.width, t.height);
      stamp("    download done");
      if (e != hipSuccess) {
        char buf[160];
        snprintf(buf, sizeof(buf), "download of a block's results failed: %s", hipGetErrorString(e));
        sh.fail(TPG_EHIP, buf);
        return;
      }
      {
        std::lock_guard<std::mutex> lk(sh.mu);
        bytes_down += t.width * t.height;
        if (t.slot >= 0) out[t.slot].pending--;
      }
      sh.cv.notify_all();
    }
  }
  void start_downloader() {
    if (down_running) return;
    down_stop = false;
    down_th = std::thread([this] { downloader(); });
    down_running = true;
  }
  int join_downloader() {
    if (!down_running) return TPG_OK;
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      down_stop = true;
    }
    sh.cv.notify_all();
    down_th.join();
    down_running = false;
    if (sh.failed) { tpg_set_error("%s", sh.msg.c_str()); return sh.code; }
    return TPG_OK;
  }
  int wait_slot(int slot) {  // the downloader has finished with what block b - 2 left in this slot
    std::unique_lock<std::mutex> lk(sh.mu);
    sh.cv.wait(lk, [&] { return sh.failed || out[slot].pending == 0; });
    if (sh.failed) { tpg_set_error("%s", sh.msg.c_str()); return sh.code; }
    return TPG_OK;
  }

  // rows [q, q + mb) of the caller's column-major matrix with `rows_total` rows and `ncols` columns <- the block's mb x ncols
  // matrix in device memory.  Host destination: a task for the downloader (behind `ev`); device destination: a 2-D copy
  // on this stream.
  int rows_out(void* dst, size_t elem, int64_t rows_total, int64_t q, const void* d_src, int64_t mb, int64_t ncols, hipEvent_t ev,
               int slot) {
    if (!dst || mb <= 0 || ncols <= 0) return TPG_OK;
    uint8_t* to = (uint8_t*)dst + elem * (size_t)q;
    if (tpg_is_device_ptr(dst)) {
      TPG_HIP(hipMemcpy2DAsync(to, elem * (size_t)rows_total, d_src, elem * (size_t)mb, elem * (size_t)mb, (size_t)ncols,
                               hipMemcpyDeviceToDevice, ctx->stream));
      return TPG_OK;
    }
    start_downloader();
    DownTask t;
    t.ready = ev;
    t.src = d_src;
    t.dst = to;
    t.width = elem * (size_t)mb;
    t.height = (size_t)ncols;
    t.dpitch = elem * (size_t)rows_total;
    t.slot = slot;
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      if (slot >= 0) out[slot].pending++;
      down_q.push_back(t);
    }
    sh.cv.notify_all();
    return TPG_OK;
  }

  void sample() {
    const size_t u = device_used();
    if (u > peak_used) peak_used = u;
  }

  // ------------------------------------------------------------------ the views of a block
  int make_views(const tpg_fbm* f, int slot, tpg_view** v /* [3] */) {
    const tpg_stream_job* j = job;
    v[0] = v[1] = v[2] = nullptr;
    if (ntab == 1) return tpg_view_create(ctx, f, j->rowInd1, n, nullptr, 0, block_table(slot, tab[0]), &v[0]);
    // (two tables: one read of the block's bytes; three: the odd one out on its own)
    const int a = 0, b = ntab == 2 ? 1 : (view_of_pca >= 0 ? view_of_pca : 2);
    TPG_TRY(tpg_view_create_pair(ctx, f, j->rowInd1, n, nullptr, 0, tab[a], tab[b], &v[a], &v[b]));
    if (ntab == 3) {
      const int c = 3 - a - b;
      TPG_TRY(tpg_view_create(ctx, f, j->rowInd1, n, nullptr, 0, tab[c], &v[c]));
    }
    return TPG_OK;
  }

  // ------------------------------------------------------------------ first sweep
  int alloc_common() {
    const size_t unit = src->unit();
    const int64_t bmax = std::min<int64_t>(B, P1 - P0);
    for (int k = 0; k < 2 && k < nblocks; k++) TPG_HIP(tpg_pmalloc((void**)&d_blk[k], (size_t)bmax * unit));
    for (int k = 0; k < 2; k++) {
      if (!out[k].ev) TPG_HIP(hipEventCreateWithFlags(&out[k].ev, hipEventDisableTiming));
      if (!out[k].ev2) TPG_HIP(hipEventCreateWithFlags(&out[k].ev2, hipEventDisableTiming));
    }
    for (int k = 0; k < 4; k++)
      if (!ev_fin[k]) TPG_HIP(hipEventCreateWithFlags(&ev_fin[k], hipEventDisableTiming));
    return TPG_OK;
  }

  int sweep1() {
    const tpg_stream_job* j = job;
    TpgEnter _enter(ctx);
    t_start = now();
    base_used = device_used();
    peak_used = base_used;
    TPG_TRY(alloc_common());
    const int G = j->ngroups, P = j->P;
    const int64_t bmax = std::min<int64_t>(B, P1 - P0);
    if (want_pw) {
      if (comm) TPG_TRY(tpg_pairwise_create_sharded(ctx, comm, n, &pw));
      else TPG_TRY(tpg_pairwise_create(ctx, n, nullptr, &pw));
    }
    if (want_fst) {
      TPG_TRY(d_fsum.alloc(sizeof(double) * 2 * (size_t)P * (size_t)j->nfst));
      TPG_TRY(d_fpart.alloc(sizeof(double) * 2 * (size_t)P));
      TPG_HIP(hipMemsetAsync(d_fsum.p, 0, sizeof(double) * 2 * (size_t)P * (size_t)j->nfst, ctx->stream));
    }
    if (want_pca) {
      TPG_TRY(d_K.alloc(sizeof(double) * (size_t)n * (size_t)n));
      have_K = false;
      fro = 0.0;
      // short blocks take the digit-split Gram kernel (pca.hip), whose weights are fixed point (22 fractional bits: 1e-10 on
      // the singular values of a whole panel).  Under a budget -- many short blocks, and a second sweep over PCIe that the
      // extra digits hide behind -- the weights keep eight more bits (restored in cleanup()); without one the pipeline is
      // paced by its kernels, and a block's Gram matrix costs what the resident path's costs.
      saved_fbits = ctx->pca_digit_fbits;
      if (budget) ctx->pca_digit_fbits = 30;
      // the Gram in batches of blocks (see `batch_gram`): a run without a budget that keeps its views and has more than two
      // blocks (with one or two the batches ARE the blocks).  TPG_STREAM_GRAM_BATCH=0: block by block (A/B)
      const char* gb = getenv("TPG_STREAM_GRAM_BATCH");
      big_views = budget == 0 && keep_views && nblocks > 1;
      batch_gram = big_views && nblocks > 2 && src->kind != SRC_BYTES && !(gb && atoi(gb) == 0);
      gram_from = 0;
      if (big_views) {
        const size_t per128 = (size_t)ceil_div(n, 128) * 4096;
        TPG_TRY(bigL.alloc((size_t)ceil_div(P1 - P0, 128) * per128));
        TPG_TRY(big_c.alloc(8 * (size_t)(P1 - P0)));
        TPG_TRY(big_s.alloc(8 * (size_t)(P1 - P0)));
      }
    }
    for (int s = 0; s < 2 && s < nblocks; s++) {
      OutSlot& o = out[s];
      if (j->alt_freq) TPG_TRY(o.af.alloc(16 * (size_t)bmax));
      if (j->grouped_alt_freq) TPG_TRY(o.gaf.alloc(16 * (size_t)G * (size_t)bmax));
      if (j->grouped_missingness) TPG_TRY(o.gm.alloc(8 * (size_t)G * (size_t)bmax));
      if (j->loci_counts) TPG_TRY(o.lc.alloc(16 * (size_t)bmax));
      for (int i = 0; i < j->nfst; i++)
        if (j->fst_by_locus[i]) {
          TPG_TRY(o.fl[i].alloc(8 * (size_t)P * (size_t)bmax));
          if (j->fst_return_num_dem) TPG_TRY(o.fd[i].alloc(8 * (size_t)P * (size_t)bmax));
        }
      if (want_pca) {
        TPG_TRY(o.dc.alloc(8 * (size_t)bmax));
        TPG_TRY(o.ds.alloc(8 * (size_t)bmax));
      }
    }
    if (nblocks == 0) return TPG_OK;
    if (ntab == 1) set_bedpack(tab[0]);
    TPG_TRY(start_uploader());
    std::vector<double> tot_scratch((size_t)(P > 0 ? P : 1));
    for (int64_t b = 0; b < nblocks; b++) {
      int64_t q0, q1;
      block_range(b, &q0, &q1);
      const int64_t mb = q1 - q0;
      const int slot = (int)(b & 1);
      stamp("wait for block", b);
      TPG_TRY(wait_block(b));
      stamp("got block", b);
      const tpg_fbm f = block_fbm(slot, mb);
      tpg_view* v[3];
      struct Views {
        tpg_view** v;
        ~Views() { for (int t = 0; t < 3; t++) tpg_view_free(v[t]); }
      } views{v};
      TPG_TRY(make_views(&f, slot, v));
      // (a view creation ends with a host round trip behind its pack kernel: the block buffer has been read)
      release_block(b);
      stamp("packed", b);
      OutSlot& o = out[slot];
      TPG_TRY(wait_slot(slot));
      stamp("output slot free", b);
      if (want_loc || want_fst) {
        const tpg_view* vl = v[view_of_loc];
        if (j->alt_freq) TPG_TRY(tpg_alt_freq_dip_pseudo(ctx, vl, j->ploidy, j->as_counts, o.af.as<double>()));
        if (j->grouped_alt_freq)
          TPG_TRY(tpg_grouped_alt_freq_dip_pseudo(ctx, vl, j->groupIds0, G, j->ploidy, j->as_counts, o.gaf.as<double>()));
        if (j->grouped_missingness) TPG_TRY(tpg_grouped_missingness(ctx, vl, j->groupIds0, G, o.gm.as<double>()));
        if (j->loci_counts) TPG_TRY(tpg_loci_counts(ctx, vl, o.lc.as<int32_t>()));
        for (int i = 0; i < j->nfst; i++) {
          if (j->fst_tot[i]) {
            TPG_TRY(tpg_pairwise_pop_fst_sums(ctx, vl, j->groupIds0, G, j->ploidy, j->fst_method[i], j->pairs1, P, d_fpart.as<double>(),
                                              d_fpart.as<double>() + P));
            hipLaunchKernelGGL(tpg_stream_add_kernel, dim3(8), dim3(256), 0, ctx->stream, d_fsum.as<double>() + 2 * (size_t)P * (size_t)i,
                               (const double*)d_fpart.as<double>(), (int64_t)2 * P);
            TPG_CHECK_LAUNCH();
          }
          if (j->fst_by_locus[i])
            TPG_TRY(tpg_pairwise_pop_fst(ctx, vl, j->groupIds0, G, j->ploidy, j->fst_method[i], j->pairs1, P, 1, j->fst_return_num_dem ? 1 : 0,
                                         tot_scratch.data(), o.fl[i].as<double>(), j->fst_return_num_dem ? o.fd[i].as<double>() : nullptr));
        }
        TPG_HIP(hipEventRecord(o.ev, ctx->stream));
        TPG_TRY(rows_out(j->alt_freq, 8, m, q0, o.af.p, mb, 2, o.ev, slot));
        TPG_TRY(rows_out(j->grouped_alt_freq, 8, m, q0, o.gaf.p, mb, 2 * (int64_t)G, o.ev, slot));
        TPG_TRY(rows_out(j->grouped_missingness, 8, m, q0, o.gm.p, mb, G, o.ev, slot));
        // m x 4 int32 ROW-major: the block's rows are one contiguous piece
        if (j->loci_counts) TPG_TRY(rows_out(j->loci_counts, 16, 1, q0, o.lc.p, mb, 1, o.ev, slot));
        for (int i = 0; i < j->nfst; i++) {
          TPG_TRY(rows_out(j->fst_by_locus[i], 8, m, q0, o.fl[i].p, mb, P, o.ev, slot));
          if (j->fst_return_num_dem && j->fst_by_locus[i]) TPG_TRY(rows_out(j->fst_by_locus_den[i], 8, m, q0, o.fd[i].p, mb, P, o.ev, slot));
        }
      }
      stamp("per-locus enqueued", b);
      if (want_pw) TPG_TRY(tpg_pairwise_accumulate_products(ctx, pw, v[view_of_pw], 0, -1, products));
      if (want_pca) {
        tpg_view* vp = v[view_of_pca];
        double *dc = o.dc.as<double>(), *ds = o.ds.as<double>();
        Kept* kp = nullptr;
        if (keep_views) {
          kept.emplace_back();
          kp = &kept.back();
          kp->q0 = q0;
          kp->mb = mb;
          TPG_TRY(kp->dc.alloc(8 * (size_t)mb));
          TPG_TRY(kp->ds.alloc(8 * (size_t)mb));
          dc = kp->dc.as<double>();
          ds = kp->ds.as<double>();
        }
        TPG_TRY(tpg_pca_center_scale(ctx, vp, dc, ds));  // TPG_ENUMERIC on a missing value / a zero scale, as big_SVD stops
        if (!batch_gram) {
          if (!have_K) TPG_TRY(tpg_pca_gram(ctx, vp, dc, ds, d_K.as<double>()));
          else TPG_TRY(tpg_pca_gram_add(ctx, vp, dc, ds, d_K.as<double>()));
          have_K = true;
        }
        if (j->square_frobenius) {
          double fb = 0;
          TPG_TRY(tpg_square_frobenius(ctx, vp, dc, ds, &fb));
          fro += fb;
        }
        // center / scale of the block leave now (the kept copies live until the end of the run; a slot's until block b + 2)
        hipEvent_t ev = o.ev2;
        TPG_HIP(hipEventRecord(ev, ctx->stream));
        TPG_TRY(rows_out(j->center, 8, m, q0, dc, mb, 1, ev, keep_views ? -1 : slot));
        TPG_TRY(rows_out(j->scale, 8, m, q0, ds, mb, 1, ev, keep_views ? -1 : slot));
        if (kp) {
          kp->v = vp;  // the imputed view stays for the loadings
          v[view_of_pca] = nullptr;
          // what the loadings read is L; the other layouts go back to the pool now
          if (kp->v->T) { tpg_pfree(kp->v->T); kp->v->T = nullptr; }
          if (kp->v->T4) { tpg_pfree(kp->v->T4); kp->v->T4 = nullptr; }
          if (kp->v->lc_part) { tpg_pfree(kp->v->lc_part); kp->v->lc_part = nullptr; kp->v->lc_chunks = 0; }
          if (kp->v->gc_cache.cnt) { tpg_pfree(kp->v->gc_cache.cnt); kp->v->gc_cache.cnt = nullptr; kp->v->gc_cache.nclass = 0; }
          kp->v->gc_cls.clear();
        }
        if (big_views) {
          // the block's L joins the others (a block starts on a multiple of 128 loci: whole locus tiles), its own copy goes back
          const size_t per128 = (size_t)ceil_div(n, 128) * 4096;  // bytes of L per 128 loci
          uint8_t* at = bigL.as<uint8_t>() + (size_t)((q0 - P0) / 128) * per128;
          TPG_HIP(tpg_copy_dev(ctx, at, kp->v->L, (size_t)kp->v->KG * per128));
          TPG_HIP(tpg_copy_dev(ctx, big_c.as<double>() + (q0 - P0), dc, 8 * (size_t)mb));
          TPG_HIP(tpg_copy_dev(ctx, big_s.as<double>() + (q0 - P0), ds, 8 * (size_t)mb));
          tpg_pfree(kp->v->L);  // stream-ordered
          kp->v->L = (uint4*)at;
          kp->L_borrowed = true;
          if (batch_gram && b == nblocks - 1) {
            tpg_view bv{};
            bv.ctx = ctx;
            bv.n = n;
            bv.m = (q1 - P0) - gram_from;
            bv.Q = ceil_div(n, 128);
            bv.KG = ceil_div(bv.m, 128);
            bv.L = (uint4*)(bigL.as<uint8_t>() + (size_t)(gram_from / 128) * per128);
            bv.bytes_each = (size_t)bv.KG * per128;
            const double *bc = big_c.as<double>() + gram_from, *bs = big_s.as<double>() + gram_from;
            if (!have_K) TPG_TRY(tpg_pca_gram(ctx, &bv, bc, bs, d_K.as<double>()));
            else TPG_TRY(tpg_pca_gram_add(ctx, &bv, bc, bs, d_K.as<double>()));
            have_K = true;
            gram_from = q1 - P0;
            stamp("gram batch enqueued", b);
          }
        }
      }
      sample();
    }
    up_th.join();
    if (sh.failed) { tpg_set_error("%s", sh.msg.c_str()); return sh.code; }
    t_sweep1 = now() - t_start;
    stamp("first sweep enqueued");
    return TPG_OK;
  }

  // ------------------------------------------------------------------ after the sweep
  // phase A (rank-local + exchanges): pairwise epilogues, Fst ratios, Gram all-reduce; phase B: eigen step, loadings
  int finish() {
    const tpg_stream_job* j = job;
    TpgEnter _enter(ctx);
    const int P = j->P, k = j->k;
    if (want_pw) {
      if (comm) {
        TPG_TRY(tpg_pairwise_reduce(ctx, comm, pw));
        // every device finishes its band of tiles and writes it into the caller's matrices itself
        TPG_TRY(tpg_pairwise_epilogues_sharded(ctx, comm, pw, j->ibs_type, m, j->ibs, j->king, j->allele_sharing, j->grm));
      } else {
        double* outs[4] = {j->ibs, j->king, j->allele_sharing, j->grm};
        double* dev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int q = 0; q < 4; q++) {
          if (!outs[q]) continue;
          if (tpg_is_device_ptr(outs[q])) { dev[q] = outs[q]; continue; }
          TPG_TRY(d_nn[q].alloc(8 * (size_t)n * (size_t)n));
          dev[q] = d_nn[q].as<double>();
        }
        TPG_TRY(tpg_pairwise_epilogues(ctx, pw, j->ibs_type, m, dev[0], dev[1], dev[2], dev[3]));
        TPG_HIP(hipEventRecord(ev_fin[0], ctx->stream));
        stamp("epilogues enqueued");
        // the N x N results go down beside the eigen step
        for (int q = 0; q < 4; q++)
          if (outs[q] && dev[q] != outs[q]) TPG_TRY(rows_out(outs[q], 8, n * n, 0, dev[q], n * n, 1, ev_fin[0], -1));
      }
      tpg_pairwise_free(pw);
      pw = nullptr;
    }
    if (want_fst) {
      const int64_t cnt = 2 * (int64_t)P * j->nfst;
      if (comm) TPG_TRY(tpg_comm_allreduce(comm, d_fsum.p, cnt, 1));
      std::vector<double> sums((size_t)cnt);
      TPG_HIP(tpg_download(ctx, sums.data(), d_fsum.p, sizeof(double) * (size_t)cnt));
      std::vector<double> ratio((size_t)P);
      for (int i = 0; i < j->nfst; i++) {
        if (!j->fst_tot[i]) continue;
        const double* s = sums.data() + 2 * (size_t)P * (size_t)i;
        for (int q = 0; q < P; q++) ratio[(size_t)q] = s[q] / s[P + q];
        if (comm && comm->rank != 0) continue;  // (one writer of the caller's array)
        if (tpg_is_device_ptr(j->fst_tot[i])) TPG_HIP(tpg_h2d_async(ctx, j->fst_tot[i], ratio.data(), sizeof(double) * (size_t)P));
        else memcpy(j->fst_tot[i], ratio.data(), sizeof(double) * (size_t)P);
      }
    }
    if (!want_pca) return TPG_OK;
    if (!have_K) TPG_HIP(hipMemsetAsync(d_K.p, 0, sizeof(double) * (size_t)n * (size_t)n, ctx->stream));  // a device without loci
    if (comm) {
      TPG_TRY(tpg_pca_gram_allreduce(ctx, comm, d_K.as<double>(), n));
      if (j->square_frobenius) TPG_TRY(tpg_comm_allreduce_f64(ctx, comm, &fro, 1));
    }
    if (j->square_frobenius && (!comm || comm->rank == 0)) *j->square_frobenius = fro;
    TPG_TRY(d_u.alloc(8 * (size_t)n * (size_t)k));
    std::vector<double> lam((size_t)k);
    TPG_TRY(tpg_sym_eig_topk_tol(ctx, d_K.as<double>(), n, k, j->pca_tol > 0 ? std::max(j->pca_tol, 1e-12) : 1e-12, lam.data(), d_u.as<double>()));
    d_K.free();
    stamp("eigen step done");
    dh.resize((size_t)k);
    for (int q = 0; q < k; q++) dh[(size_t)q] = sqrt(lam[(size_t)q] > 0 ? lam[(size_t)q] : 0.0);
    if (!comm || comm->rank == 0) {
      if (tpg_is_device_ptr(j->d)) TPG_HIP(tpg_h2d_async(ctx, j->d, dh.data(), 8 * (size_t)k));
      else memcpy(j->d, dh.data(), 8 * (size_t)k);
      TPG_HIP(hipEventRecord(ev_fin[1], ctx->stream));
      TPG_TRY(rows_out(j->u, 8, n * k, 0, d_u.p, n * k, 1, ev_fin[1], -1));
    }
    return loadings();
  }

  // v = Z'u / d of this device's loci: from the kept views, or from a second sweep over the store
  int loadings() {
    const tpg_stream_job* j = job;
    const int k = j->k;
    const int64_t bmax = std::min<int64_t>(B, P1 - P0);
    if (nblocks == 0) return TPG_OK;
    for (int s = 0; s < 2 && s < nblocks; s++) TPG_TRY(out[s].dv.alloc(8 * (size_t)k * (size_t)bmax));
    if (keep_views && big_views) {
      const size_t per128 = (size_t)ceil_div(n, 128) * 4096;
      tpg_view bv{};
      bv.ctx = ctx;
      bv.n = n;
      bv.m = P1 - P0;
      bv.Q = ceil_div(n, 128);
      bv.KG = ceil_div(bv.m, 128);
      bv.L = bigL.as<uint4>();
      bv.bytes_each = (size_t)bv.KG * per128;
      TPG_TRY(d_vbig.alloc(8 * (size_t)k * (size_t)bv.m));
      TPG_TRY(tpg_pca_loadings(ctx, &bv, big_c.as<double>(), big_s.as<double>(), d_u.as<double>(), dh.data(), k, d_vbig.as<double>()));
      TPG_HIP(hipEventRecord(ev_fin[2], ctx->stream));
      TPG_TRY(rows_out(j->v, 8, m, P0, d_vbig.p, bv.m, k, ev_fin[2], -1));
      for (Kept& kp : kept) free_kept(kp);
      sample();
      return TPG_OK;
    }
    if (keep_views) {
      int64_t b = 0;
      for (Kept& kp : kept) {
        const int slot = (int)(b & 1);
        OutSlot& o = out[slot];
        TPG_TRY(wait_slot(slot));
        TPG_TRY(tpg_pca_loadings(ctx, kp.v, kp.dc.as<double>(), kp.ds.as<double>(), d_u.as<double>(), dh.data(), k, o.dv.as<double>()));
        TPG_HIP(hipEventRecord(o.ev, ctx->stream));
        TPG_TRY(rows_out(j->v, 8, m, kp.q0, o.dv.p, kp.mb, k, o.ev, slot));
        free_kept(kp);
        b++;
        sample();
      }
      return TPG_OK;
    }
    sweeps = 2;
    set_bedpack(code_pca);  // one table now, whatever the first sweep needed
    TPG_TRY(start_uploader());
    for (int64_t b = 0; b < nblocks; b++) {
      int64_t q0, q1;
      block_range(b, &q0, &q1);
      const int64_t mb = q1 - q0;
      const int slot = (int)(b & 1);
      TPG_TRY(wait_block(b));
      const tpg_fbm f = block_fbm(slot, mb);
      tpg_view* vp = nullptr;
      TPG_TRY(tpg_view_create(ctx, &f, j->rowInd1, n, nullptr, 0, block_table(slot, code_pca), &vp));
      release_block(b);
      OutSlot& o = out[slot];
      int rc = wait_slot(slot);
      // the same per-locus center and scale as in the first sweep (the same counts through the same arithmetic)
      if (rc == TPG_OK) rc = tpg_pca_center_scale(ctx, vp, o.dc.as<double>(), o.ds.as<double>());
      if (rc == TPG_OK) rc = tpg_pca_loadings(ctx, vp, o.dc.as<double>(), o.ds.as<double>(), d_u.as<double>(), dh.data(), k, o.dv.as<double>());
      tpg_view_free(vp);
      TPG_TRY(rc);
      TPG_HIP(hipEventRecord(o.ev, ctx->stream));
      TPG_TRY(rows_out(j->v, 8, m, q0, o.dv.p, mb, k, o.ev, slot));
      sample();
    }
    up_th.join();
    if (sh.failed) { tpg_set_error("%s", sh.msg.c_str()); return sh.code; }
    return TPG_OK;
  }

  int sweeps = 1;

  int end() {  // everything enqueued has run, every download has landed
    TpgEnter _enter(ctx);
    stamp("everything enqueued");
    int rc = join_downloader();
    stamp("downloads landed");
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (rc == TPG_OK && e != hipSuccess) { tpg_set_error("stream: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
    sample();
    return rc;
  }

  void cleanup() {
    sh.fail(TPG_EINVAL, "stopped");  // (a no-op after a failure; after success nobody is waiting)
    if (up_th.joinable()) up_th.join();
    if (down_th.joinable()) down_th.join();
    down_running = false;
    if (ctx) {
      TpgEnter _enter(ctx);
      if (saved_fbits) { ctx->pca_digit_fbits = saved_fbits; saved_fbits = 0; }
      (void)hipStreamSynchronize(ctx->stream);
      if (down_ctx) (void)hipStreamSynchronize(down_ctx->stream);
      if (up_ctx) (void)hipStreamSynchronize(up_ctx->stream);
      for (Kept& kp : kept) { free_kept(kp); kp.dc.free(); kp.ds.free(); }
      kept.clear();
      bigL.free();
      big_c.free();
      big_s.free();
      d_vbig.free();
      if (pw) { tpg_pairwise_free(pw); pw = nullptr; }
      for (int k = 0; k < 2; k++) {
        if (d_blk[k]) { tpg_pfree(d_blk[k]); d_blk[k] = nullptr; }
        OutSlot& o = out[k];
        o.af.free(); o.gaf.free(); o.gm.free(); o.lc.free(); o.dc.free(); o.ds.free(); o.dv.free();
        for (int i = 0; i < TPG_STREAM_MAX_FST; i++) { o.fl[i].free(); o.fd[i].free(); }
        if (o.ev) { (void)hipEventDestroy(o.ev); o.ev = nullptr; }
        if (o.ev2) { (void)hipEventDestroy(o.ev2); o.ev2 = nullptr; }
      }
      for (int k = 0; k < 4; k++) {
        d_nn[k].free();
        if (ev_fin[k]) { (void)hipEventDestroy(ev_fin[k]); ev_fin[k] = nullptr; }
      }
      d_K.free(); d_fsum.free(); d_fpart.free(); d_u.free();
    }
    if (own_workers) {
      if (up_ctx) tpg_ctx_destroy(up_ctx);
      if (down_ctx) tpg_ctx_destroy(down_ctx);
    }
    up_ctx = down_ctx = nullptr;
  }
};

static int check_job(const tpg_stream* s, const tpg_stream_job* job, int64_t* n, int64_t* m) {
  TPG_REQUIRE(s && job, TPG_EINVAL, "null argument");
  TPG_REQUIRE(job->struct_size == sizeof(tpg_stream_job), TPG_EINVAL, "tpg_stream_job of %zu bytes, this library's has %zu",
              job->struct_size, sizeof(tpg_stream_job));
  *n = job->rowInd1 ? job->n : s->src.nrow;
  *m = job->colInd1 ? job->m : s->src.ncol;
  TPG_REQUIRE(*n > 0 && *m > 0, TPG_EINVAL, "empty selection (%lld x %lld)", (long long)*n, (long long)*m);
  if (job->rowInd1)
    for (int64_t i = 0; i < *n; i++)
      TPG_REQUIRE(job->rowInd1[i] >= 1 && job->rowInd1[i] <= s->src.nrow, TPG_EINVAL, "rowInd[%lld] = %d out of [1,%lld]", (long long)i,
                  job->rowInd1[i], (long long)s->src.nrow);
  if (job->colInd1)
    for (int64_t q = 0; q < *m; q++)
      TPG_REQUIRE(job->colInd1[q] >= 1 && job->colInd1[q] <= s->src.ncol, TPG_EINVAL, "colInd[%lld] = %d out of [1,%lld]", (long long)q,
                  job->colInd1[q], (long long)s->src.ncol);
  return TPG_OK;
}

static void fill_report(const StreamRun& r, size_t budget, tpg_stream_report* rep) {
  if (!rep) return;
  memset(rep, 0, sizeof(*rep));
  rep->blocks = r.nblocks;
  rep->block_loci = r.B;
  rep->sweeps = r.sweeps;
  rep->views_kept = r.keep_views ? 1 : 0;
  rep->bytes_up = r.bytes_up;
  rep->bytes_down = r.bytes_down;
  rep->budget_bytes = budget;
  rep->planned_bytes = r.planned;
  rep->state_bytes = r.state;
  rep->peak_device_bytes = r.peak_used > r.base_used ? r.peak_used - r.base_used : 0;
  rep->seconds = StreamRun::now() - r.t_start;
  rep->seconds_first_sweep = r.t_sweep1;
}

static int open_common(tpg_ctx* ctx, const StreamSource& src, size_t budget, tpg_stream** out) {
  TPG_REQUIRE(ctx && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(src.nrow > 0 && src.ncol > 0, TPG_EINVAL, "empty store (%lld x %lld)", (long long)src.nrow, (long long)src.ncol);
  tpg_stream* s = new tpg_stream();
  s->ctx = ctx;
  s->src = src;
  s->budget = budget;
  *out = s;
  return TPG_OK;
}

static int map_file(const char* path, size_t need, void** base) {
  const int fd = open(path, O_RDONLY);
  TPG_REQUIRE(fd >= 0, TPG_EINVAL, "cannot open %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < need) {
    close(fd);
    tpg_set_error("%s is smaller than the %zu bytes of the store", path, need);
    return TPG_EINVAL;
  }
  void* p = mmap(nullptr, need, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  TPG_REQUIRE(p != MAP_FAILED, TPG_EINVAL, "mmap of %s failed", path);
  *base = p;
  return TPG_OK;
}

}  // namespace

extern "C" int tpg_stream_open_host(tpg_ctx* ctx, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, size_t budget_bytes,
                                    tpg_stream** out) {
  TPG_REQUIRE(fbm_bytes, TPG_EINVAL, "null store");
  StreamSource src;
  src.kind = SRC_BYTES;
  src.bytes = fbm_bytes;
  src.nrow = nrow;
  src.ncol = ncol;
  return open_common(ctx, src, budget_bytes, out);
}

extern "C" int tpg_stream_open_bk(tpg_ctx* ctx, const char* path, int64_t nrow, int64_t ncol, size_t budget_bytes, tpg_stream** out) {
  TPG_REQUIRE(path && nrow > 0 && ncol > 0, TPG_EINVAL, "bad argument");
  void* base = nullptr;
  const size_t need = (size_t)nrow * (size_t)ncol;
  TPG_TRY(map_file(path, need, &base));
  const int rc = tpg_stream_open_host(ctx, (const uint8_t*)base, nrow, ncol, budget_bytes, out);
  if (rc != TPG_OK) { munmap(base, need); return rc; }
  (*out)->map_base = base;
  (*out)->map_len = need;
  return TPG_OK;
}

extern "C" int tpg_stream_open_bed_host(tpg_ctx* ctx, const uint8_t* payload, int64_t n, int64_t m, size_t budget_bytes, tpg_stream** out) {
  TPG_REQUIRE(payload, TPG_EINVAL, "null store");
  StreamSource src;
  src.kind = SRC_BED;
  src.bytes = payload;
  src.nrow = n;
  src.ncol = m;
  src.bpl = (n + 3) / 4;
  return open_common(ctx, src, budget_bytes, out);
}

extern "C" int tpg_stream_open_bed(tpg_ctx* ctx, const char* path, int64_t n, int64_t m, size_t budget_bytes, tpg_stream** out) {
  TPG_REQUIRE(path && n > 0 && m > 0, TPG_EINVAL, "bad argument");
  void* base = nullptr;
  const size_t need = 3 + (size_t)((n + 3) / 4) * (size_t)m;
  TPG_TRY(map_file(path, need, &base));
  const uint8_t* b = (const uint8_t*)base;
  if (b[0] != 0x6C || b[1] != 0x1B || b[2] != 0x01) {
    tpg_set_error("%s is not a SNP-major PLINK .bed (magic %02x %02x %02x)", path, b[0], b[1], b[2]);
    munmap(base, need);
    return TPG_EINVAL;
  }
  const int rc = tpg_stream_open_bed_host(ctx, b + 3, n, m, budget_bytes, out);
  if (rc != TPG_OK) { munmap(base, need); return rc; }
  (*out)->map_base = base;
  (*out)->map_len = need;
  return TPG_OK;
}

extern "C" int tpg_stream_open_synth(tpg_ctx* ctx, uint64_t seed, int64_t nrow, int64_t ncol, int npop, uint32_t miss_thresh,
                                     int imputed_bytes, size_t budget_bytes, tpg_stream** out) {
  TPG_REQUIRE(npop > 0 && npop <= 1024, TPG_EINVAL, "bad synth shape");
  StreamSource src;
  src.kind = SRC_SYNTH;
  src.nrow = nrow;
  src.ncol = ncol;
  src.seed = seed;
  src.npop = npop;
  src.miss = miss_thresh;
  src.imputed = imputed_bytes;
  return open_common(ctx, src, budget_bytes, out);
}

extern "C" void tpg_stream_close(tpg_stream* s) {
  if (!s) return;
  if (s->map_base) munmap(s->map_base, s->map_len);
  if (s->up_ctx) tpg_ctx_destroy(s->up_ctx);
  if (s->down_ctx) tpg_ctx_destroy(s->down_ctx);
  delete s;
}

extern "C" int tpg_stream_run(tpg_ctx* ctx, tpg_stream* s, const tpg_stream_job* job, tpg_stream_report* report) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && s && job, TPG_EINVAL, "null argument");
  TPG_REQUIRE(s->ctx == ctx, TPG_EINVAL, "the stream was opened on another context");
  int64_t n = 0, m = 0;
  TPG_TRY(check_job(s, job, &n, &m));
  StreamRun run;
  run.ctx = ctx;
  run.src = &s->src;
  run.job = job;
  run.budget = s->budget;
  run.n = n;
  run.m = m;
  run.P0 = 0;
  run.P1 = m;
  if (!s->up_ctx) TPG_TRY(tpg_ctx_create(ctx->device, &s->up_ctx));
  if (!s->down_ctx) TPG_TRY(tpg_ctx_create(ctx->device, &s->down_ctx));
  run.up_ctx = s->up_ctx;
  run.down_ctx = s->down_ctx;
  run.own_workers = false;
  TPG_TRY(run.setup());
  TPG_TRY(run.plan());
  int rc = run.sweep1();
  if (rc == TPG_OK) rc = run.finish();
  const int rc2 = run.end();
  if (rc == TPG_OK) rc = rc2;
  std::string err = rc == TPG_OK ? "" : tpg_last_error();
  fill_report(run, s->budget, report);
  run.cleanup();
  if (rc != TPG_OK) tpg_set_error("%s", err.c_str());
  return rc;
}

// for the tpg_multi_* entry points of comm.hip when a device's share of the panel is not to be held whole (multi_stream_budget)
int tpg_multi_stream_host(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, size_t budget, const tpg_stream_job* job) {
  tpg_stream s;
  s.ctx = tpg_multi_ctx(mg, 0);
  s.src.kind = SRC_BYTES;
  s.src.bytes = fbm_bytes;
  s.src.nrow = nrow;
  s.src.ncol = ncol;
  s.budget = budget;
  return tpg_multi_stream_run(mg, &s, job, nullptr);
}

// Several devices (one process): every device streams its contiguous share of colInd (tpg_shard_loci), then the exchanges.
// Phases are separate thread teams, as in comm.hip: a failure in a rank-local phase is known to all before anyone enters a
// collective.
extern "C" int tpg_multi_stream_run(tpg_multi* mg, tpg_stream* s, const tpg_stream_job* job, tpg_stream_report* report) {
  TPG_REQUIRE(mg && s && job, TPG_EINVAL, "null argument");
  int64_t n = 0, m = 0;
  TPG_TRY(check_job(s, job, &n, &m));
  const int ndev = tpg_multi_ndev(mg);
  void* outs[] = {job->ibs, job->king, job->allele_sharing, job->grm, job->alt_freq, job->grouped_alt_freq, job->grouped_missingness,
                  job->loci_counts, job->d, job->u, job->v, job->center, job->scale};
  for (void* p : outs) TPG_REQUIRE(!p || !tpg_is_device_ptr(p), TPG_EINVAL, "tpg_multi_stream_run writes host memory only");
  tpg_stage_keep(2 * ndev + 1);
  std::vector<std::unique_ptr<StreamRun>> runs;
  for (int r = 0; r < ndev; r++) {
    runs.emplace_back(new StreamRun());
    StreamRun& run = *runs.back();
    run.ctx = tpg_multi_ctx(mg, r);
    run.comm = tpg_multi_comm(mg, r);
    run.src = &s->src;
    run.job = job;
    run.budget = s->budget;
    run.n = n;
    run.m = m;
    TPG_TRY(tpg_shard_loci(m, ndev, r, &run.P0, &run.P1));
  }
  auto phase = [&](auto fn) -> int {
    std::vector<int> rcs((size_t)ndev, TPG_OK);
    std::vector<std::string> errs((size_t)ndev);
    std::vector<std::thread> th;
    for (int r = 0; r < ndev; r++)
      th.emplace_back([&, r]() {
        rcs[(size_t)r] = fn(*runs[(size_t)r]);
        if (rcs[(size_t)r] != TPG_OK) errs[(size_t)r] = tpg_last_error();
      });
    for (auto& t : th) t.join();
    for (int r = 0; r < ndev; r++)
      if (rcs[(size_t)r] != TPG_OK) {
        tpg_set_error("device %d: %s", runs[(size_t)r]->ctx->device, errs[(size_t)r].c_str());
        return rcs[(size_t)r];
      }
    return TPG_OK;
  };
  int rc = phase([&](StreamRun& run) -> int {
    TpgEnter _enter(run.ctx);
    TPG_TRY(tpg_ctx_create(run.ctx->device, &run.up_ctx));
    TPG_TRY(tpg_ctx_create(run.ctx->device, &run.down_ctx));
    TPG_TRY(run.setup());
    TPG_TRY(run.plan());
    return run.sweep1();
  });
  if (rc == TPG_OK)
    rc = phase([&](StreamRun& run) -> int {
      int r1 = run.finish();
      const int r2 = run.end();
      return r1 != TPG_OK ? r1 : r2;
    });
  std::string err = rc == TPG_OK ? "" : tpg_last_error();
  if (report) {
    fill_report(*runs[0], s->budget, report);
    for (int r = 1; r < ndev; r++) {
      tpg_stream_report q;
      fill_report(*runs[(size_t)r], s->budget, &q);
      report->blocks = std::max(report->blocks, q.blocks);
      report->sweeps = std::max(report->sweeps, q.sweeps);
      report->bytes_up += q.bytes_up;
      report->bytes_down += q.bytes_down;
      report->peak_device_bytes = std::max(report->peak_device_bytes, q.peak_device_bytes);
      report->planned_bytes = std::max(report->planned_bytes, q.planned_bytes);
    }
  }
  for (auto& run : runs) run->cleanup();
  if (rc != TPG_OK) tpg_set_error("%s", err.c_str());
  return rc;
}
