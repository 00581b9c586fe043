/*
 * tpg_rshim.c -- the R side of the drop-in boundary: `.Call` entry points with the reference's own symbol names and
 * arities (src/RcppExports.cpp:348-371 of the reference; R side R/RcppExports.R:4-91), forwarding to the C ABI of
 * libtpg_hip.so (include/tpg.h).  Plain C against R's C API only: no Rcpp, no bigstatsr headers -- everything the
 * reference's C++ takes from `BM` through bigstatsr's accessors is taken here from the FBM's plain R fields
 * (`backingfile`, `nrow`, `ncol`, `code256`) and the backing file itself.
 *
 * STATUS: written against R's documented C API.  R is not installed in the build image, so what is checked there is:
 * (1) the file compiles with -Wall -Wextra -Werror against tests/rmock/ (declarations of the ~35 R API functions it
 * uses, written from R's documentation -- a syntax and signature guard, NOT R), and (2) linked against the small mock
 * runtime of tests/rmock/rmock.c it is driven on the GPU by tests/test_gpu_rshim.py exactly as the R drivers drive it
 * (block loop of R/snp_ibs.R:69-82 on FBM objects whose fields are read through Rf_eval).  See INTEGRATION.md.
 *
 * Where it goes: tidypopgen/src/tpg_rshim.c, replacing the `[[Rcpp::export]]` bodies of the functions listed in
 * tpg_rshim_entries[] (INTEGRATION.md section 2 says how the registration tables are merged), or the package
 * shim/tpgshim beside an unmodified tidypopgen.  R code is UNCHANGED and correct by default: every increment_* call
 * adds its block's sums to k / k2 before it returns, as the reference does.  Opt-in fast path: TPG_RSHIM_DEFERRED=1
 * keeps the sums in HBM across the block loop; the three pairwise drivers then need ONE line after their loop
 * (`tpg_flush()`, see _tidypopgen_tpg_flush below).
 *
 * Threading: R calls these from its main thread only; the shim holds one tpg_ctx per R session.
 */
#define _POSIX_C_SOURCE 200809L /* strdup, mmap */
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include "tpg.h"

/* ---- session state -------------------------------------------------------------------------------------------- */

static tpg_ctx* g_ctx = NULL;

/* TPG_RSHIM_DEFERRED=1: the increment_* functions keep their sums in HBM until tpg_flush() (the R drivers must then
 * call it after their block loop).  Default: off -- an unmodified driver gets the reference's semantics. */
static int deferred(void) {
  const char* e = getenv("TPG_RSHIM_DEFERRED");
  return e && e[0] == '1';
}

static tpg_ctx* ctx(void) {
  if (!g_ctx) {
    const char* dev = getenv("TPG_DEVICE");
    /* TPG_RSHIM_NUMA_BIND=1 (opt-in: it changes the CPU affinity of the R session's thread, as starting R under
       `numactl --cpunodebind` would): stay on the host NUMA node of the GPU -- INTEGRATION.md 3b */
    const char* nb = getenv("TPG_RSHIM_NUMA_BIND");
    if (nb && nb[0] == '1') (void)tpg_host_bind_near_device(dev ? atoi(dev) : 0, NULL);
    if (tpg_ctx_create(dev ? atoi(dev) : 0, &g_ctx) != TPG_OK) Rf_error("tidypopgen (GPU): %s", tpg_last_error());
    if (deferred() && tpg_increment_defer(g_ctx, 1) != TPG_OK) Rf_error("tidypopgen (GPU): %s", tpg_last_error());
  }
  return g_ctx;
}

/* every failure of the library becomes an R error, as BEGIN_RCPP / END_RCPP turn C++ exceptions into R errors */
#define TPG_R(call)                                                         \
  do {                                                                      \
    if ((call) != TPG_OK) Rf_error("tidypopgen (GPU): %s", tpg_last_error()); \
  } while (0)

static SEXP field(SEXP env, const char* name) { /* a field or an active binding of the reference-class object */
  SEXP v = Rf_eval(Rf_install(name), env);
  if (v == R_UnboundValue) Rf_error("FBM object has no field '%s'", name);
  return v;
}

static int64_t field_i64(SEXP env, const char* name) {
  SEXP v = field(env, name);
  return TYPEOF(v) == REALSXP ? (int64_t)REAL(v)[0] : (int64_t)Rf_asInteger(v);
}

static const char* field_path(SEXP env, const char* name) {
  SEXP v = field(env, name);
  if (TYPEOF(v) != STRSXP || XLENGTH(v) < 1) Rf_error("FBM field '%s' is not a file name", name);
  return R_ExpandFileName(CHAR(STRING_ELT(v, 0)));
}

/* A backing file mapped into this process, and (opt-in, for genotype FBMs) its copy in HBM.
 * DEFAULT: no HBM copy outlives a call.  Every per-locus entry point uploads the columns its colInd covers (the R drivers
 * call block by block, so a driver loop moves the file once), exactly as the increment_* mirrors do: an FBM that is
 * rewritten in place between two calls -- bigsnpr::snp_fastImputeSimple (R/gt_impute_simple.R:86), gt_set_imputed-style
 * writes, any store through bigstatsr's own mapping -- can never be served stale.
 * TPG_RSHIM_CACHE=1 (opt-in): the whole genotype FBM is uploaded once and kept in HBM while the file's size, modification
 * time and a fingerprint of 512 pages spread over it are what they were at upload.  That is a HEURISTIC: stores through a
 * mapping do not reliably move the modification time before msync / munmap on tmpfs or NFS, and a sparse edit can miss
 * the sampled pages (0.04 % of a 5 GB file) -- enable it only for read-only data sets, or call tpg_invalidate(BM) /
 * tpg_release() after anything that writes to the FBM. */
typedef struct {
  char* path;
  void* map;
  size_t bytes;
  int writable;
  int64_t nrow, ncol;
  tpg_fbm* dev; /* NULL for the double N x N accumulators, and always unless TPG_RSHIM_CACHE=1 */
  int64_t mtime_ns;
  uint64_t fingerprint;
  uint64_t stamp; /* g_call of the last increment_* call that used this (writable) mapping */
  uint64_t dev_id, ino; /* st_dev / st_ino of the file the mapping was made of */
} mapped_file;

/* an array of POINTERS: a mapped_file* handed out stays valid when the table grows or an entry is forgotten */
static mapped_file** g_files = NULL;
static int g_nfiles = 0;
static uint64_t g_call = 0; /* counts increment_* calls */

static int cache_on(void) {
  const char* e = getenv("TPG_RSHIM_CACHE");
  return e && e[0] == '1';
}

static uint64_t fingerprint_of(const uint8_t* p, size_t bytes) { /* FNV-1a over up to 512 whole 4-KiB pages (2 MB read) */
  const size_t page = 4096, npages = (bytes + page - 1) / page, want = npages < 512 ? npages : 512;
  uint64_t h = 1469598103934665603ull;
  for (size_t k = 0; k < want; k++) {
    const size_t pg = want > 1 ? k * (npages - 1) / (want - 1) : 0, off = pg * page;
    const size_t len = bytes - off < page ? bytes - off : page;
    for (size_t t = 0; t < len; t += 8) {
      uint64_t w = 0;
      memcpy(&w, p + off + t, len - t < 8 ? len - t : 8);
      h = (h ^ w) * 1099511628211ull;
    }
  }
  return h;
}

static int64_t mtime_of(const char* path, size_t* size) {
  struct stat st;
  if (stat(path, &st) != 0) Rf_error("cannot stat backing file '%s'", path);
  *size = (size_t)st.st_size;
  return (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
}

static void forget_file(int k) { /* unmap, free the HBM copy, close the gap in g_files */
  mapped_file* f = g_files[k];
  if (f->dev) tpg_fbm_free(f->dev);
  munmap(f->map, f->bytes);
  free(f->path);
  free(f);
  g_files[k] = g_files[g_nfiles - 1];
  g_nfiles--;
}

static mapped_file* map_file(const char* path, size_t bytes, int writable, int64_t nrow, int64_t ncol) {
  /* a mapping is reused only while the PATH still names the file it was made of: a backing file unlinked and created again
     at the same path and size (file.remove() + a new FBM, an explicit backingfile=) is another inode, and adding into the
     old, unlinked one would leave R reading zeros from the new file without any error */
  struct stat st;
  const int have_st = stat(path, &st) == 0;
  for (int k = 0; k < g_nfiles; k++)
    if (strcmp(g_files[k]->path, path) == 0 && g_files[k]->bytes == bytes && g_files[k]->writable == writable) {
      if (have_st && g_files[k]->dev_id == (uint64_t)st.st_dev && g_files[k]->ino == (uint64_t)st.st_ino) return g_files[k];
      /* Under TPG_RSHIM_DEFERRED=1 the library may still hold this mapping's address for sums it has not written yet: they
         belong to the OLD file (the one the block loop was filling), so they are written there before the mapping goes --
         never into unmapped memory, never into the new file. */
      if (writable && deferred() && g_ctx && tpg_increment_flush(g_ctx) != TPG_OK)
        Rf_error("tidypopgen (GPU): %s", tpg_last_error());
      forget_file(k);
      break;
    }
  int fd = open(path, writable ? O_RDWR : O_RDONLY);
  if (fd < 0) Rf_error("cannot open backing file '%s'", path);
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) {
    close(fd);
    Rf_error("backing file '%s' is smaller than the FBM it should hold", path);
  }
  /* MAP_SHARED: the same pages bigstatsr's own mapping of the file reads and writes */
  void* p = mmap(NULL, bytes, writable ? (PROT_READ | PROT_WRITE) : PROT_READ, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) Rf_error("mmap of '%s' failed", path);
#ifdef MADV_POPULATE_WRITE
  /* an accumulator is written in full by the first call that uses it: one batched populate (Linux 5.14) instead of a write
     fault per page of a shared file mapping from the adding threads (49 000 per matrix at n = 5 000) */
  if (writable) (void)madvise(p, bytes, MADV_POPULATE_WRITE);
#endif
  mapped_file** nf = (mapped_file**)realloc(g_files, sizeof(mapped_file*) * (size_t)(g_nfiles + 1));
  mapped_file* f = nf ? (mapped_file*)calloc(1, sizeof(mapped_file)) : NULL;
  char* pc = f ? strdup(path) : NULL;
  if (nf) g_files = nf;
  if (!pc) {
    free(f);
    munmap(p, bytes);
    Rf_error("out of memory");
  }
  g_files[g_nfiles++] = f;
  f->path = pc;
  f->map = p;
  f->bytes = bytes;
  f->writable = writable;
  f->nrow = nrow;
  f->ncol = ncol;
  f->dev_id = (uint64_t)st.st_dev;
  f->ino = (uint64_t)st.st_ino;
  return f;
}

/* the genotype FBM.code256 behind `BM`: its host mapping */
static mapped_file* genotype_fbm(SEXP BM) {
  const int64_t nrow = field_i64(BM, "nrow"), ncol = field_i64(BM, "ncol");
  return map_file(field_path(BM, "backingfile"), (size_t)nrow * (size_t)ncol, 0, nrow, ncol);
}

/* TPG_RSHIM_CACHE=1 only: the HBM copy of the whole FBM, re-uploaded when the heuristic above says the bytes changed */
static tpg_fbm* genotype_fbm_dev(SEXP BM) {
  mapped_file* f = genotype_fbm(BM);
  size_t size = 0;
  const int64_t mt = mtime_of(f->path, &size);
  if (size < f->bytes) Rf_error("backing file '%s' shrank below the FBM it should hold", f->path);
  const uint64_t fp = fingerprint_of((const uint8_t*)f->map, f->bytes);
  if (f->dev && (mt != f->mtime_ns || fp != f->fingerprint)) { /* the bytes changed under us */
    tpg_fbm_free(f->dev);
    f->dev = NULL;
  }
  /* the library maps the file, touches its pages with a team of threads and uploads it with one copy */
  if (!f->dev) {
    TPG_R(tpg_fbm_open_bk(ctx(), f->path, f->nrow, f->ncol, &f->dev));
    f->mtime_ns = mt;
    f->fingerprint = fp;
  }
  return f->dev;
}

static const double* code256_of(SEXP BM) {
  SEXP c = field(BM, "code256");
  if (TYPEOF(c) != REALSXP || XLENGTH(c) != 256) Rf_error("BM$code256 is not a double[256]");
  return REAL(c);
}

/* a double FBM (the N x N accumulators the R drivers allocate with bigstatsr::FBM(n, n, init = 0)).  Its mapping is kept
 * while the block loop that uses it runs -- mapping a 200 MB file afresh for every block costs its 49 000 page faults
 * every time: 65 ms per call against 14 with the mapping kept (two matrices of 5 000 x 5 000, 8 threads adding) -- and is
 * dropped by the first increment_* call that does NOT use it (the next analysis), by tpg_flush / tpg_release and at unload,
 * so that a session holds at most one analysis' pair of R temp files mapped. */
static double* double_fbm(SEXP K, int64_t n) {
  const int64_t nrow = field_i64(K, "nrow"), ncol = field_i64(K, "ncol");
  if (nrow != n || ncol != n) Rf_error("accumulator FBM is %lld x %lld, expected %lld x %lld", (long long)nrow,
                                       (long long)ncol, (long long)n, (long long)n);
  mapped_file* f = map_file(field_path(K, "backingfile"), sizeof(double) * (size_t)n * (size_t)n, 1, n, n);
  f->stamp = g_call;
  return (double*)f->map;
}

static void release_accumulators(int all) {
  for (int k = g_nfiles - 1; k >= 0; k--)
    if (g_files[k]->writable && (all || g_files[k]->stamp != g_call)) forget_file(k);
}

/* after an increment_* call: by default its sums are already in k / k2; the accumulators of EARLIER analyses can go */
static void after_increment(void) {
  if (!deferred()) release_accumulators(0);
}

/* the packed (rowInd, colInd, code256) view a per-locus entry point works on.  Default: the columns colInd covers are
 * uploaded for this call alone (the contiguous blocks of the R drivers: their covering range; a scattered colInd: gathered
 * on the host first) and released with it; TPG_RSHIM_CACHE=1: packed from the cached HBM copy of the whole FBM. */
static tpg_view* view_of(SEXP BM, SEXP rowInd, SEXP colInd, int raw_bytes) {
  if (TYPEOF(rowInd) != INTSXP || TYPEOF(colInd) != INTSXP) Rf_error("rowInd / colInd must be integer vectors");
  const int64_t n = (int64_t)XLENGTH(rowInd), m = (int64_t)XLENGTH(colInd);
  const double* code = raw_bytes ? NULL : code256_of(BM);
  tpg_view* v = NULL;
  if (cache_on()) {
    TPG_R(tpg_view_create(ctx(), genotype_fbm_dev(BM), INTEGER(rowInd), n, INTEGER(colInd), m, code, &v));
    return v;
  }
  mapped_file* f = genotype_fbm(BM);
  const uint8_t* bytes = (const uint8_t*)f->map;
  const int64_t nrow = f->nrow, ncol = f->ncol;
  if (m < 1) Rf_error("tidypopgen (GPU): empty colInd");
  const int* ci = INTEGER(colInd);
  int lo = ci[0], hi = ci[0];
  for (int64_t j = 0; j < m; j++) {
    if (ci[j] < 1 || ci[j] > ncol) Rf_error("tidypopgen (GPU): colInd[%lld] = %d out of [1,%lld]", (long long)j, ci[j], (long long)ncol);
    if (ci[j] < lo) lo = ci[j];
    if (ci[j] > hi) hi = ci[j];
  }
  const int64_t span = (int64_t)hi - lo + 1;
  int* cols = (int*)R_alloc((size_t)m, sizeof(int)); /* R's transient storage: freed when .Call returns or errors */
  /* tpg_view_create_from_host: upload for this ONE code table (2 bits per genotype over PCIe where table and bytes allow it),
     pack, release the uploaded columns -- nothing of the FBM outlives the call */
  if (span <= 2 * m + 64) {
    for (int64_t j = 0; j < m; j++) cols[j] = ci[j] - (lo - 1);
    TPG_R(tpg_view_create_from_host(ctx(), bytes + (size_t)(lo - 1) * (size_t)nrow, nrow, span, INTEGER(rowInd), n, cols, m, code, &v));
  } else {
    uint8_t* stage = (uint8_t*)malloc((size_t)nrow * (size_t)m);
    if (!stage) Rf_error("tidypopgen (GPU): out of memory gathering %lld columns", (long long)m);
    for (int64_t j = 0; j < m; j++) {
      memcpy(stage + (size_t)j * (size_t)nrow, bytes + (size_t)(ci[j] - 1) * (size_t)nrow, (size_t)nrow);
      cols[j] = (int)(j + 1);
    }
    const int rc = tpg_view_create_from_host(ctx(), stage, nrow, m, INTEGER(rowInd), n, cols, m, code, &v); /* waited for: the staging buffer may go */
    free(stage);
    TPG_R(rc);
  }
  return v;
}

static SEXP named_list(int n, const char** names, SEXP* values) {
  SEXP out = PROTECT(Rf_allocVector(VECSXP, n));
  SEXP nm = PROTECT(Rf_allocVector(STRSXP, n));
  for (int k = 0; k < n; k++) {
    SET_VECTOR_ELT(out, k, values[k]);
    SET_STRING_ELT(nm, k, Rf_mkChar(names[k]));
  }
  Rf_setAttrib(out, R_NamesSymbol, nm);
  UNPROTECT(2);
  return out;
}

static void set_colnames2(SEXP mat, const char* a, const char* b) {
  SEXP cn = PROTECT(Rf_allocVector(STRSXP, 2));
  SET_STRING_ELT(cn, 0, Rf_mkChar(a));
  SET_STRING_ELT(cn, 1, Rf_mkChar(b));
  SEXP dn = PROTECT(Rf_allocVector(VECSXP, 2));
  SET_VECTOR_ELT(dn, 0, R_NilValue);
  SET_VECTOR_ELT(dn, 1, cn);
  Rf_setAttrib(mat, R_DimNamesSymbol, dn);
  UNPROTECT(2);
}

/* views are released even when the library call fails (Rf_error does not return) */
#define TPG_R_VIEW(v, call)          \
  do {                               \
    int _rc = (call);                \
    tpg_view_free(v);                \
    if (_rc != TPG_OK) Rf_error("tidypopgen (GPU): %s", tpg_last_error()); \
  } while (0)

/* ---- per-locus sweeps --------------------------------------------------------------------------------------- */

/* alt_freq_dip_pseudo_cpp(BM, rowInd, colInd, ploidy, ncores, as_counts)   src/alt_freq_dip_pseudo_cpp.cpp:8-58 */
SEXP _tidypopgen_alt_freq_dip_pseudo_cpp(SEXP BM, SEXP rowInd, SEXP colInd, SEXP ploidy, SEXP ncores, SEXP as_counts) {
  (void)ncores;
  const int counts = Rf_asLogical(as_counts);
  SEXP pl = PROTECT(Rf_coerceVector(ploidy, REALSXP));
  tpg_view* v = view_of(BM, rowInd, colInd, 0);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)XLENGTH(colInd), 2));
  TPG_R_VIEW(v, tpg_alt_freq_dip_pseudo(ctx(), v, REAL(pl), counts, REAL(out)));
  set_colnames2(out, counts ? "n_alt" : "freq", "n_valid"); /* :44, :56 */
  UNPROTECT(2);
  return out;
}

/* grouped_alt_freq_dip_pseudo_cpp(BM, rowInd, colInd, groupIds, ngroups, ploidy, ncores, as_counts)
   src/grouped_alt_freq_dip_pseudo_cpp.cpp:8-58 -> m x 2G */
SEXP _tidypopgen_grouped_alt_freq_dip_pseudo_cpp(SEXP BM, SEXP rowInd, SEXP colInd, SEXP groupIds, SEXP ngroups,
                                                 SEXP ploidy, SEXP ncores, SEXP as_counts) {
  (void)ncores;
  const int G = Rf_asInteger(ngroups);
  SEXP pl = PROTECT(Rf_coerceVector(ploidy, REALSXP));
  SEXP gid = PROTECT(Rf_coerceVector(groupIds, INTSXP));
  tpg_view* v = view_of(BM, rowInd, colInd, 0);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)XLENGTH(colInd), 2 * G));
  TPG_R_VIEW(v, tpg_grouped_alt_freq_dip_pseudo(ctx(), v, INTEGER(gid), G, REAL(pl), Rf_asLogical(as_counts), REAL(out)));
  UNPROTECT(3);
  return out;
}

/* grouped_missingness_cpp(BM, rowInd, colInd, groupIds, ngroups, ncores)   src/grouped_missingness_cpp.cpp:8-33 */
SEXP _tidypopgen_grouped_missingness_cpp(SEXP BM, SEXP rowInd, SEXP colInd, SEXP groupIds, SEXP ngroups, SEXP ncores) {
  (void)ncores;
  const int G = Rf_asInteger(ngroups);
  SEXP gid = PROTECT(Rf_coerceVector(groupIds, INTSXP));
  tpg_view* v = view_of(BM, rowInd, colInd, 0);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)XLENGTH(colInd), G));
  TPG_R_VIEW(v, tpg_grouped_missingness(ctx(), v, INTEGER(gid), G, REAL(out)));
  UNPROTECT(2);
  return out;
}

/* grouped_summaries_dip_pseudo_cpp(BM, rowInd, colInd, groupIds, ngroups, ploidy, ncores)
   src/grouped_summaries_dip_pseudo_cpp.cpp:11-63 -> list(freq_alt, freq_ref, n, het_obs), each m x G */
SEXP _tidypopgen_grouped_summaries_dip_pseudo_cpp(SEXP BM, SEXP rowInd, SEXP colInd, SEXP groupIds, SEXP ngroups,
                                                  SEXP ploidy, SEXP ncores) {
  (void)ncores;
  const int G = Rf_asInteger(ngroups);
  const int m = (int)XLENGTH(colInd);
  SEXP pl = PROTECT(Rf_coerceVector(ploidy, REALSXP));
  SEXP gid = PROTECT(Rf_coerceVector(groupIds, INTSXP));
  tpg_view* v = view_of(BM, rowInd, colInd, 0);
  SEXP mats[4];
  for (int k = 0; k < 4; k++) mats[k] = PROTECT(Rf_allocMatrix(REALSXP, m, G));
  TPG_R_VIEW(v, tpg_grouped_summaries_dip_pseudo(ctx(), v, INTEGER(gid), G, REAL(pl), REAL(mats[0]), REAL(mats[1]),
                                                 REAL(mats[2]), REAL(mats[3])));
  static const char* names[4] = {"freq_alt", "freq_ref", "n", "het_obs"}; /* :59-62 */
  SEXP out = named_list(4, names, mats);
  UNPROTECT(6);
  return out;
}

/* gt_ind_hetero(BM, rowInd, colInd, ncores)   src/gt_ind_hetero.cpp:11-42 -> integer 2 x n {n_het; n_na} */
SEXP _tidypopgen_gt_ind_hetero(SEXP BM, SEXP rowInd, SEXP colInd, SEXP ncores) {
  (void)ncores;
  tpg_view* v = view_of(BM, rowInd, colInd, 0);
  SEXP out = PROTECT(Rf_allocMatrix(INTSXP, 2, (int)XLENGTH(rowInd)));
  TPG_R_VIEW(v, tpg_gt_ind_hetero(ctx(), v, INTEGER(out)));
  UNPROTECT(1);
  return out;
}

/* gt_pi_diploid(BM, rowInd, colInd, ncores)   src/gt_pi_diploid.cpp:7-38 */
SEXP _tidypopgen_gt_pi_diploid(SEXP BM, SEXP rowInd, SEXP colInd, SEXP ncores) {
  (void)ncores;
  tpg_view* v = view_of(BM, rowInd, colInd, 0);
  SEXP out = PROTECT(Rf_allocVector(REALSXP, XLENGTH(colInd)));
  TPG_R_VIEW(v, tpg_gt_pi_diploid(ctx(), v, REAL(out)));
  UNPROTECT(1);
  return out;
}

/* gt_grouped_pi_diploid(BM, rowInd, colInd, groupIds, ngroups, ncores)   src/gt_grouped_pi_diploid.cpp:7-42 */
SEXP _tidypopgen_gt_grouped_pi_diploid(SEXP BM, SEXP rowInd, SEXP colInd, SEXP groupIds, SEXP ngroups, SEXP ncores) {
  (void)ncores;
  const int G = Rf_asInteger(ngroups);
  const int m = (int)XLENGTH(colInd);
  SEXP gid = PROTECT(Rf_coerceVector(groupIds, INTSXP));
  tpg_view* v = view_of(BM, rowInd, colInd, 0);
  SEXP mats[2];
  for (int k = 0; k < 2; k++) mats[k] = PROTECT(Rf_allocMatrix(REALSXP, m, G));
  TPG_R_VIEW(v, tpg_gt_grouped_pi_diploid(ctx(), v, INTEGER(gid), G, REAL(mats[0]), REAL(mats[1])));
  static const char* names[2] = {"pi", "n"};
  SEXP out = named_list(2, names, mats);
  UNPROTECT(3);
  return out;
}

/* ---- pairwise population Fst loops --------------------------------------------------------------------------- */

/* common body of the three loop functions.  pairwise_combn arrives as a 2 x P double matrix (NumericMatrix). */
static SEXP fst_loop(int method, SEXP pairwise_combn, SEXP n, SEXP freq_alt, SEXP freq_ref, SEXP het_obs, SEXP by_locus,
                     SEXP return_num_dem) {
  SEXP dim = Rf_getAttrib(n, R_DimSymbol);
  if (TYPEOF(n) != REALSXP || Rf_length(dim) != 2) Rf_error("n must be a numeric matrix");
  const int m = INTEGER(dim)[0], G = INTEGER(dim)[1];
  SEXP pc = PROTECT(Rf_coerceVector(pairwise_combn, INTSXP));
  const int P = (int)(XLENGTH(pc) / 2);
  int byl = Rf_asLogical(by_locus);
  const int rnd = Rf_asLogical(return_num_dem);
  const int want_a = byl || rnd;
  SEXP tot = PROTECT(Rf_allocVector(REALSXP, P));
  SEXP a = PROTECT(Rf_allocMatrix(REALSXP, want_a ? m : 0, want_a ? P : 0)); /* empty 0 x 0 when not asked, :17-21 */
  SEXP b = PROTECT(Rf_allocMatrix(REALSXP, rnd ? m : 0, rnd ? P : 0));
  TPG_R(tpg_pairwise_fst_loop(ctx(), method, INTEGER(pc), P, m, G, REAL(n), REAL(freq_alt),
                              freq_ref == R_NilValue ? NULL : REAL(freq_ref), het_obs == R_NilValue ? NULL : REAL(het_obs),
                              byl, rnd, REAL(tot), want_a ? REAL(a) : NULL, rnd ? REAL(b) : NULL));
  SEXP out;
  if (!rnd) { /* :54-60 */
    static const char* names[2] = {"fst_locus", "fst_tot"};
    SEXP vals[2] = {a, tot};
    out = named_list(2, names, vals);
  } else {
    static const char* names[2] = {"Fst_by_locus_num", "Fst_by_locus_den"};
    SEXP vals[2] = {a, b};
    out = named_list(2, names, vals);
  }
  UNPROTECT(4);
  return out;
}

/* pairwise_fst_hudson_loop(pairwise_combn, n, freq_alt, freq_ref, by_locus, return_num_dem)
   src/pairwise_fst_hudson_loop.cpp:5-63 */
SEXP _tidypopgen_pairwise_fst_hudson_loop(SEXP pairwise_combn, SEXP n, SEXP freq_alt, SEXP freq_ref, SEXP by_locus,
                                          SEXP return_num_dem) {
  return fst_loop(TPG_FST_HUDSON, pairwise_combn, n, freq_alt, freq_ref, R_NilValue, by_locus, return_num_dem);
}

/* pairwise_fst_wc84_loop(pairwise_combn, n, freq_alt, het_obs, by_locus, return_num_dem)
   src/pairwise_fst_wc84_loop.cpp:5-121 */
SEXP _tidypopgen_pairwise_fst_wc84_loop(SEXP pairwise_combn, SEXP n, SEXP freq_alt, SEXP het_obs, SEXP by_locus,
                                        SEXP return_num_dem) {
  return fst_loop(TPG_FST_WC84, pairwise_combn, n, freq_alt, R_NilValue, het_obs, by_locus, return_num_dem);
}

/* pairwise_fst_nei87_loop(pairwise_combn, n, het_obs, freq_alt, freq_ref, by_locus, return_num_dem)
   src/pairwise_fst_nei87_loop.cpp:5-115 */
SEXP _tidypopgen_pairwise_fst_nei87_loop(SEXP pairwise_combn, SEXP n, SEXP het_obs, SEXP freq_alt, SEXP freq_ref,
                                         SEXP by_locus, SEXP return_num_dem) {
  return fst_loop(TPG_FST_NEI87, pairwise_combn, n, freq_alt, freq_ref, het_obs, by_locus, return_num_dem);
}

/* ---- pairwise individual matrices: the per-block increment functions ----------------------------------------------
 * The R drivers (R/snp_ibs.R:59-82, R/snp_king.R:51-77, R/snp_allele_sharing.R:49-70) call these once per locus block
 * with the same FBM and the same two N x N double FBMs.
 * Default: every call uploads the columns of its block, accumulates, and adds the sums to k / k2 before it returns.
 * TPG_RSHIM_DEFERRED=1: the accumulators stay in HBM across the calls and the sums reach k / k2 when
 * _tidypopgen_tpg_flush is called.  The scratch matrices the reference fills are not touched. */

/* increment_ibs_counts(k, k2, genotype0, genotype1, genotype2, BM, rowInd, colInd)   src/snp_ibs.cpp:22-74 */
SEXP _tidypopgen_increment_ibs_counts(SEXP k, SEXP k2, SEXP g0, SEXP g1, SEXP g2, SEXP BM, SEXP rowInd, SEXP colInd) {
  (void)g0; (void)g1; (void)g2;
  g_call++;
  mapped_file* f = genotype_fbm(BM);
  const int64_t n = (int64_t)XLENGTH(rowInd);
  TPG_R(tpg_increment_ibs_counts(ctx(), double_fbm(k, n), double_fbm(k2, n), (const uint8_t*)f->map, f->nrow, f->ncol,
                                 INTEGER(rowInd), n, INTEGER(colInd), (int64_t)XLENGTH(colInd)));
  after_increment();
  return R_NilValue;
}

/* increment_king_numerator(k, n_Aa_i, genotype0, genotype1, genotype2, genotype_valid, BM, rowInd, colInd)
   src/snp_king.cpp:21-74 */
SEXP _tidypopgen_increment_king_numerator(SEXP k, SEXP n_Aa_i, SEXP g0, SEXP g1, SEXP g2, SEXP gv, SEXP BM, SEXP rowInd,
                                          SEXP colInd) {
  (void)g0; (void)g1; (void)g2; (void)gv;
  g_call++;
  mapped_file* f = genotype_fbm(BM);
  const int64_t n = (int64_t)XLENGTH(rowInd);
  TPG_R(tpg_increment_king_numerator(ctx(), double_fbm(k, n), double_fbm(n_Aa_i, n), (const uint8_t*)f->map, f->nrow,
                                     f->ncol, INTEGER(rowInd), n, INTEGER(colInd), (int64_t)XLENGTH(colInd)));
  after_increment();
  return R_NilValue;
}

/* increment_as_counts(k, k2, na_mat, dos_mat, BM, rowInd, colInd)   src/snp_as.cpp:22-67
 * Quirk Q1: the reference adds +1 to every numerator for a block one column narrower than the scratch
 * matrices.  Off by default (the intended value, what the reference's own test asserts); TPG_EMULATE_AS_PAD_QUIRK=1
 * reproduces the reference binary bit for bit. */
SEXP _tidypopgen_increment_as_counts(SEXP k, SEXP k2, SEXP na_mat, SEXP dos_mat, SEXP BM, SEXP rowInd, SEXP colInd) {
  (void)na_mat;
  g_call++;
  mapped_file* f = genotype_fbm(BM);
  const int64_t n = (int64_t)XLENGTH(rowInd), m = (int64_t)XLENGTH(colInd);
  double* K = double_fbm(k, n);
  TPG_R(tpg_increment_as_counts(ctx(), K, double_fbm(k2, n), (const uint8_t*)f->map, f->nrow, f->ncol, INTEGER(rowInd),
                                n, INTEGER(colInd), m));
  const char* q = getenv("TPG_EMULATE_AS_PAD_QUIRK");
  if (q && q[0] == '1') {
    SEXP dim = Rf_getAttrib(dos_mat, R_DimSymbol);
    if (Rf_length(dim) == 2 && (int64_t)INTEGER(dim)[1] == m + 1) TPG_R(tpg_increment_as_note_narrow_block(ctx(), K, n));
  }
  after_increment();
  return R_NilValue;
}

/* tpg_flush(): under TPG_RSHIM_DEFERRED=1 the one line the three pairwise drivers gain after their block loop -- writes
 * the sums held in HBM into the k / k2 FBMs (one download of two N x N matrices per analysis); a no-op otherwise.  Not a
 * reference symbol. */
SEXP _tidypopgen_tpg_flush(void) {
  if (g_ctx) TPG_R(tpg_increment_flush(g_ctx));
  release_accumulators(1);
  return R_NilValue;
}

/* tpg_release(): drop the HBM copies and mappings (e.g. after the FBM was modified, or to free the GPU) */
SEXP _tidypopgen_tpg_release(void) {
  if (g_ctx) {
    TPG_R(tpg_increment_flush(g_ctx));
    TPG_R(tpg_resident_drop(g_ctx));
  }
  while (g_nfiles > 0) forget_file(g_nfiles - 1);
  free(g_files);
  g_files = NULL;
  g_nfiles = 0;
  return R_NilValue;
}

/* tpg_invalidate(BM): forget the HBM copy of this FBM (only TPG_RSHIM_CACHE=1 keeps one).  The caller's job after anything
 * that writes to the FBM (gt_impute_simple, gt_set_imputed ...): the tpgshim package exports it as tpg_invalidate() and does
 * NOT wrap the mutating functions itself; a write that leaves size and mtime alone is otherwise caught by the fingerprint */
SEXP _tidypopgen_tpg_invalidate(SEXP BM) {
  mapped_file* f = genotype_fbm(BM);
  if (f->dev) {
    tpg_fbm_free(f->dev);
    f->dev = NULL;
  }
  return R_NilValue;
}

/* ---- PCA projection -------------------------------------------------------------------------------------------- */

/* fbm256_prod_and_rowSumsSq(BM, ind_row, ind_col, center, scale, V)   src/fbm_prod_and_rowSumSq.cpp:10-47
   -> list(XV n x K, rowSumsSq n), unnamed as in the reference (:46) */
SEXP _tidypopgen_fbm256_prod_and_rowSumsSq(SEXP BM, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale, SEXP V) {
  SEXP dim = Rf_getAttrib(V, R_DimSymbol);
  if (TYPEOF(V) != REALSXP || Rf_length(dim) != 2) Rf_error("V must be a numeric matrix");
  const int K = INTEGER(dim)[1];
  if ((R_xlen_t)INTEGER(dim)[0] != XLENGTH(ind_col)) Rf_error("Incompatibility between dimensions."); /* myassert_size, :23 */
  SEXP ce = PROTECT(Rf_coerceVector(center, REALSXP));
  SEXP sc = PROTECT(Rf_coerceVector(scale, REALSXP));
  tpg_view* v = view_of(BM, ind_row, ind_col, 0);
  SEXP XV = PROTECT(Rf_allocMatrix(REALSXP, (int)XLENGTH(ind_row), K));
  SEXP rss = PROTECT(Rf_allocVector(REALSXP, XLENGTH(ind_row)));
  TPG_R_VIEW(v, tpg_fbm256_prod_and_rowSumsSq(ctx(), v, REAL(ce), REAL(sc), REAL(V), K, REAL(XV), REAL(rss)));
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 2));
  SET_VECTOR_ELT(out, 0, XV);
  SET_VECTOR_ELT(out, 1, rss);
  UNPROTECT(5);
  return out;
}

/* ---- whole analyses on all the GPUs of the node (additions; not reference symbols) ---------------------------------
 * The per-block entry points above are literal drop-ins and run on one GPU.  An R session is ONE process, so the way to
 * the other GPUs is one call per analysis: the library gives every device a share of colInd (its own upload, pack and
 * sweep on a host thread per device) and exchanges what is additive over loci itself (RCCL).  These replace the BODY of
 * the R drivers -- snp_ibs / snp_king / snp_allele_sharing / pairwise_grm (R/snp_ibs.R:42-104 ...), pairwise_pop_fst's
 * numeric part (R/pairwise_pop_fst.R:116-161), loci_alt_freq on a grouped tibble (R/loci_alt_freq.R:174-197),
 * gt_pca_partialSVD's big_SVD call (R/gt_pca_partialSVD.R:82-89) -- INTEGRATION.md 2c shows the R side.
 * Devices: TPG_DEVICES (a count; default = all visible). */
static tpg_multi* g_multi = NULL;

static tpg_multi* multi(void) {
  if (!g_multi) {
    int ndev = 0;
    const char* e = getenv("TPG_DEVICES");
    if (e) ndev = atoi(e);
    else TPG_R(tpg_device_count(&ndev));
    if (ndev < 1) Rf_error("tidypopgen (GPU): no HIP device");
    TPG_R(tpg_multi_create(ndev, NULL, &g_multi));
  }
  return g_multi;
}

static void check_ind(SEXP rowInd, SEXP colInd) {
  if (TYPEOF(rowInd) != INTSXP || TYPEOF(colInd) != INTSXP) Rf_error("rowInd / colInd must be integer vectors");
}

/* tpg_snp_pairwise(BM, rowInd, colInd, adjusted_counts, which) -> list(ibs, king, allele_sharing, grm), each n x n or NULL.
 * which: integer mask of the matrices wanted (1 ibs, 2 king, 4 allele_sharing, 8 grm; NULL = all four).  Only the
 * cross-products those need are accumulated (tpg_multi_pairwise): 2 of 5 for the GRM alone, 4 for KING + GRM. */
SEXP _tidypopgen_tpg_snp_pairwise(SEXP BM, SEXP rowInd, SEXP colInd, SEXP adjusted_counts, SEXP which) {
  check_ind(rowInd, colInd);
  mapped_file* f = genotype_fbm(BM);
  const uint8_t* bytes = (const uint8_t*)f->map;
  const int64_t nrow = f->nrow, ncol = f->ncol;
  const int n = (int)XLENGTH(rowInd);
  const int want = which == R_NilValue ? 15 : Rf_asInteger(which);
  if (want < 1 || want > 15) Rf_error("tidypopgen (HIP): which must be a mask of 1 (ibs), 2 (king), 4 (allele_sharing), 8 (grm)");
  SEXP mats[4];
  for (int k = 0; k < 4; k++) mats[k] = PROTECT((want >> k) & 1 ? Rf_allocMatrix(REALSXP, n, n) : R_NilValue);
  TPG_R(tpg_multi_pairwise(multi(), bytes, nrow, ncol, INTEGER(rowInd), n, INTEGER(colInd), (int64_t)XLENGTH(colInd),
                           Rf_asLogical(adjusted_counts) ? TPG_IBS_ADJUSTED_COUNTS : TPG_IBS_PROPORTION,
                           mats[0] != R_NilValue ? REAL(mats[0]) : NULL, mats[1] != R_NilValue ? REAL(mats[1]) : NULL,
                           mats[2] != R_NilValue ? REAL(mats[2]) : NULL, mats[3] != R_NilValue ? REAL(mats[3]) : NULL));
  static const char* names[4] = {"ibs", "king", "allele_sharing", "grm"};
  SEXP out = named_list(4, names, mats);
  UNPROTECT(4);
  return out;
}

/* tpg_grouped_alt_freq(BM, rowInd, colInd, groupIds, ngroups, ploidy, as_counts) -> m x 2G (groupIds NULL: m x 2) */
SEXP _tidypopgen_tpg_grouped_alt_freq(SEXP BM, SEXP rowInd, SEXP colInd, SEXP groupIds, SEXP ngroups, SEXP ploidy,
                                      SEXP as_counts) {
  check_ind(rowInd, colInd);
  mapped_file* f = genotype_fbm(BM);
  const int grouped = groupIds != R_NilValue;
  const int G = grouped ? Rf_asInteger(ngroups) : 0;
  SEXP pl = PROTECT(Rf_coerceVector(ploidy, REALSXP));
  SEXP gid = PROTECT(grouped ? Rf_coerceVector(groupIds, INTSXP) : R_NilValue);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)XLENGTH(colInd), grouped ? 2 * G : 2));
  TPG_R(tpg_multi_grouped_alt_freq(multi(), (const uint8_t*)f->map, f->nrow, f->ncol, INTEGER(rowInd), (int64_t)XLENGTH(rowInd),
                                   INTEGER(colInd), (int64_t)XLENGTH(colInd), code256_of(BM), grouped ? INTEGER(gid) : NULL, G,
                                   REAL(pl), Rf_asLogical(as_counts), REAL(out)));
  UNPROTECT(3);
  return out;
}

/* tpg_pairwise_pop_fst(BM, rowInd, colInd, groupIds, ngroups, ploidy, method, pairwise_combn, by_locus, return_num_dem)
 * method: 0 Hudson, 1 Nei87, 2 WC84.  Same list as the three loop functions return. */
SEXP _tidypopgen_tpg_pairwise_pop_fst(SEXP BM, SEXP rowInd, SEXP colInd, SEXP groupIds, SEXP ngroups, SEXP ploidy, SEXP method,
                                      SEXP pairwise_combn, SEXP by_locus, SEXP return_num_dem) {
  check_ind(rowInd, colInd);
  mapped_file* f = genotype_fbm(BM);
  const int G = Rf_asInteger(ngroups), m = (int)XLENGTH(colInd);
  SEXP pl = PROTECT(Rf_coerceVector(ploidy, REALSXP));
  SEXP gid = PROTECT(Rf_coerceVector(groupIds, INTSXP));
  SEXP pc = PROTECT(Rf_coerceVector(pairwise_combn, INTSXP));
  const int P = (int)(XLENGTH(pc) / 2), rnd = Rf_asLogical(return_num_dem), want_a = Rf_asLogical(by_locus) || rnd;
  SEXP tot = PROTECT(Rf_allocVector(REALSXP, P));
  SEXP a = PROTECT(Rf_allocMatrix(REALSXP, want_a ? m : 0, want_a ? P : 0));
  SEXP b = PROTECT(Rf_allocMatrix(REALSXP, rnd ? m : 0, rnd ? P : 0));
  TPG_R(tpg_multi_pop_fst(multi(), (const uint8_t*)f->map, f->nrow, f->ncol, INTEGER(rowInd), (int64_t)XLENGTH(rowInd),
                          INTEGER(colInd), m, code256_of(BM), INTEGER(gid), G, REAL(pl), Rf_asInteger(method), INTEGER(pc), P,
                          want_a, rnd, REAL(tot), want_a ? REAL(a) : NULL, rnd ? REAL(b) : NULL));
  SEXP out;
  if (!rnd) {
    static const char* names[2] = {"fst_locus", "fst_tot"};
    SEXP vals[2] = {a, tot};
    out = named_list(2, names, vals);
  } else {
    static const char* names[2] = {"Fst_by_locus_num", "Fst_by_locus_den"};
    SEXP vals[2] = {a, b};
    out = named_list(2, names, vals);
  }
  UNPROTECT(6);
  return out;
}

/* tpg_pca_partial_svd(BM, rowInd, colInd, k) -> list(d, u, v, center, scale, square_frobenius): what big_SVD returns to
 * gt_pca_partialSVD (R/gt_pca_partialSVD.R:82-105) plus the squared Frobenius norm of R/square_frobenius.R */
SEXP _tidypopgen_tpg_pca_partial_svd(SEXP BM, SEXP rowInd, SEXP colInd, SEXP k) {
  check_ind(rowInd, colInd);
  mapped_file* f = genotype_fbm(BM);
  const int n = (int)XLENGTH(rowInd), m = (int)XLENGTH(colInd), K = Rf_asInteger(k);
  SEXP vals[6];
  vals[0] = PROTECT(Rf_allocVector(REALSXP, K));
  vals[1] = PROTECT(Rf_allocMatrix(REALSXP, n, K));
  vals[2] = PROTECT(Rf_allocMatrix(REALSXP, m, K));
  vals[3] = PROTECT(Rf_allocVector(REALSXP, m));
  vals[4] = PROTECT(Rf_allocVector(REALSXP, m));
  vals[5] = PROTECT(Rf_allocVector(REALSXP, 1));
  TPG_R(tpg_multi_pca_partial_svd(multi(), (const uint8_t*)f->map, f->nrow, f->ncol, INTEGER(rowInd), n, INTEGER(colInd), m,
                                  code256_of(BM), K, REAL(vals[0]), REAL(vals[1]), REAL(vals[2]), REAL(vals[3]), REAL(vals[4]),
                                  REAL(vals[5])));
  static const char* names[6] = {"d", "u", "v", "center", "scale", "square_frobenius"};
  SEXP out = named_list(6, names, vals);
  UNPROTECT(6);
  return out;
}

/* ---- registration ------------------------------------------------------------------------------------------------
 * Same names and arities as the reference's table (src/RcppExports.cpp:348-371).  These rows replace the rows of the
 * same name there; the other rows of that table (compute_np_mn, the HWE functions, the VCF / packedancestry readers,
 * write_to_FBM) keep pointing at the reference's own C++. */
const R_CallMethodDef tpg_rshim_entries[] = {
    {"_tidypopgen_alt_freq_dip_pseudo_cpp", (DL_FUNC)&_tidypopgen_alt_freq_dip_pseudo_cpp, 6},
    {"_tidypopgen_fbm256_prod_and_rowSumsSq", (DL_FUNC)&_tidypopgen_fbm256_prod_and_rowSumsSq, 6},
    {"_tidypopgen_grouped_alt_freq_dip_pseudo_cpp", (DL_FUNC)&_tidypopgen_grouped_alt_freq_dip_pseudo_cpp, 8},
    {"_tidypopgen_grouped_missingness_cpp", (DL_FUNC)&_tidypopgen_grouped_missingness_cpp, 6},
    {"_tidypopgen_grouped_summaries_dip_pseudo_cpp", (DL_FUNC)&_tidypopgen_grouped_summaries_dip_pseudo_cpp, 7},
    {"_tidypopgen_gt_grouped_pi_diploid", (DL_FUNC)&_tidypopgen_gt_grouped_pi_diploid, 6},
    {"_tidypopgen_gt_ind_hetero", (DL_FUNC)&_tidypopgen_gt_ind_hetero, 4},
    {"_tidypopgen_gt_pi_diploid", (DL_FUNC)&_tidypopgen_gt_pi_diploid, 4},
    {"_tidypopgen_pairwise_fst_hudson_loop", (DL_FUNC)&_tidypopgen_pairwise_fst_hudson_loop, 6},
    {"_tidypopgen_pairwise_fst_nei87_loop", (DL_FUNC)&_tidypopgen_pairwise_fst_nei87_loop, 7},
    {"_tidypopgen_pairwise_fst_wc84_loop", (DL_FUNC)&_tidypopgen_pairwise_fst_wc84_loop, 6},
    {"_tidypopgen_increment_as_counts", (DL_FUNC)&_tidypopgen_increment_as_counts, 7},
    {"_tidypopgen_increment_ibs_counts", (DL_FUNC)&_tidypopgen_increment_ibs_counts, 8},
    {"_tidypopgen_increment_king_numerator", (DL_FUNC)&_tidypopgen_increment_king_numerator, 9},
    /* additions (not in the reference): */
    {"_tidypopgen_tpg_flush", (DL_FUNC)&_tidypopgen_tpg_flush, 0},
    {"_tidypopgen_tpg_release", (DL_FUNC)&_tidypopgen_tpg_release, 0},
    {"_tidypopgen_tpg_invalidate", (DL_FUNC)&_tidypopgen_tpg_invalidate, 1},
    {"_tidypopgen_tpg_snp_pairwise", (DL_FUNC)&_tidypopgen_tpg_snp_pairwise, 5},
    {"_tidypopgen_tpg_grouped_alt_freq", (DL_FUNC)&_tidypopgen_tpg_grouped_alt_freq, 7},
    {"_tidypopgen_tpg_pairwise_pop_fst", (DL_FUNC)&_tidypopgen_tpg_pairwise_pop_fst, 10},
    {"_tidypopgen_tpg_pca_partial_svd", (DL_FUNC)&_tidypopgen_tpg_pca_partial_svd, 4},
    {NULL, NULL, 0}};

#ifdef TPG_RSHIM_STANDALONE
/* The shim as a package of its own (useDynLib(tpgshim, .registration = TRUE)): used to try the GPU path beside an
 * unmodified tidypopgen by assigning these functions over tidypopgen's internal wrappers. */
void R_init_tpgshim(DllInfo* dll) {
  R_registerRoutines(dll, NULL, tpg_rshim_entries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
#endif

void R_unload_tpgshim(DllInfo* dll) {
  (void)dll;
  _tidypopgen_tpg_release();
  if (g_multi) {
    tpg_multi_destroy(g_multi);
    g_multi = NULL;
  }
  if (g_ctx) {
    tpg_ctx_destroy(g_ctx);
    g_ctx = NULL;
  }
}
