/* tests/host/tpg_stub.c -- NOT the library: a host-only stand-in for the few libtpg_hip.so entry points shim/tpg_rshim.c
 * calls, so that the shim's own logic (the table of mapped files, the opt-in HBM cache and its invalidation, the per-call
 * column uploads with their index rebasing, accumulator mappings kept across a block loop, the `which` mask) can run
 * under -fsanitize=address,undefined without a GPU (tests/test_host_sanitizers.py).  Views are plain n x m code
 * matrices; alt_freq is computed for real (diploids), so the test can compare what comes back through the shim with
 * numpy; the increment_* stand-ins add block sizes to K / K2 and touch every element of both. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tpg.h"

struct tpg_ctx { int device; int defer; };
struct tpg_fbm { uint8_t* bytes; int64_t nrow, ncol; };
struct tpg_view { uint8_t* codes; int64_t n, m; };
struct tpg_multi { int ndev; };

static char g_err[256] = "";
int g_stub_fbm_alive = 0, g_stub_view_alive = 0, g_stub_uploads = 0; /* leak / traffic counters the test reads */
int64_t g_stub_bytes_uploaded = 0;

const char* tpg_last_error(void) { return g_err; }
int tpg_device_count(int* count) { *count = 1; return TPG_OK; }
int tpg_host_bind_near_device(int device, int* node) { (void)device; if (node) *node = -1; return TPG_OK; }
int tpg_ctx_create(int device, tpg_ctx** out) { *out = (tpg_ctx*)calloc(1, sizeof(tpg_ctx)); (*out)->device = device; return TPG_OK; }
void tpg_ctx_destroy(tpg_ctx* ctx) { free(ctx); }

int tpg_fbm_from_host(tpg_ctx* ctx, const uint8_t* bytes, int64_t nrow, int64_t ncol, tpg_fbm** out) {
  (void)ctx;
  tpg_fbm* f = (tpg_fbm*)malloc(sizeof(tpg_fbm));
  f->bytes = (uint8_t*)malloc((size_t)nrow * (size_t)ncol);
  memcpy(f->bytes, bytes, (size_t)nrow * (size_t)ncol); /* reads exactly what the shim says is there */
  f->nrow = nrow;
  f->ncol = ncol;
  g_stub_fbm_alive++;
  g_stub_uploads++;
  g_stub_bytes_uploaded += nrow * ncol;
  *out = f;
  return TPG_OK;
}
int tpg_fbm_open_bk(tpg_ctx* ctx, const char* path, int64_t nrow, int64_t ncol, tpg_fbm** out) {
  FILE* fp = fopen(path, "rb");
  if (!fp) { snprintf(g_err, sizeof(g_err), "cannot open %s", path); return TPG_EINVAL; }
  uint8_t* buf = (uint8_t*)malloc((size_t)nrow * (size_t)ncol);
  const size_t got = fread(buf, 1, (size_t)nrow * (size_t)ncol, fp);
  fclose(fp);
  int rc = got == (size_t)nrow * (size_t)ncol ? tpg_fbm_from_host(ctx, buf, nrow, ncol, out) : TPG_EINVAL;
  free(buf);
  return rc;
}
void tpg_fbm_free(tpg_fbm* f) {
  if (!f) return;
  free(f->bytes);
  free(f);
  g_stub_fbm_alive--;
}

int tpg_view_create(tpg_ctx* ctx, const tpg_fbm* fbm, const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m,
                    const double* code256, tpg_view** out) {
  (void)ctx;
  if (!rowInd1) n = fbm->nrow;
  if (!colInd1) m = fbm->ncol;
  tpg_view* v = (tpg_view*)malloc(sizeof(tpg_view));
  v->codes = (uint8_t*)malloc((size_t)n * (size_t)m);
  v->n = n;
  v->m = m;
  for (int64_t j = 0; j < m; j++) {
    const int64_t c = colInd1 ? colInd1[j] - 1 : j;
    if (c < 0 || c >= fbm->ncol) { snprintf(g_err, sizeof(g_err), "colInd out of range"); free(v->codes); free(v); return TPG_EINVAL; }
    for (int64_t i = 0; i < n; i++) {
      const int64_t r = rowInd1 ? rowInd1[i] - 1 : i;
      if (r < 0 || r >= fbm->nrow) { snprintf(g_err, sizeof(g_err), "rowInd out of range"); free(v->codes); free(v); return TPG_EINVAL; }
      const uint8_t b = fbm->bytes[(size_t)r + (size_t)c * (size_t)fbm->nrow];
      uint8_t code = 3;
      if (!code256) code = b < 3 ? b : 3;
      else if (code256[b] == code256[b]) code = (uint8_t)code256[b];
      v->codes[(size_t)i + (size_t)j * (size_t)n] = code;
    }
  }
  g_stub_view_alive++;
  *out = v;
  return TPG_OK;
}
int tpg_view_create_from_host(tpg_ctx* ctx, const uint8_t* bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1, int64_t n,
                              const int32_t* colInd1, int64_t m, const double* code256, tpg_view** out) {
  tpg_fbm* f = NULL;
  if (tpg_fbm_from_host(ctx, bytes, nrow, ncol, &f) != TPG_OK) return TPG_EINVAL;
  const int rc = tpg_view_create(ctx, f, rowInd1, n, colInd1, m, code256, out);
  tpg_fbm_free(f);
  return rc;
}
void tpg_view_free(tpg_view* v) {
  if (!v) return;
  free(v->codes);
  free(v);
  g_stub_view_alive--;
}

/* src/alt_freq_dip_pseudo_cpp.cpp:21-57 for diploids */
int tpg_alt_freq_dip_pseudo(tpg_ctx* ctx, const tpg_view* v, const double* ploidy, int as_counts, double* out) {
  (void)ctx; (void)ploidy;
  for (int64_t j = 0; j < v->m; j++) {
    double alt = 0, valid = 0;
    for (int64_t i = 0; i < v->n; i++) {
      const uint8_t c = v->codes[(size_t)i + (size_t)j * (size_t)v->n];
      if (c < 3) { alt += c; valid += 2; }
    }
    out[j] = as_counts ? alt : alt / valid;
    out[j + v->m] = valid;
  }
  return TPG_OK;
}

static int increment(double* K, double* K2, int64_t n, int64_t m) {
  for (int64_t k = 0; k < n * n; k++) { K[k] += 0.0; K2[k] += 0.0; } /* both mappings are n x n doubles, writable */
  K[0] += (double)m;
  K2[n * n - 1] += (double)n;
  return TPG_OK;
}
int tpg_increment_defer(tpg_ctx* ctx, int on) { ctx->defer = on; return TPG_OK; }
int tpg_increment_ibs_counts(tpg_ctx* ctx, double* K, double* K2, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol,
                             const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m) {
  (void)ctx; (void)nrow; (void)ncol; (void)rowInd1;
  volatile uint8_t acc = 0;
  for (int64_t j = 0; j < m; j++) acc ^= fbm_bytes[(size_t)(colInd1[j] - 1) * (size_t)nrow]; /* the mapping covers the columns named */
  return increment(K, K2, n, m);
}
int tpg_increment_king_numerator(tpg_ctx* ctx, double* K, double* N, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol,
                                 const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m) {
  return tpg_increment_ibs_counts(ctx, K, N, fbm_bytes, nrow, ncol, rowInd1, n, colInd1, m);
}
int tpg_increment_as_counts(tpg_ctx* ctx, double* K, double* K2, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol,
                            const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m) {
  return tpg_increment_ibs_counts(ctx, K, K2, fbm_bytes, nrow, ncol, rowInd1, n, colInd1, m);
}
int tpg_increment_as_note_narrow_block(tpg_ctx* ctx, double* K, int64_t n) { (void)ctx; for (int64_t k = 0; k < n * n; k++) K[k] += 1.0; return TPG_OK; }
int tpg_increment_flush(tpg_ctx* ctx) { (void)ctx; return TPG_OK; }
int tpg_resident_drop(tpg_ctx* ctx) { (void)ctx; return TPG_OK; }

/* one process, "all GPUs": the stand-in fills what was asked for and leaves the rest alone */
int tpg_multi_create(int ndev, const int* devices, tpg_multi** out) { (void)devices; *out = (tpg_multi*)calloc(1, sizeof(tpg_multi)); (*out)->ndev = ndev; return TPG_OK; }
void tpg_multi_destroy(tpg_multi* mg) { free(mg); }
int tpg_multi_pairwise(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1, int64_t n,
                       const int32_t* colInd1, int64_t m, int ibs_type, double* ibs, double* king, double* allele_sharing,
                       double* grm) {
  (void)mg; (void)fbm_bytes; (void)nrow; (void)ncol; (void)rowInd1; (void)colInd1; (void)ibs_type;
  double* outs[4] = {ibs, king, allele_sharing, grm};
  for (int q = 0; q < 4; q++)
    if (outs[q])
      for (int64_t k = 0; k < n * n; k++) outs[q][k] = (double)(q + 1) * (double)m;
  return TPG_OK;
}

/* not exercised by the sanitizer test: present so that the shim links */
#define NOT_HERE(name) snprintf(g_err, sizeof(g_err), "tests/host/tpg_stub.c has no " name); return TPG_EUNSUPPORTED
int tpg_grouped_alt_freq_dip_pseudo(tpg_ctx* c, const tpg_view* v, const int32_t* g, int G, const double* p, int a, double* o) { (void)c; (void)v; (void)g; (void)G; (void)p; (void)a; (void)o; NOT_HERE("grouped_alt_freq"); }
int tpg_grouped_missingness(tpg_ctx* c, const tpg_view* v, const int32_t* g, int G, double* o) { (void)c; (void)v; (void)g; (void)G; (void)o; NOT_HERE("grouped_missingness"); }
int tpg_grouped_summaries_dip_pseudo(tpg_ctx* c, const tpg_view* v, const int32_t* g, int G, const double* p, double* a, double* b, double* n, double* h) { (void)c; (void)v; (void)g; (void)G; (void)p; (void)a; (void)b; (void)n; (void)h; NOT_HERE("grouped_summaries"); }
int tpg_gt_ind_hetero(tpg_ctx* c, const tpg_view* v, int32_t* o) { (void)c; (void)v; (void)o; NOT_HERE("gt_ind_hetero"); }
int tpg_gt_pi_diploid(tpg_ctx* c, const tpg_view* v, double* o) { (void)c; (void)v; (void)o; NOT_HERE("gt_pi_diploid"); }
int tpg_gt_grouped_pi_diploid(tpg_ctx* c, const tpg_view* v, const int32_t* g, int G, double* a, double* b) { (void)c; (void)v; (void)g; (void)G; (void)a; (void)b; NOT_HERE("gt_grouped_pi_diploid"); }
int tpg_pairwise_fst_loop(tpg_ctx* c, int me, const int32_t* p, int P, int64_t m, int G, const double* n, const double* fa, const double* fr, const double* h, int bl, int nd, double* t, double* a, double* b) { (void)c; (void)me; (void)p; (void)P; (void)m; (void)G; (void)n; (void)fa; (void)fr; (void)h; (void)bl; (void)nd; (void)t; (void)a; (void)b; NOT_HERE("pairwise_fst_loop"); }
int tpg_fbm256_prod_and_rowSumsSq(tpg_ctx* c, const tpg_view* v, const double* ce, const double* sc, const double* V, int K, double* XV, double* rss) { (void)c; (void)v; (void)ce; (void)sc; (void)V; (void)K; (void)XV; (void)rss; NOT_HERE("fbm256_prod_and_rowSumsSq"); }
int tpg_multi_grouped_alt_freq(tpg_multi* mg, const uint8_t* f, int64_t nr, int64_t nc, const int32_t* r, int64_t n, const int32_t* c, int64_t m, const double* code, const int32_t* g, int G, const double* p, int a, double* o) { (void)mg; (void)f; (void)nr; (void)nc; (void)r; (void)n; (void)c; (void)m; (void)code; (void)g; (void)G; (void)p; (void)a; (void)o; NOT_HERE("multi_grouped_alt_freq"); }
int tpg_multi_pop_fst(tpg_multi* mg, const uint8_t* f, int64_t nr, int64_t nc, const int32_t* r, int64_t n, const int32_t* c, int64_t m, const double* code, const int32_t* g, int G, const double* p, int me, const int32_t* pr, int P, int bl, int nd, double* t, double* a, double* b) { (void)mg; (void)f; (void)nr; (void)nc; (void)r; (void)n; (void)c; (void)m; (void)code; (void)g; (void)G; (void)p; (void)me; (void)pr; (void)P; (void)bl; (void)nd; (void)t; (void)a; (void)b; NOT_HERE("multi_pop_fst"); }
int tpg_multi_pca_partial_svd(tpg_multi* mg, const uint8_t* f, int64_t nr, int64_t nc, const int32_t* r, int64_t n, const int32_t* c, int64_t m, const double* code, int k, double* d, double* u, double* v, double* ce, double* sc, double* fro) { (void)mg; (void)f; (void)nr; (void)nc; (void)r; (void)n; (void)c; (void)m; (void)code; (void)k; (void)d; (void)u; (void)v; (void)ce; (void)sc; (void)fro; NOT_HERE("multi_pca_partial_svd"); }
