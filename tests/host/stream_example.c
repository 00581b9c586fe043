/* stream_example.c -- the streamed whole-analysis entry points of include/tpg.h called from plain C, the way INTEGRATION.md 3a shows
 * them (tests/test_gpu_stream.py::test_stream_api_from_c compiles this with gcc against the header, links libtpg_hip.so and
 * compares what it prints with the Python mirror's results on the same bytes).
 *   stream_example <file with nrow x ncol FBM bytes, column-major> <nrow> <ncol> <budget_bytes> <ngroups> <k>            */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tpg.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    if ((call) != TPG_OK) {                                                \
      fprintf(stderr, "%s failed: %s\n", #call, tpg_last_error());         \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static double sum(const double* x, size_t n) {
  double s = 0;
  for (size_t i = 0; i < n; i++)
    if (x[i] == x[i]) s += x[i];
  return s;
}

int main(int argc, char** argv) {
  if (argc != 7) return 2;
  const long long nrow = atoll(argv[2]), ncol = atoll(argv[3]);
  const size_t budget = (size_t)atoll(argv[4]);
  const int G = atoi(argv[5]), k = atoi(argv[6]);
  tpg_ctx* ctx;
  CHECK(tpg_ctx_create(0, &ctx));
  tpg_stream* st;
  CHECK(tpg_stream_open_bk(ctx, argv[1], nrow, ncol, budget, &st));
  double code_imp[256];
  for (int b = 0; b < 256; b++) code_imp[b] = b < 3 ? b : (b >= 4 && b < 7 ? b - 4 : 0.0 / 0.0);
  int32_t* gid = (int32_t*)malloc(sizeof(int32_t) * (size_t)nrow);
  for (long long i = 0; i < nrow; i++) gid[i] = (int32_t)(i % G);
  const int P = G * (G - 1) / 2;
  int32_t* pairs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)P);
  for (int a = 1, q = 0; a <= G; a++)
    for (int b = a + 1; b <= G; b++, q++) { pairs[2 * q] = a; pairs[2 * q + 1] = b; }
  double* ibs = (double*)malloc(8 * (size_t)nrow * (size_t)nrow);
  double* gaf = (double*)malloc(8 * (size_t)ncol * 2 * (size_t)G);
  double* fst = (double*)malloc(8 * (size_t)P);
  double *d = (double*)malloc(8 * (size_t)k), *u = (double*)malloc(8 * (size_t)nrow * (size_t)k), *v = (double*)malloc(8 * (size_t)ncol * (size_t)k);
  double *cen = (double*)malloc(8 * (size_t)ncol), *sca = (double*)malloc(8 * (size_t)ncol), fro = 0;
  tpg_stream_job job;
  memset(&job, 0, sizeof job);
  job.struct_size = sizeof job;
  job.ibs = ibs;
  job.code256 = NULL; /* raw bytes: 0 / 1 / 2 valid, the rest missing */
  job.groupIds0 = gid;
  job.ngroups = G;
  job.grouped_alt_freq = gaf;
  job.nfst = 1;
  job.fst_method[0] = TPG_FST_HUDSON;
  job.pairs1 = pairs;
  job.P = P;
  job.fst_tot[0] = fst;
  job.code256_pca = code_imp;
  job.k = k;
  job.d = d; job.u = u; job.v = v; job.center = cen; job.scale = sca; job.square_frobenius = &fro;
  tpg_stream_report rep;
  CHECK(tpg_stream_run(ctx, st, &job, &rep));
  printf("blocks %lld sweeps %d bytes_up %zu\n", (long long)rep.blocks, rep.sweeps, rep.bytes_up);
  printf("ibs_sum %.17g\n", sum(ibs, (size_t)nrow * (size_t)nrow));
  printf("gaf_sum %.17g\n", sum(gaf, (size_t)ncol * 2 * (size_t)G));
  printf("fst_sum %.17g\n", sum(fst, (size_t)P));
  printf("d");
  for (int q = 0; q < k; q++) printf(" %.17g", d[q]);
  printf("\nfro %.17g\ncenter_sum %.17g\n", fro, sum(cen, (size_t)ncol));
  /* a wrong struct size is refused, not read past */
  job.struct_size = sizeof job - 8;
  if (tpg_stream_run(ctx, st, &job, &rep) == TPG_OK) return 3;
  tpg_stream_close(st);
  tpg_ctx_destroy(ctx);
  printf("C_OK\n");
  return 0;
}
