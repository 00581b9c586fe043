// comm.hip -- SNP-block shards over the GPUs of one node: communicators and the collectives of the hot path.
//
// The locus axis is the reference's own block axis (R/snp_ibs.R:59-82 cuts colInd into blocks and sums the per-block
// N x N increments), so a shard is a contiguous range of loci and every quantity on the path is either a disjoint
// per-locus slice (no exchange) or additive over loci:
//     pairwise cross-products V, D, H, A   int32 slabs    reduce-scatter: rank r finishes band r of the tiles
//     Fst numerator / denominator sums     2 P doubles    all-reduce
//     PCA Gram matrix, Frobenius norm      FP64           all-reduce (the eigen step is replicated)
//     mean behind the GRM                  2 doubles      all-reduce
// One rank = one context = one GPU.  Two ways to get ranks:
//   * one process per GPU (torchrun / mpirun): tpg_comm_unique_id on rank 0, the launcher broadcasts the 128 bytes,
//     tpg_comm_init_rank everywhere;
//   * one process driving several GPUs (an R session): tpg_multi_create -> ncclCommInitAll, one host thread per
//     device inside every tpg_multi_* call.
// Transport is RCCL over xGMI, loaded at run time (dlopen: the library itself does not link against it, and a process
// that already holds a copy -- PyTorch ships one -- is not given a second).  A third kind of communicator takes an
// all-reduce callback through host memory instead: it exists so that the sharded code paths can be rehearsed where
// RCCL cannot run (several ranks sharing one GPU, gloo on CPUs) -- tests only, never the bench.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "common.h"

// ---------------------------------------------------------------------------
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.handle ? &api : nullptr;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (api.handle) break;
  }
  if (!api.handle) return nullptr;
#define RSYM(field, sym)                                                   \
  do {                                                                     \
    *(void**)(&api.field) = dlsym(api.handle, sym);                        \
    if (!api.field) { dlclose(api.handle); api.handle = nullptr; return nullptr; } \
  } while (0)
  RSYM(GetUniqueId, "ncclGetUniqueId");
  RSYM(CommInitRank, "ncclCommInitRank");
  RSYM(CommInitAll, "ncclCommInitAll");
  RSYM(CommDestroy, "ncclCommDestroy");
  RSYM(AllReduce, "ncclAllReduce");
  RSYM(ReduceScatter, "ncclReduceScatter");
  RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
  return &api;
}

#define TPG_RCCL(api, call)                                                                              \
  do {                                                                                                   \
    ncclResult_t _r = (call);                                                                            \
    if (_r != ncclSuccess) {                                                                             \
      tpg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, (api)->GetErrorString(_r));            \
      return TPG_EHIP;                                                                                   \
    }                                                                                                    \
  } while (0)

static_assert(sizeof(ncclUniqueId) == 128, "tpg_comm_unique_id hands out 128 bytes");

extern "C" int tpg_comm_unique_id(uint8_t* id128) {
  TPG_REQUIRE(id128, TPG_EINVAL, "null argument");
  RcclApi* api = rccl();
  TPG_REQUIRE(api, TPG_EHIP, "RCCL (librccl.so) could not be loaded: %s", dlerror() ? dlerror() : "not found");
  ncclUniqueId id;
  TPG_RCCL(api, api->GetUniqueId(&id));
  memcpy(id128, &id, 128);
  return TPG_OK;
}

extern "C" int tpg_comm_init_rank(tpg_ctx* ctx, int nranks, int rank, const uint8_t* id128, tpg_comm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, TPG_EINVAL, "bad rank %d of %d", rank, nranks);
  tpg_comm* c = new tpg_comm();
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  // a single rank exchanges nothing: no communicator, every collective is the identity (TPG_COMM_FORCE_RCCL=1 builds
  // a real one-rank RCCL communicator all the same: the rehearsal of the RCCL calls a one-GPU box allows)
  const bool force = getenv("TPG_COMM_FORCE_RCCL") && getenv("TPG_COMM_FORCE_RCCL")[0] == '1';
  if (nranks > 1 || force) {
    RcclApi* api = rccl();
    if (!api || (!id128 && nranks > 1)) {
      delete c;
      tpg_set_error(api ? "null unique id" : "RCCL (librccl.so) could not be loaded");
      return api ? TPG_EINVAL : TPG_EHIP;
    }
    ncclUniqueId id;
    if (id128) memcpy(&id, id128, 128);
    else if (api->GetUniqueId(&id) != ncclSuccess) { delete c; tpg_set_error("ncclGetUniqueId failed"); return TPG_EHIP; }
    ncclComm_t nc = nullptr;
    ncclResult_t r = api->CommInitRank(&nc, nranks, id, rank);  // the context's device is current (TpgEnter)
    if (r != ncclSuccess) {
      delete c;
      tpg_set_error("ncclCommInitRank(rank %d of %d): %s", rank, nranks, api->GetErrorString(r));
      return TPG_EHIP;
    }
    c->nccl = nc;
  }
  *out = c;
  return TPG_OK;
}

extern "C" int tpg_comm_init_host(tpg_ctx* ctx, int nranks, int rank,
                                  int (*allreduce)(void* user, void* buf, int64_t count, int dtype), void* user,
                                  tpg_comm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && out && (allreduce || nranks == 1), TPG_EINVAL, "null argument");
  TPG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, TPG_EINVAL, "bad rank %d of %d", rank, nranks);
  tpg_comm* c = new tpg_comm();
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  c->host_fn = allreduce;
  c->host_user = user;
  *out = c;
  return TPG_OK;
}

extern "C" void tpg_comm_destroy(tpg_comm* comm) {
  if (!comm) return;
  if (comm->nccl) {
    RcclApi* api = rccl();
    if (api) {
      TpgEnter _enter(comm->ctx);
      (void)hipStreamSynchronize(comm->ctx->stream);
      (void)api->CommDestroy((ncclComm_t)comm->nccl);
    }
  }
  delete comm;
}

extern "C" int tpg_comm_rank(const tpg_comm* comm) { return comm ? comm->rank : 0; }
extern "C" int tpg_comm_size(const tpg_comm* comm) { return comm ? comm->nranks : 1; }

// ---------------------------------------------------------------------------
// collectives on device memory, stream-ordered on the context's stream

static size_t dtype_size(int dtype) { return dtype == 0 ? sizeof(int32_t) : sizeof(double); }

static int host_allreduce(tpg_comm* comm, void* d_buf, int64_t count, int dtype, int64_t keep_off, int64_t keep_count) {
  // through host memory: device -> host, the caller's all-reduce, the part this rank keeps -> device
  const size_t es = dtype_size(dtype);
  std::vector<uint8_t> h((size_t)count * es);
  TPG_HIP(hipMemcpyAsync(h.data(), d_buf, h.size(), hipMemcpyDeviceToHost, comm->ctx->stream));
  TPG_HIP(hipStreamSynchronize(comm->ctx->stream));
  const int rc = comm->host_fn(comm->host_user, h.data(), count, dtype);
  TPG_REQUIRE(rc == 0, TPG_EHIP, "the host all-reduce callback failed (%d)", rc);
  TPG_HIP(hipMemcpyAsync((uint8_t*)d_buf + (size_t)keep_off * es, h.data() + (size_t)keep_off * es, (size_t)keep_count * es,
                         hipMemcpyHostToDevice, comm->ctx->stream));
  TPG_HIP(hipStreamSynchronize(comm->ctx->stream));
  return TPG_OK;
}

int tpg_comm_allreduce(tpg_comm* comm, void* d_buf, int64_t count, int dtype) {
  TPG_REQUIRE(comm && d_buf && count >= 0 && (dtype == 0 || dtype == 1), TPG_EINVAL, "bad all-reduce arguments");
  if ((comm->nranks == 1 && !comm->nccl) || count == 0) return TPG_OK;
  if (comm->host_fn) return host_allreduce(comm, d_buf, count, dtype, 0, count);
  RcclApi* api = rccl();
  TPG_REQUIRE(api && comm->nccl, TPG_EHIP, "communicator has no transport");
  TPG_RCCL(api, api->AllReduce(d_buf, d_buf, (size_t)count, dtype == 0 ? ncclInt32 : ncclFloat64, ncclSum,
                               (ncclComm_t)comm->nccl, comm->ctx->stream));
  return TPG_OK;
}

// d_buf holds nranks chunks of chunk_count int32; afterwards chunk `rank` holds the sum over the ranks of that chunk
// (the other chunks are left with this rank's own partial values)
int tpg_comm_reduce_scatter_i32(tpg_comm* comm, int32_t* d_buf, int64_t chunk_count) {
  TPG_REQUIRE(comm && d_buf && chunk_count >= 0, TPG_EINVAL, "bad reduce-scatter arguments");
  if ((comm->nranks == 1 && !comm->nccl) || chunk_count == 0) return TPG_OK;
  if (comm->host_fn)
    return host_allreduce(comm, d_buf, chunk_count * comm->nranks, 0, chunk_count * comm->rank, chunk_count);
  RcclApi* api = rccl();
  TPG_REQUIRE(api && comm->nccl, TPG_EHIP, "communicator has no transport");
  TPG_RCCL(api, api->ReduceScatter(d_buf, d_buf + (size_t)chunk_count * (size_t)comm->rank, (size_t)chunk_count, ncclInt32,
                                   ncclSum, (ncclComm_t)comm->nccl, comm->ctx->stream));
  return TPG_OK;
}

// sum a vector of doubles over the ranks, in place; buf may be host or device memory (Fst numerator / denominator
// sums, the N x N Gram matrix, the squared Frobenius norm)
extern "C" int tpg_comm_allreduce_f64(tpg_ctx* ctx, tpg_comm* comm, double* buf, int64_t count) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && comm && buf && count >= 0, TPG_EINVAL, "bad argument");
  TPG_REQUIRE(comm->ctx == ctx, TPG_EINVAL, "the communicator belongs to another context");
  if ((comm->nranks == 1 && !comm->nccl) || count == 0) return TPG_OK;
  ProfScope ps(ctx, "allreduce_f64");
  if (tpg_is_device_ptr(buf)) return tpg_comm_allreduce(comm, buf, count, 1);
  if (comm->host_fn) {
    const int rc = comm->host_fn(comm->host_user, buf, count, 1);
    TPG_REQUIRE(rc == 0, TPG_EHIP, "the host all-reduce callback failed (%d)", rc);
    return TPG_OK;
  }
  double* d = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d, sizeof(double) * (size_t)count));
  hipError_t e = hipMemcpyAsync(d, buf, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream);
  int rc = e == hipSuccess ? tpg_comm_allreduce(comm, d, count, 1) : TPG_EHIP;
  if (rc == TPG_OK) {
    e = hipMemcpyAsync(buf, d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  tpg_pfree(d);
  if (e != hipSuccess) { tpg_set_error("all-reduce staging: %s", hipGetErrorString(e)); return TPG_EHIP; }
  return rc;
}

// Contiguous locus range [begin, end) of `rank`: boundaries on multiples of 128 loci (the K-group width of the packed
// layouts; tpg_pairwise_accumulate wants aligned ranges), sizes differ by at most 128.
extern "C" int tpg_shard_loci(int64_t m_total, int nranks, int rank, int64_t* begin, int64_t* end) {
  TPG_REQUIRE(begin && end && m_total >= 0 && nranks >= 1 && rank >= 0 && rank < nranks, TPG_EINVAL, "bad argument");
  const int64_t groups = ceil_div(m_total, 128);
  const int64_t g0 = groups * rank / nranks, g1 = groups * (rank + 1) / nranks;
  *begin = std::min(g0 * 128, m_total);
  *end = std::min(g1 * 128, m_total);
  return TPG_OK;
}

// ---------------------------------------------------------------------------
// One process, several GPUs.
struct tpg_multi {
  int ndev = 0;
  std::vector<tpg_ctx*> ctx;
  std::vector<tpg_comm*> comm;
};

extern "C" void tpg_multi_destroy(tpg_multi* mg) {
  if (!mg) return;
  for (auto c : mg->comm) tpg_comm_destroy(c);
  for (auto c : mg->ctx) tpg_ctx_destroy(c);
  delete mg;
}

extern "C" int tpg_multi_create(int ndev, const int* devices, tpg_multi** out) {
  TPG_REQUIRE(out, TPG_EINVAL, "null out");
  TPG_REQUIRE(ndev >= 1 && ndev <= 64, TPG_EINVAL, "bad device count %d", ndev);
  tpg_multi* mg = new tpg_multi();
  mg->ndev = ndev;
  std::vector<int> devs((size_t)ndev);
  for (int i = 0; i < ndev; i++) devs[(size_t)i] = devices ? devices[i] : i;
  for (int i = 0; i < ndev; i++) {
    tpg_ctx* c = nullptr;
    int rc = tpg_ctx_create(devs[(size_t)i], &c);
    if (rc != TPG_OK) { tpg_multi_destroy(mg); return rc; }
    mg->ctx.push_back(c);
  }
  std::vector<ncclComm_t> nc((size_t)ndev, nullptr);
  if (ndev > 1) {
    RcclApi* api = rccl();
    if (!api) { tpg_multi_destroy(mg); tpg_set_error("RCCL (librccl.so) could not be loaded"); return TPG_EHIP; }
    ncclResult_t r = api->CommInitAll(nc.data(), ndev, devs.data());
    if (r != ncclSuccess) {
      tpg_multi_destroy(mg);
      tpg_set_error("ncclCommInitAll(%d devices): %s", ndev, api->GetErrorString(r));
      return TPG_EHIP;
    }
  }
  for (int i = 0; i < ndev; i++) {
    tpg_comm* c = new tpg_comm();
    c->ctx = mg->ctx[(size_t)i];
    c->nranks = ndev;
    c->rank = i;
    c->nccl = nc[(size_t)i];
    mg->comm.push_back(c);
  }
  *out = mg;
  return TPG_OK;
}

extern "C" int tpg_multi_ndev(const tpg_multi* mg) { return mg ? mg->ndev : 0; }
extern "C" tpg_ctx* tpg_multi_ctx(tpg_multi* mg, int i) { return mg && i >= 0 && i < mg->ndev ? mg->ctx[(size_t)i] : nullptr; }
extern "C" tpg_comm* tpg_multi_comm(tpg_multi* mg, int i) { return mg && i >= 0 && i < mg->ndev ? mg->comm[(size_t)i] : nullptr; }

// Run fn(rank) on one host thread per device (HIP's current device and this library's error string are per thread);
// the first failure is reported on the calling thread.
template <typename F>
static int multi_run(tpg_multi* mg, F fn) {
  std::vector<int> rcs((size_t)mg->ndev, TPG_OK);
  std::vector<std::string> errs((size_t)mg->ndev);
  auto body = [&](int r) {
    rcs[(size_t)r] = fn(r);
    if (rcs[(size_t)r] != TPG_OK) errs[(size_t)r] = tpg_last_error();
  };
  if (mg->ndev == 1) {
    body(0);
  } else {
    std::vector<std::thread> th;
    for (int r = 0; r < mg->ndev; r++) th.emplace_back(body, r);
    for (auto& t : th) t.join();
  }
  for (int r = 0; r < mg->ndev; r++)
    if (rcs[(size_t)r] != TPG_OK) {
      tpg_set_error("device %d: %s", mg->ctx[(size_t)r]->device, errs[(size_t)r].c_str());
      return rcs[(size_t)r];
    }
  return TPG_OK;
}

// snp_ibs / snp_king / snp_allele_sharing / pairwise_grm of one host FBM on all the devices of `mg`
// (R/snp_ibs.R:42-104, R/snp_king.R:32-103, R/snp_allele_sharing.R:33-82, R/pairwise_grm.R:30-51 -- their block loops
// become: every device takes a contiguous share of colInd, uploads just those columns, packs, accumulates; one
// reduce-scatter; every device finishes its band of tiles and writes it straight into the caller's host matrices).
extern "C" int tpg_multi_pairwise(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol,
                                  const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m, int ibs_type,
                                  double* ibs, double* king, double* allele_sharing, double* grm) {
  TPG_REQUIRE(mg && fbm_bytes, TPG_EINVAL, "null argument");
  TPG_REQUIRE(nrow > 0 && ncol > 0, TPG_EINVAL, "empty FBM");
  if (!rowInd1) n = nrow;
  if (!colInd1) m = ncol;
  TPG_REQUIRE(n > 0 && m > 0, TPG_EINVAL, "empty view");
  if (colInd1)
    for (int64_t j = 0; j < m; j++)
      TPG_REQUIRE(colInd1[j] >= 1 && colInd1[j] <= ncol, TPG_EINVAL, "colInd[%lld] = %d out of [1,%lld]", (long long)j,
                  colInd1[j], (long long)ncol);
  struct Rank { tpg_fbm* f = nullptr; tpg_view* v = nullptr; tpg_pairwise* pw = nullptr; };
  std::vector<Rank> st((size_t)mg->ndev);
  // phase 1, no exchange: upload, pack and accumulate this device's loci.  The phases are separate thread teams so
  // that a failure on one device (out of memory, a bad index) is known to all before anyone enters a collective --
  // a rank that skipped the reduce-scatter would leave the others waiting in it for ever.
  int rc = multi_run(mg, [&](int r) -> int {
    tpg_ctx* ctx = mg->ctx[(size_t)r];
    TpgEnter _enter(ctx);
    Rank& me = st[(size_t)r];
    int64_t j0, j1;
    TPG_TRY(tpg_shard_loci(m, mg->ndev, r, &j0, &j1));
    TPG_TRY(tpg_pairwise_create_sharded(ctx, mg->comm[(size_t)r], n, &me.pw));
    if (j1 <= j0) return TPG_OK;  // more devices than 128-locus groups: nothing of its own, still takes part below
    // this device's loci: a contiguous byte range of the column-major FBM when colInd is the identity, else the
    // covering range of its share of colInd with the indices rebased onto it
    int64_t c0 = j0, c1 = j1;  // 0-based FBM columns [c0, c1)
    std::vector<int32_t> cols;
    if (colInd1) {
      int32_t lo = colInd1[j0], hi = colInd1[j0];
      for (int64_t j = j0; j < j1; j++) { lo = std::min(lo, colInd1[j]); hi = std::max(hi, colInd1[j]); }
      c0 = lo - 1;
      c1 = hi;
      cols.resize((size_t)(j1 - j0));
      for (int64_t j = j0; j < j1; j++) cols[(size_t)(j - j0)] = colInd1[j] - (int32_t)c0;
    }
    TPG_TRY(tpg_fbm_from_host(ctx, fbm_bytes + (size_t)c0 * (size_t)nrow, nrow, c1 - c0, &me.f));
    TPG_TRY(tpg_view_create(ctx, me.f, rowInd1, n, colInd1 ? cols.data() : nullptr, j1 - j0, nullptr /* raw bytes */, &me.v));
    return tpg_pairwise_accumulate(ctx, me.pw, me.v, 0, -1);
  });
  // phase 2: one reduce-scatter, then every device finishes its band and writes it into the caller's matrices
  if (rc == TPG_OK)
    rc = multi_run(mg, [&](int r) -> int {
      tpg_ctx* ctx = mg->ctx[(size_t)r];
      tpg_comm* comm = mg->comm[(size_t)r];
      TpgEnter _enter(ctx);
      TPG_TRY(tpg_pairwise_reduce(ctx, comm, st[(size_t)r].pw));
      return tpg_pairwise_epilogues_sharded(ctx, comm, st[(size_t)r].pw, ibs_type, m, ibs, king, allele_sharing, grm);
    });
  std::string err = rc == TPG_OK ? "" : tpg_last_error();
  for (int r = 0; r < mg->ndev; r++) {
    TpgEnter _enter(mg->ctx[(size_t)r]);
    tpg_pairwise_free(st[(size_t)r].pw);
    tpg_view_free(st[(size_t)r].v);
    tpg_fbm_free(st[(size_t)r].f);
  }
  if (rc != TPG_OK) tpg_set_error("%s", err.c_str());
  return rc;
}
