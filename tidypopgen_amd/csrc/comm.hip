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

#include <condition_variable>
#include <mutex>
#include <string>
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
  ncclResult_t (*AllToAllv)(const void*, const size_t*, const size_t*, void*, const size_t*, const size_t*, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;
static std::string g_rccl_path;  // the name the library was loaded by
static std::string g_rccl_why;  // why the load failed (dlerror() text, read once: a second dlerror() returns NULL)
static std::once_flag g_rccl_once;

static void rccl_load() {
  RcclApi& api = g_rccl;
  // TPG_RCCL_LIBRARY=<path>: load THAT library and nothing else (tests/host/mock_rccl.cpp: the nccl* call sites below with
  // N > 1 ranks on a one-GPU box, where RCCL refuses a device listed twice; tpg_comm_transport says which library ran)
  const char* forced = getenv("TPG_RCCL_LIBRARY");
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    if (forced && forced[0]) nm = forced;
    api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (api.handle) { g_rccl_path = nm; break; }
    const char* why = dlerror();
    g_rccl_why = why ? why : "not found";
    if (forced && forced[0]) break;
  }
  if (!api.handle) return;
#define RSYM(field, sym)                                                                          \
  do {                                                                                            \
    *(void**)(&api.field) = dlsym(api.handle, sym);                                               \
    if (!api.field) {                                                                             \
      g_rccl_why = std::string("librccl.so lacks ") + sym;                                        \
      dlclose(api.handle);                                                                        \
      api.handle = nullptr;                                                                       \
      return;                                                                                     \
    }                                                                                             \
  } while (0)
  RSYM(GetUniqueId, "ncclGetUniqueId");
  RSYM(CommInitRank, "ncclCommInitRank");
  RSYM(CommInitAll, "ncclCommInitAll");
  RSYM(CommDestroy, "ncclCommDestroy");
  RSYM(AllReduce, "ncclAllReduce");
  RSYM(ReduceScatter, "ncclReduceScatter");
  RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
  // ncclAllToAllv is an RCCL extension only the optional class exchange of the PCA Gram uses: a library without it
  // keeps every other multi-GPU path (tpg_comm_alltoall_usable then says no, on every rank alike -- same library)
  *(void**)(&api.AllToAllv) = dlsym(api.handle, "ncclAllToAllv");
}

// several host threads (one context each) may ask at once: the loader runs once, the others wait for it
static RcclApi* rccl() {
  std::call_once(g_rccl_once, rccl_load);
  return g_rccl.handle ? &g_rccl : nullptr;
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
  TPG_REQUIRE(api, TPG_EHIP, "RCCL (librccl.so) could not be loaded: %s", g_rccl_why.c_str());
  ncclUniqueId id;
  TPG_RCCL(api, api->GetUniqueId(&id));
  memcpy(id128, &id, 128);
  return TPG_OK;
}

// the status word of tpg_comm_agree lives in device memory reserved HERE: a rank that has just run out of memory must
// still be able to tell the others
static void comm_delete(tpg_comm* c) {
  if (!c) return;
  if (c->d_status) (void)hipFree(c->d_status);
  delete c;
}

static tpg_comm* comm_new(tpg_ctx* ctx, int nranks, int rank) {
  tpg_comm* c = new tpg_comm();
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  if (hipMalloc((void**)&c->d_status, sizeof(int32_t) * TPG_COMM_STATUS_INTS) != hipSuccess) {
    (void)hipGetLastError();
    delete c;
    tpg_set_error("hipMalloc of the communicator's status word failed");
    return nullptr;
  }
  return c;
}

extern "C" int tpg_comm_init_rank(tpg_ctx* ctx, int nranks, int rank, const uint8_t* id128, tpg_comm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, TPG_EINVAL, "bad rank %d of %d", rank, nranks);
  tpg_comm* c = comm_new(ctx, nranks, rank);
  if (!c) return TPG_EHIP;
  // a single rank exchanges nothing: no communicator, every collective is the identity (TPG_COMM_FORCE_RCCL=1 builds
  // a real one-rank RCCL communicator all the same: the rehearsal of the RCCL calls a one-GPU box allows)
  const bool force = getenv("TPG_COMM_FORCE_RCCL") && getenv("TPG_COMM_FORCE_RCCL")[0] == '1';
  if (nranks > 1 || force) {
    RcclApi* api = rccl();
    if (!api || (!id128 && nranks > 1)) {
      comm_delete(c);
      if (api) tpg_set_error("null unique id");
      else tpg_set_error("RCCL (librccl.so) could not be loaded: %s", g_rccl_why.c_str());
      return api ? TPG_EINVAL : TPG_EHIP;
    }
    ncclUniqueId id;
    if (id128) memcpy(&id, id128, 128);
    else if (api->GetUniqueId(&id) != ncclSuccess) { comm_delete(c); tpg_set_error("ncclGetUniqueId failed"); return TPG_EHIP; }
    ncclComm_t nc = nullptr;
    ncclResult_t r = api->CommInitRank(&nc, nranks, id, rank);  // the context's device is current (TpgEnter)
    if (r != ncclSuccess) {
      comm_delete(c);
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
  tpg_comm* c = comm_new(ctx, nranks, rank);
  if (!c) return TPG_EHIP;
  c->host_fn = allreduce;
  c->host_user = user;
  *out = c;
  return TPG_OK;
}

extern "C" void tpg_comm_destroy(tpg_comm* comm) {
  if (!comm) return;
  TpgEnter _enter(comm->ctx);
  if (comm->nccl) {
    RcclApi* api = rccl();
    if (api) {
      (void)hipStreamSynchronize(comm->ctx->stream);
      (void)api->CommDestroy((ncclComm_t)comm->nccl);
    }
  }
  comm_delete(comm);
}

// "none" (one rank: every exchange is the identity), "host callback" (rehearsal), or "rccl: <library as loaded>"
extern "C" const char* tpg_comm_transport(const tpg_comm* comm) {
  static thread_local std::string s;
  if (!comm || (!comm->nccl && !comm->host_fn)) return "none";
  if (comm->host_fn) return "host callback";
  s = "rccl: " + g_rccl_path;
  return s.c_str();
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

// Every rank passes the status of the rank-local steps it has just done (allocations, a shard with a zero scale ...) and
// all of them get the same answer: TPG_OK only if every rank said so, else the highest error code any rank reported.  Called
// before a data collective, so that a rank that failed does not leave the others waiting in it for ever (RCCL has no
// timeout).  A tiny all-reduce of one counter per error code; the identity on a single rank.
int tpg_comm_agree(tpg_comm* comm, int rc) {
  if (!comm || (comm->nranks == 1 && !comm->nccl)) return rc;
  const std::string own = rc != TPG_OK ? tpg_last_error() : "";
  int32_t h[TPG_COMM_STATUS_INTS] = {};
  h[rc >= 0 && rc < TPG_COMM_STATUS_INTS ? rc : TPG_EHIP] = 1;
  int xrc = TPG_OK;
  if (comm->host_fn) {
    if (comm->host_fn(comm->host_user, h, TPG_COMM_STATUS_INTS, 0) != 0) { tpg_set_error("the host all-reduce callback failed"); xrc = TPG_EHIP; }
  } else {
    RcclApi* api = rccl();
    hipStream_t s = comm->ctx->stream;
    hipError_t e = api && comm->nccl ? hipMemcpyAsync(comm->d_status, h, sizeof(h), hipMemcpyHostToDevice, s) : hipErrorNotInitialized;
    if (e == hipSuccess && api->AllReduce(comm->d_status, comm->d_status, TPG_COMM_STATUS_INTS, ncclInt32, ncclSum,
                                          (ncclComm_t)comm->nccl, s) != ncclSuccess)
      e = hipErrorUnknown;
    if (e == hipSuccess) e = hipMemcpyAsync(h, comm->d_status, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { tpg_set_error("status exchange between the ranks failed: %s", hipGetErrorString(e)); xrc = TPG_EHIP; }
  }
  if (xrc != TPG_OK) return xrc;
  if (rc != TPG_OK) { tpg_set_error("%s", own.c_str()); return rc; }
  int failed = 0, worst = TPG_OK;
  for (int c = 1; c < TPG_COMM_STATUS_INTS; c++)
    if (h[c] > 0) { failed += h[c]; worst = c; }
  if (!failed) return TPG_OK;
  tpg_set_error("%d of %d ranks failed before the exchange (error code %d there): every rank gives up", failed, comm->nranks, worst);
  return worst;
}

// All-to-all of 8-byte words between the ranks (device memory, stream-ordered): rank r sends scnt[d] words at soff[d] of
// d_send to rank d and receives rcnt[s] words from rank s at roff[s] of d_recv (what ncclAllToAllv does).  Used by the
// PCA Gram to give every rank whole weight classes (gramcls.hip: the packed genotype columns travel, 1 280 bytes per
// locus at n = 5 000).  On the rehearsal transport the exchange is emulated by an all-reduce of a buffer in which every
// rank fills only the slots it sends (tests: small panels).
int tpg_comm_alltoallv64(tpg_comm* comm, const void* d_send, const size_t* scnt, const size_t* soff, void* d_recv,
                         const size_t* rcnt, const size_t* roff) {
  TPG_REQUIRE(comm && scnt && soff && rcnt && roff, TPG_EINVAL, "bad all-to-all arguments");
  const int R = comm->nranks, me = comm->rank;
  hipStream_t s = comm->ctx->stream;
  if (R == 1 && !comm->nccl) {
    TPG_REQUIRE(scnt[0] == rcnt[0], TPG_EINVAL, "all-to-all counts do not match");
    if (scnt[0]) TPG_HIP(hipMemcpyAsync((uint64_t*)d_recv + roff[0], (const uint64_t*)d_send + soff[0], 8 * scnt[0], hipMemcpyDeviceToDevice, s));
    return TPG_OK;
  }
  if (comm->host_fn) {
    std::vector<int32_t> cm((size_t)R * R, 0);
    auto sizes = [&]() -> int {
      for (int d = 0; d < R; d++) {
        TPG_REQUIRE(scnt[d] < (1ull << 30), TPG_EUNSUPPORTED, "rehearsal all-to-all too large");
        cm[(size_t)me * R + d] = (int32_t)scnt[d];
      }
      return TPG_OK;
    };
    TPG_TRY(tpg_comm_agree(comm, sizes()));
    TPG_REQUIRE(comm->host_fn(comm->host_user, cm.data(), (int64_t)R * R, 0) == 0, TPG_EHIP, "the host all-reduce callback failed");
    std::vector<size_t> base((size_t)R + 1, 0);
    for (int d = 0; d < R; d++) {
      size_t in = 0;
      for (int r = 0; r < R; r++) in += (size_t)cm[(size_t)r * R + d];
      base[(size_t)d + 1] = base[(size_t)d] + in;
    }
    // everything that can fail on this rank alone between the two host all-reduces is checked first and AGREED on: a rank
    // that bailed out here would leave the others blocked in the payload all-reduce (a condition variable, no timeout)
    const size_t total = base[(size_t)R];
    std::vector<uint64_t> H;
    auto local = [&]() -> int {
      for (int r = 0; r < R; r++) TPG_REQUIRE((size_t)cm[(size_t)r * R + me] == rcnt[r], TPG_EINVAL, "all-to-all counts do not match");
      TPG_REQUIRE(total < (1ull << 28), TPG_EUNSUPPORTED, "rehearsal all-to-all too large");
      H.assign(total ? total : 1, 0);
      for (int d = 0; d < R; d++) {
        size_t o = base[(size_t)d];
        for (int r = 0; r < me; r++) o += (size_t)cm[(size_t)r * R + d];
        if (scnt[d]) TPG_HIP(hipMemcpyAsync(H.data() + o, (const uint64_t*)d_send + soff[d], 8 * scnt[d], hipMemcpyDeviceToHost, s));
      }
      TPG_HIP(hipStreamSynchronize(s));
      return TPG_OK;
    };
    TPG_TRY(tpg_comm_agree(comm, local()));
    TPG_REQUIRE(comm->host_fn(comm->host_user, H.data(), (int64_t)(2 * total), 0) == 0, TPG_EHIP, "the host all-reduce callback failed");
    size_t o = base[(size_t)me];
    for (int r = 0; r < R; r++) {
      if (rcnt[r]) TPG_HIP(hipMemcpyAsync((uint64_t*)d_recv + roff[r], H.data() + o, 8 * rcnt[r], hipMemcpyHostToDevice, s));
      o += rcnt[r];
    }
    TPG_HIP(hipStreamSynchronize(s));
    return TPG_OK;
  }
  RcclApi* api = rccl();
  TPG_REQUIRE(api && comm->nccl, TPG_EHIP, "communicator has no transport");
  TPG_REQUIRE(api->AllToAllv, TPG_EUNSUPPORTED, "this librccl.so has no ncclAllToAllv");
  TPG_RCCL(api, api->AllToAllv(d_send, scnt, soff, d_recv, rcnt, roff, ncclUint64, (ncclComm_t)comm->nccl, s));
  return TPG_OK;
}

// The all-to-all is the one collective of this library that a run on one GPU cannot exercise over RCCL.  Before its first
// real use every communicator sends a few known words round and checks what arrives; a transport error or a wrong word
// on ANY rank switches the class exchange off on ALL of them (tpg_comm_agree), and the PCA falls back to the Gram matrix
// of every rank's own loci, which needs all-reduces only.
bool tpg_comm_alltoall_usable(tpg_comm* comm) {
  if (!comm) return false;
  if (comm->a2a_state != 0) return comm->a2a_state > 0;
  if (comm->nccl && !comm->host_fn) {
    RcclApi* api = rccl();
    if (!api || !api->AllToAllv) { comm->a2a_state = -1; return false; }  // every rank loads the same library: alike everywhere
  }
  const int R = comm->nranks, me = comm->rank, W = 4;
  std::vector<uint64_t> hs((size_t)R * W), hr((size_t)R * W, 0);
  std::vector<size_t> cnt((size_t)R, (size_t)W), off((size_t)R);
  for (int d = 0; d < R; d++) {
    off[(size_t)d] = (size_t)d * W;
    for (int w = 0; w < W; w++) hs[(size_t)d * W + w] = 0x5450470000000000ull + (uint64_t)me * 65536 + (uint64_t)d * 256 + (uint64_t)w;
  }
  uint64_t *ds = nullptr, *dr = nullptr;
  int rc = TPG_OK;
  hipStream_t s = comm->ctx->stream;
  if (tpg_pmalloc((void**)&ds, 8 * hs.size()) != hipSuccess || tpg_pmalloc((void**)&dr, 8 * hr.size()) != hipSuccess) rc = TPG_EHIP;
  rc = tpg_comm_agree(comm, rc);
  if (rc == TPG_OK) {
    if (hipMemcpyAsync(ds, hs.data(), 8 * hs.size(), hipMemcpyHostToDevice, s) != hipSuccess) rc = TPG_EHIP;
    if (hipMemsetAsync(dr, 0, 8 * hr.size(), s) != hipSuccess) rc = TPG_EHIP;
    rc = tpg_comm_agree(comm, rc);
  }
  if (rc == TPG_OK) {
    rc = tpg_comm_alltoallv64(comm, ds, cnt.data(), off.data(), dr, cnt.data(), off.data());
    if (rc == TPG_OK && (hipMemcpyAsync(hr.data(), dr, 8 * hr.size(), hipMemcpyDeviceToHost, s) != hipSuccess ||
                         hipStreamSynchronize(s) != hipSuccess))
      rc = TPG_EHIP;
    if (rc == TPG_OK)
      for (int r = 0; r < R; r++)
        for (int w = 0; w < W; w++)
          if (hr[(size_t)r * W + w] != 0x5450470000000000ull + (uint64_t)r * 65536 + (uint64_t)me * 256 + (uint64_t)w) rc = TPG_EHIP;
    rc = tpg_comm_agree(comm, rc);
  }
  tpg_pfree(ds);
  tpg_pfree(dr);
  comm->a2a_state = rc == TPG_OK ? 1 : -1;
  if (rc != TPG_OK && getenv("TPG_DEBUG")) fprintf(stderr, "[tpg] all-to-all self-test failed on this communicator: no class exchange\n");
  return rc == TPG_OK;
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
//
#include "host/host_inproc.h"  // InprocGroup, InprocRank, inproc_allreduce

struct tpg_multi {
  int ndev = 0;
  std::vector<tpg_ctx*> ctx;
  std::vector<tpg_comm*> comm;
  InprocGroup inproc;
  std::vector<InprocRank> inproc_rank;
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
  bool host_transport = getenv("TPG_MULTI_HOST_TRANSPORT") && getenv("TPG_MULTI_HOST_TRANSPORT")[0] == '1';
  // a device listed twice: RCCL refuses it, the device threads exchange in process -- unless TPG_MULTI_FORCE_RCCL=1 says the
  // loaded library takes it (the mock of the tests does: ncclCommInitAll and every nccl* call site then run with ndev ranks)
  const bool force_rccl = getenv("TPG_MULTI_FORCE_RCCL") && getenv("TPG_MULTI_FORCE_RCCL")[0] == '1';
  for (int i = 0; i < ndev && !force_rccl; i++)
    for (int j = 0; j < i; j++)
      if (devs[(size_t)i] == devs[(size_t)j]) host_transport = true;
  if (ndev > 1 && host_transport) {
    mg->inproc.n = ndev;
    mg->inproc.slot.assign((size_t)ndev, nullptr);
    mg->inproc.slot_count.assign((size_t)ndev, 0);
    mg->inproc.slot_dtype.assign((size_t)ndev, 0);
    mg->inproc_rank.resize((size_t)ndev);
  } else if (ndev > 1) {
    RcclApi* api = rccl();
    if (!api) { tpg_multi_destroy(mg); tpg_set_error("RCCL (librccl.so) could not be loaded: %s", g_rccl_why.c_str()); return TPG_EHIP; }
    ncclResult_t r = api->CommInitAll(nc.data(), ndev, devs.data());
    if (r != ncclSuccess) {
      tpg_multi_destroy(mg);
      tpg_set_error("ncclCommInitAll(%d devices): %s", ndev, api->GetErrorString(r));
      return TPG_EHIP;
    }
  }
  for (int i = 0; i < ndev; i++) {
    TpgEnter _enter(mg->ctx[(size_t)i]);
    tpg_comm* c = comm_new(mg->ctx[(size_t)i], ndev, i);
    if (!c) {
      RcclApi* api = rccl();
      for (int j = i; j < ndev; j++)
        if (nc[(size_t)j] && api) (void)api->CommDestroy(nc[(size_t)j]);
      tpg_multi_destroy(mg);
      return TPG_EHIP;
    }
    c->nccl = nc[(size_t)i];
    if (mg->inproc.n) {
      mg->inproc_rank[(size_t)i] = InprocRank{&mg->inproc, i};
      c->host_fn = inproc_allreduce;
      c->host_user = &mg->inproc_rank[(size_t)i];
    }
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

// This device's share of the view (rowInd, colInd, code256) of a HOST FBM: the contiguous range [j0, j1) of colInd
// (tpg_shard_loci), only the FBM columns that range touches uploaded, packed.  An empty share (more devices than
// 128-locus groups) leaves f and v NULL: the device still takes part in the exchanges.
struct MultiShard {
  tpg_fbm* f = nullptr;
  tpg_view* v = nullptr;
  int64_t j0 = 0, j1 = 0;
};

static int multi_shard_view(tpg_multi* mg, int r, const uint8_t* fbm_bytes, int64_t nrow, const int32_t* rowInd1, int64_t n,
                            const int32_t* colInd1, int64_t m, const double* code256, MultiShard* me) {
  tpg_ctx* ctx = mg->ctx[(size_t)r];
  TPG_TRY(tpg_shard_loci(m, mg->ndev, r, &me->j0, &me->j1));
  const int64_t j0 = me->j0, j1 = me->j1;
  if (j1 <= j0) return TPG_OK;
  // a contiguous byte range of the column-major FBM when colInd is the identity, else the covering range of this
  // device's share of colInd with the indices rebased onto it
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
  // (one code table per call: the share goes up as 2 bits per genotype where table and bytes allow it)
  const double* table = code256;
  TPG_TRY(tpg_fbm_from_host_for_table(ctx, fbm_bytes + (size_t)c0 * (size_t)nrow, nrow, c1 - c0, code256, &me->f, &table));
  return tpg_view_create(ctx, me->f, rowInd1, n, colInd1 ? cols.data() : nullptr, j1 - j0, table, &me->v);
}

static void multi_shard_free(tpg_multi* mg, std::vector<MultiShard>& st) {
  for (int r = 0; r < mg->ndev; r++) {
    TpgEnter _enter(mg->ctx[(size_t)r]);
    tpg_view_free(st[(size_t)r].v);
    tpg_fbm_free(st[(size_t)r].f);
  }
}

static int multi_check_args(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1,
                            int64_t* n, const int32_t* colInd1, int64_t* m) {
  TPG_REQUIRE(mg && fbm_bytes, TPG_EINVAL, "null argument");
  TPG_REQUIRE(nrow > 0 && ncol > 0, TPG_EINVAL, "empty FBM");
  if (!rowInd1) *n = nrow;
  if (!colInd1) *m = ncol;
  TPG_REQUIRE(*n > 0 && *m > 0, TPG_EINVAL, "empty view");
  if (colInd1)
    for (int64_t j = 0; j < *m; j++)
      TPG_REQUIRE(colInd1[j] >= 1 && colInd1[j] <= ncol, TPG_EINVAL, "colInd[%lld] = %d out of [1,%lld]", (long long)j,
                  colInd1[j], (long long)ncol);
  return TPG_OK;
}

// rows [j0, j0 + ml) of the caller's m x ncols column-major host (or device) matrix <- a device's ml x ncols block
static int multi_rows_to_caller(tpg_ctx* ctx, double* dst, int64_t m, int64_t j0, const double* d_src, int64_t ml, int64_t ncols) {
  if (ml <= 0 || ncols <= 0) return TPG_OK;
  TPG_HIP(hipMemcpy2DAsync(dst + j0, sizeof(double) * (size_t)m, d_src, sizeof(double) * (size_t)ml, sizeof(double) * (size_t)ml,
                           (size_t)ncols, hipMemcpyDefault, ctx->stream));
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  return TPG_OK;
}

struct PoolBuf {  // device scratch of one device thread, back to its pool on scope exit
  void* p = nullptr;
  int alloc(size_t bytes) { TPG_HIP(tpg_pmalloc(&p, bytes ? bytes : 16)); return TPG_OK; }
  ~PoolBuf() { tpg_pfree(p); }
  template <typename T> T* as() { return (T*)p; }
};

// A device holds its share of the panel whole (one upload, one pack) unless the share's bytes exceed the streaming budget:
// then the call is the streamed form (stream.hip: every device sweeps its share in blocks, two block buffers, the same
// exchanges afterwards) -- a 5 000 x 80 000 000 panel (400 GB) goes through the same entry points as a 5-GB one.
// TPG_STREAM_BUDGET=<bytes> sets the budget (default: a quarter of the device's memory).
static bool multi_stream_budget(tpg_multi* mg, int64_t nrow, int64_t m, size_t* budget) {
  size_t b = 0;
  if (const char* e = getenv("TPG_STREAM_BUDGET")) {
    b = (size_t)strtoull(e, nullptr, 10);
  } else {
    TpgEnter _enter(mg->ctx[0]);
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return false; }
    b = tot / 4;
  }
  if (b == 0) return false;
  *budget = b;
  return (size_t)nrow * (size_t)ceil_div(m, mg->ndev) > b;
}
static void stream_job_init(tpg_stream_job* job, const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m) {
  memset(job, 0, sizeof(*job));
  job->struct_size = sizeof(*job);
  job->rowInd1 = rowInd1;
  job->n = n;
  job->colInd1 = colInd1;
  job->m = m;
}

// snp_ibs / snp_king / snp_allele_sharing / pairwise_grm of one host FBM on all the devices of `mg`
// (R/snp_ibs.R:42-104, R/snp_king.R:32-103, R/snp_allele_sharing.R:33-82, R/pairwise_grm.R:30-51 -- their block loops
// become: every device takes a contiguous share of colInd, uploads just those columns, packs, accumulates; one
// reduce-scatter; every device finishes its band of tiles and writes it straight into the caller's host matrices).
extern "C" int tpg_multi_pairwise(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol,
                                  const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m, int ibs_type,
                                  double* ibs, double* king, double* allele_sharing, double* grm) {
  TPG_TRY(multi_check_args(mg, fbm_bytes, nrow, ncol, rowInd1, &n, colInd1, &m));
  size_t budget = 0;
  if (multi_stream_budget(mg, nrow, m, &budget)) {
    tpg_stream_job job;
    stream_job_init(&job, rowInd1, n, colInd1, m);
    job.ibs_type = ibs_type;
    job.ibs = ibs;
    job.king = king;
    job.allele_sharing = allele_sharing;
    job.grm = grm;
    return tpg_multi_stream_host(mg, fbm_bytes, nrow, ncol, budget, &job);
  }
  std::vector<MultiShard> st((size_t)mg->ndev);
  std::vector<tpg_pairwise*> pw((size_t)mg->ndev, nullptr);
  // only the cross-products the requested matrices are made of (2 of 5 for allele sharing / GRM alone, 3 for IBS, 4 for KING)
  const int products = tpg_pw_products_for(ibs != nullptr, king != nullptr, allele_sharing || grm);
  TPG_REQUIRE(products, TPG_EINVAL, "no output requested");
  // phase 1, no exchange: upload, pack and accumulate this device's loci.  The phases are separate thread teams so
  // that a failure on one device (out of memory, a bad index) is known to all before anyone enters a collective --
  // a rank that skipped the reduce-scatter would leave the others waiting in it for ever.
  int rc = multi_run(mg, [&](int r) -> int {
    tpg_ctx* ctx = mg->ctx[(size_t)r];
    TpgEnter _enter(ctx);
    TPG_TRY(tpg_pairwise_create_sharded(ctx, mg->comm[(size_t)r], n, &pw[(size_t)r]));
    TPG_TRY(multi_shard_view(mg, r, fbm_bytes, nrow, rowInd1, n, colInd1, m, nullptr /* raw bytes */, &st[(size_t)r]));
    if (!st[(size_t)r].v) return TPG_OK;  // nothing of its own, still takes part below
    return tpg_pairwise_accumulate_products(ctx, pw[(size_t)r], st[(size_t)r].v, 0, -1, products);
  });
  // phase 2: one reduce-scatter, then every device finishes its band and writes it into the caller's matrices
  if (rc == TPG_OK)
    rc = multi_run(mg, [&](int r) -> int {
      tpg_ctx* ctx = mg->ctx[(size_t)r];
      tpg_comm* comm = mg->comm[(size_t)r];
      TpgEnter _enter(ctx);
      TPG_TRY(tpg_pairwise_reduce(ctx, comm, pw[(size_t)r]));
      return tpg_pairwise_epilogues_sharded(ctx, comm, pw[(size_t)r], ibs_type, m, ibs, king, allele_sharing, grm);
    });
  std::string err = rc == TPG_OK ? "" : tpg_last_error();
  for (int r = 0; r < mg->ndev; r++) {
    TpgEnter _enter(mg->ctx[(size_t)r]);
    tpg_pairwise_free(pw[(size_t)r]);
  }
  multi_shard_free(mg, st);
  if (rc != TPG_OK) tpg_set_error("%s", err.c_str());
  return rc;
}

// loci_alt_freq (grouped: R/loci_alt_freq.R:174-197 around grouped_alt_freq_dip_pseudo_cpp; ungrouped when groupIds0
// is NULL: R/loci_alt_freq.R:328-379 around alt_freq_dip_pseudo_cpp) of one host FBM on all devices.  The outputs are
// per locus, so the devices' shares are disjoint row ranges of `out` (m x 2G, or m x 2): no exchange at all.
extern "C" int tpg_multi_grouped_alt_freq(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol,
                                          const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m,
                                          const double* code256, const int32_t* groupIds0, int ngroups,
                                          const double* ploidy, int as_counts, double* out) {
  TPG_TRY(multi_check_args(mg, fbm_bytes, nrow, ncol, rowInd1, &n, colInd1, &m));
  TPG_REQUIRE(out, TPG_EINVAL, "null output");
  TPG_REQUIRE(!groupIds0 || ngroups > 0, TPG_EINVAL, "ngroups must be positive");
  const int64_t ncols = groupIds0 ? 2 * (int64_t)ngroups : 2;
  size_t budget = 0;
  if (multi_stream_budget(mg, nrow, m, &budget)) {
    tpg_stream_job job;
    stream_job_init(&job, rowInd1, n, colInd1, m);
    job.code256 = code256;
    job.ploidy = ploidy;
    job.groupIds0 = groupIds0;
    job.ngroups = ngroups;
    job.as_counts = as_counts;
    if (groupIds0) job.grouped_alt_freq = out;
    else job.alt_freq = out;
    return tpg_multi_stream_host(mg, fbm_bytes, nrow, ncol, budget, &job);
  }
  std::vector<MultiShard> st((size_t)mg->ndev);
  int rc = multi_run(mg, [&](int r) -> int {
    tpg_ctx* ctx = mg->ctx[(size_t)r];
    TpgEnter _enter(ctx);
    MultiShard& me = st[(size_t)r];
    TPG_TRY(multi_shard_view(mg, r, fbm_bytes, nrow, rowInd1, n, colInd1, m, code256, &me));
    if (!me.v) return TPG_OK;
    const int64_t ml = me.j1 - me.j0;
    PoolBuf d_out;
    TPG_TRY(d_out.alloc(sizeof(double) * (size_t)ml * (size_t)ncols));
    if (groupIds0) TPG_TRY(tpg_grouped_alt_freq_dip_pseudo(ctx, me.v, groupIds0, ngroups, ploidy, as_counts, d_out.as<double>()));
    else TPG_TRY(tpg_alt_freq_dip_pseudo(ctx, me.v, ploidy, as_counts, d_out.as<double>()));
    return multi_rows_to_caller(ctx, out, m, me.j0, d_out.as<double>(), ml, ncols);
  });
  std::string err = rc == TPG_OK ? "" : tpg_last_error();
  multi_shard_free(mg, st);
  if (rc != TPG_OK) tpg_set_error("%s", err.c_str());
  return rc;
}

// pairwise_pop_fst (R/pairwise_pop_fst.R:116-161: grouped_summaries_dip_pseudo_cpp + one of the three loop functions)
// of one host FBM on all devices.  By-locus outputs are disjoint row ranges of the m x P matrices; the totals are
// ratios of sums over loci, so every device returns the numerator / denominator sums of its loci
// (tpg_pairwise_pop_fst_sums) and they are added here on the host, in device order (2 P doubles per device: the same
// answer on every run; one process needs no collective for that).
extern "C" int tpg_multi_pop_fst(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1,
                                 int64_t n, const int32_t* colInd1, int64_t m, const double* code256,
                                 const int32_t* groupIds0, int ngroups, const double* ploidy, int method,
                                 const int32_t* pairs1, int P, int by_locus, int return_num_dem, double* fst_tot,
                                 double* out_a, double* out_b) {
  TPG_TRY(multi_check_args(mg, fbm_bytes, nrow, ncol, rowInd1, &n, colInd1, &m));
  TPG_REQUIRE(groupIds0 && pairs1 && P > 0 && ngroups > 0, TPG_EINVAL, "null or empty argument");
  if (return_num_dem) by_locus = 1;  // R/pairwise_pop_fst.R:103-106
  TPG_REQUIRE(!by_locus || out_a, TPG_EINVAL, "by_locus output requested but out_a is NULL");
  TPG_REQUIRE(!return_num_dem || out_b, TPG_EINVAL, "return_num_dem requested but out_b is NULL");
  TPG_REQUIRE(return_num_dem || fst_tot, TPG_EINVAL, "fst_tot is NULL");
  size_t budget = 0;
  if (multi_stream_budget(mg, nrow, m, &budget)) {
    tpg_stream_job job;
    stream_job_init(&job, rowInd1, n, colInd1, m);
    job.code256 = code256;
    job.ploidy = ploidy;
    job.groupIds0 = groupIds0;
    job.ngroups = ngroups;
    job.nfst = 1;
    job.fst_method[0] = method;
    job.pairs1 = pairs1;
    job.P = P;
    job.fst_return_num_dem = return_num_dem;
    job.fst_tot[0] = return_num_dem ? nullptr : fst_tot;
    job.fst_by_locus[0] = by_locus ? out_a : nullptr;
    job.fst_by_locus_den[0] = return_num_dem ? out_b : nullptr;
    return tpg_multi_stream_host(mg, fbm_bytes, nrow, ncol, budget, &job);
  }
  std::vector<MultiShard> st((size_t)mg->ndev);
  std::vector<std::vector<double>> sums((size_t)mg->ndev, std::vector<double>(2 * (size_t)P, 0.0));
  int rc = multi_run(mg, [&](int r) -> int {
    tpg_ctx* ctx = mg->ctx[(size_t)r];
    TpgEnter _enter(ctx);
    MultiShard& me = st[(size_t)r];
    TPG_TRY(multi_shard_view(mg, r, fbm_bytes, nrow, rowInd1, n, colInd1, m, code256, &me));
    if (!me.v) return TPG_OK;
    const int64_t ml = me.j1 - me.j0;
    double* sn = sums[(size_t)r].data();
    if (!return_num_dem) TPG_TRY(tpg_pairwise_pop_fst_sums(ctx, me.v, groupIds0, ngroups, ploidy, method, pairs1, P, sn, sn + P));
    if (!by_locus) return TPG_OK;
    PoolBuf da, db;
    TPG_TRY(da.alloc(sizeof(double) * (size_t)ml * (size_t)P));
    if (return_num_dem) TPG_TRY(db.alloc(sizeof(double) * (size_t)ml * (size_t)P));
    std::vector<double> tot((size_t)P);  // this shard's own ratios: not what the caller asked for
    TPG_TRY(tpg_pairwise_pop_fst(ctx, me.v, groupIds0, ngroups, ploidy, method, pairs1, P, 1, return_num_dem, tot.data(),
                                 da.as<double>(), db.as<double>()));
    TPG_TRY(multi_rows_to_caller(ctx, out_a, m, me.j0, da.as<double>(), ml, P));
    if (return_num_dem) TPG_TRY(multi_rows_to_caller(ctx, out_b, m, me.j0, db.as<double>(), ml, P));
    return TPG_OK;
  });
  if (rc == TPG_OK && fst_tot && !return_num_dem) {
    std::vector<double> tot(2 * (size_t)P, 0.0);
    for (int r = 0; r < mg->ndev; r++)
      for (size_t q = 0; q < 2 * (size_t)P; q++) tot[q] += sums[(size_t)r][q];
    std::vector<double> ratio((size_t)P);
    for (int q = 0; q < P; q++) ratio[(size_t)q] = tot[(size_t)q] / tot[(size_t)(P + q)];
    if (tpg_is_device_ptr(fst_tot)) {
      tpg_ctx* ctx = mg->ctx[0];
      TpgEnter _enter(ctx);
      hipError_t e = hipMemcpyAsync(fst_tot, ratio.data(), sizeof(double) * (size_t)P, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) { tpg_set_error("fst_tot: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
    } else {
      memcpy(fst_tot, ratio.data(), sizeof(double) * (size_t)P);
    }
  }
  std::string err = rc == TPG_OK ? "" : tpg_last_error();
  multi_shard_free(mg, st);
  if (rc != TPG_OK) tpg_set_error("%s", err.c_str());
  return rc;
}

// gt_pca_partialSVD (R/gt_pca_partialSVD.R:67-108 around bigstatsr::big_SVD) of one host FBM on all devices: every
// device packs its share of colInd and runs tpg_pca_partial_svd_sharded (Gram matrix of its loci, one all-reduce over
// RCCL, replicated eigen step, the loadings of its loci); center / scale / loadings land in the devices' row ranges of
// the caller's arrays, d / u come from device 0 (identical everywhere).  A panel too short to give every device k
// loci runs on device 0 alone.
extern "C" int tpg_multi_pca_partial_svd(tpg_multi* mg, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol,
                                         const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m,
                                         const double* code256, int k, double* d, double* u, double* vload, double* center,
                                         double* scale, double* square_frobenius) {
  TPG_TRY(multi_check_args(mg, fbm_bytes, nrow, ncol, rowInd1, &n, colInd1, &m));
  TPG_REQUIRE(d && u && vload && center && scale, TPG_EINVAL, "null output");
  TPG_REQUIRE(k >= 1 && k <= n && k <= m, TPG_EINVAL, "k = %d out of range", k);
  size_t budget = 0;
  if (multi_stream_budget(mg, nrow, m, &budget)) {
    tpg_stream_job job;
    stream_job_init(&job, rowInd1, n, colInd1, m);
    job.code256_pca = code256;
    job.k = k;
    job.d = d;
    job.u = u;
    job.v = vload;
    job.center = center;
    job.scale = scale;
    job.square_frobenius = square_frobenius;
    return tpg_multi_stream_host(mg, fbm_bytes, nrow, ncol, budget, &job);
  }
  bool spread = mg->ndev > 1;
  for (int r = 0; r < mg->ndev && spread; r++) {
    int64_t j0, j1;
    TPG_TRY(tpg_shard_loci(m, mg->ndev, r, &j0, &j1));
    if (j1 - j0 < k) spread = false;
  }
  if (!spread) {  // one device, no exchange
    tpg_ctx* ctx = mg->ctx[0];
    TpgEnter _enter(ctx);
    tpg_fbm* f = nullptr;
    tpg_view* v = nullptr;
    const double* table = code256;
    int rc = tpg_fbm_from_host_for_table(ctx, fbm_bytes, nrow, ncol, code256, &f, &table);
    if (rc == TPG_OK) rc = tpg_view_create(ctx, f, rowInd1, n, colInd1, m, table, &v);
    if (rc == TPG_OK) rc = tpg_pca_partial_svd(ctx, v, k, d, u, vload, center, scale, square_frobenius);
    std::string err = rc == TPG_OK ? "" : tpg_last_error();
    tpg_view_free(v);
    tpg_fbm_free(f);
    if (rc != TPG_OK) tpg_set_error("%s", err.c_str());
    return rc;
  }
  std::vector<MultiShard> st((size_t)mg->ndev);
  // phase 1 (no exchange): upload + pack; a failure here is known to all before anyone enters the Gram all-reduce
  int rc = multi_run(mg, [&](int r) -> int {
    TpgEnter _enter(mg->ctx[(size_t)r]);
    return multi_shard_view(mg, r, fbm_bytes, nrow, rowInd1, n, colInd1, m, code256, &st[(size_t)r]);
  });
  std::vector<double> fro((size_t)mg->ndev, 0.0);
  if (rc == TPG_OK)
    rc = multi_run(mg, [&](int r) -> int {
      tpg_ctx* ctx = mg->ctx[(size_t)r];
      TpgEnter _enter(ctx);
      MultiShard& me = st[(size_t)r];
      const int64_t ml = me.j1 - me.j0;
      PoolBuf dv, dc, ds, du, dd;
      // rank-local steps that can fail sit inside tpg_pca_partial_svd_sharded BEFORE its first exchange, where the ranks
      // agree on a status (tpg_comm_agree): nobody is left waiting in a collective
      int lrc = dv.alloc(sizeof(double) * (size_t)ml * (size_t)k);
      if (lrc == TPG_OK) lrc = dc.alloc(sizeof(double) * (size_t)ml);
      if (lrc == TPG_OK) lrc = ds.alloc(sizeof(double) * (size_t)ml);
      if (lrc == TPG_OK) lrc = du.alloc(sizeof(double) * (size_t)n * (size_t)k);
      if (lrc == TPG_OK) lrc = dd.alloc(sizeof(double) * (size_t)k);
      lrc = tpg_comm_agree(mg->comm[(size_t)r], lrc);
      TPG_TRY(lrc);
      TPG_TRY(tpg_pca_partial_svd_sharded(ctx, mg->comm[(size_t)r], me.v, k, dd.as<double>(), du.as<double>(), dv.as<double>(),
                                          dc.as<double>(), ds.as<double>(), square_frobenius ? &fro[(size_t)r] : nullptr));
      TPG_TRY(multi_rows_to_caller(ctx, vload, m, me.j0, dv.as<double>(), ml, k));
      TPG_TRY(multi_rows_to_caller(ctx, center, m, me.j0, dc.as<double>(), ml, 1));
      TPG_TRY(multi_rows_to_caller(ctx, scale, m, me.j0, ds.as<double>(), ml, 1));
      if (r == 0) {
        TPG_HIP(hipMemcpyAsync(u, du.as<double>(), sizeof(double) * (size_t)n * (size_t)k, hipMemcpyDefault, ctx->stream));
        TPG_HIP(hipMemcpyAsync(d, dd.as<double>(), sizeof(double) * (size_t)k, hipMemcpyDefault, ctx->stream));
        TPG_HIP(hipStreamSynchronize(ctx->stream));
      }
      return TPG_OK;
    });
  if (rc == TPG_OK && square_frobenius) *square_frobenius = fro[0];
  std::string err = rc == TPG_OK ? "" : tpg_last_error();
  multi_shard_free(mg, st);
  if (rc != TPG_OK) tpg_set_error("%s", err.c_str());
  return rc;
}
