// mock_rccl.cpp -- TEST INFRASTRUCTURE, not a transport.  A stand-in for librccl.so that lets the library's nccl* call sites
// (tidypopgen_amd/csrc/comm.hip) run with N > 1 ranks on a box that has ONE GPU: RCCL itself refuses two ranks on one
// device.  It exports the symbols comm.hip loads (ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll, ncclCommDestroy,
// ncclAllReduce, ncclReduceScatter, ncclAllToAllv, ncclGetErrorString) with the semantics rccl.h documents:
//   * counts and displacements are in ELEMENTS of the datatype, never bytes (rccl.h:797-817);
//   * ncclReduceScatter: sendbuff holds nranks * recvcount elements, rank r receives the sum of chunk r; in place when
//     recvbuff == sendbuff + rank * recvcount;
//   * ncclAllToAllv: rank i sends sendcounts[j] elements at sdispls[j] to rank j and receives recvcounts[j] elements from
//     rank j at rdispls[j]; a send count that differs from the peer's receive count is an error here (on hardware: a hang
//     or silent corruption).
// The ranks may be threads of one process (ncclCommInitAll with a device listed several times: tpg_multi) or processes
// (ncclCommInitRank with a broadcast id: bench.py's ranks); either way they meet in POSIX shared memory: one control file per
// communicator (arrival counter, barrier, per-rank operation records) and one payload file per rank.  By default every
// operation is synchronous -- wait for the caller's stream, copy the send buffer to the rank's payload file, barrier, check
// that all ranks posted the SAME operation with the same count and type, combine on the host in rank order, copy the result
// to the receive buffer, barrier -- which proves the call sites' arguments (units, offsets, in-place use, buffer extents, the
// order of collectives on every rank).  MOCK_RCCL_ASYNC=1 makes it stream-ordered and adversarial instead (see "One
// collective = a plan" below): the call only enqueues, the receive buffer holds poison until a delayed combine has run, so a
// consumer that is not ordered behind the collective on the stream fails its parity test.  What neither mode proves: RCCL's
// own kernels and protocols, peer mappings, xGMI.  Every wait has a timeout: a rank that never arrives is an error
// (ncclSystemError), not a hang; the wait kernel of the stream-ordered mode is bounded by the wall clock.  Every buffer is checked to be device memory and the accessed range to lie inside its
// allocation (hipMemGetAddressRange), which is how a count passed in bytes shows up even when the sums happen to agree.
// Selected by TPG_RCCL_LIBRARY=<path to this .so> (comm.hip: rccl_load).
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

extern "C" {
typedef struct mockComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7, ncclNumResults = 8 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

namespace {

constexpr int MAX_RANKS = 64;
constexpr double TIMEOUT_S = 120.0;

enum Op : int32_t { OP_NONE = 0, OP_ALLREDUCE = 1, OP_REDUCESCATTER = 2, OP_ALLTOALLV = 3 };

struct RankRecord {  // what a rank posted for the operation in flight
  int32_t op, dtype;
  uint64_t count;               // all-reduce: elements; reduce-scatter: elements per chunk
  uint64_t bytes;               // payload bytes in the rank's file
  uint64_t a2a_off[MAX_RANKS];  // all-to-all: byte offset in the payload file of the piece for rank d
  uint64_t a2a_cnt[MAX_RANKS];  // ... and its element count
};

struct Control {
  std::atomic<int32_t> arrived;   // ranks that have attached
  std::atomic<int32_t> detached;  // ranks that have left
  std::atomic<int32_t> bar_count;
  std::atomic<int32_t> bar_gen;
  std::atomic<int32_t> failed;    // some rank saw an error inside a collective: everybody returns it
  std::atomic<int32_t> bar_failed[2];  // `failed` as the rank that completed barrier generation g saw it, at [g & 1]: every rank
                                       // leaves a barrier with the SAME verdict (a rank that read `failed` later could see the
                                       // flag a faster peer raised after the barrier and skip the next one: its peer waits for ever)
  int32_t nranks;
  uint64_t ops;                   // collectives completed (statistics for the tests: mock_rccl_stats)
  RankRecord rec[MAX_RANKS];
};

struct Mapping {
  int fd = -1;
  void* p = nullptr;
  size_t bytes = 0;
  void drop() {
    if (p) munmap(p, bytes);
    if (fd >= 0) close(fd);
    fd = -1; p = nullptr; bytes = 0;
  }
};

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void nap() {
  timespec ts{0, 50 * 1000};
  nanosleep(&ts, nullptr);
}

}  // namespace

struct mockComm {
  int nranks = 0, rank = 0;
  std::string name;
  Mapping ctl;
  Mapping mine;                 // this rank's payload file (grows)
  std::vector<Mapping> peer;    // read-only views of the others', remapped when they grow
  uint64_t n_allreduce = 0, n_reducescatter = 0, n_alltoallv = 0, bytes_moved = 0;
  void* async = nullptr;  // AsyncState of the stream-ordered mode (made by the first collective)
  Control* C() { return (Control*)ctl.p; }
};

static thread_local std::string t_err = "no error";
static std::atomic<uint64_t> g_total_ops{0};
static std::atomic<uint64_t> g_counts[4];

static ncclResult_t fail(ncclResult_t r, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
  char buf[512];
  snprintf(buf, sizeof buf, fmt, a, b, c);
  t_err = std::string("mock rccl: ") + buf;
  if (getenv("MOCK_RCCL_DEBUG")) fprintf(stderr, "[mock_rccl] %s\n", t_err.c_str());
  return r;
}

static size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
  }
  return 0;
}

static std::string file_of(const std::string& name, int rank) {
  return "/dev/shm/" + name + (rank < 0 ? std::string(".ctl") : ".r" + std::to_string(rank));
}

static bool map_file(Mapping& m, const std::string& path, size_t bytes, bool create, bool writable) {
  m.drop();
  m.fd = open(path.c_str(), (writable ? O_RDWR : O_RDONLY) | (create ? O_CREAT : 0), 0600);
  if (m.fd < 0) return false;
  if (create && writable) {
    struct stat st;
    if (fstat(m.fd, &st) != 0) { m.drop(); return false; }
    if ((size_t)st.st_size < bytes && ftruncate(m.fd, (off_t)bytes) != 0) { m.drop(); return false; }
  }
  m.p = mmap(nullptr, bytes, PROT_READ | (writable ? PROT_WRITE : 0), MAP_SHARED, m.fd, 0);
  if (m.p == MAP_FAILED) { m.p = nullptr; m.drop(); return false; }
  m.bytes = bytes;
  return true;
}

// sense-reversing barrier over the control block; false = timeout or a rank reported a failure
static bool barrier(mockComm* c) {
  Control* C = c->C();
  const int gen = C->bar_gen.load(std::memory_order_acquire);
  if (C->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
    const int32_t verdict = C->failed.load();
    C->bar_failed[gen & 1].store(verdict, std::memory_order_relaxed);
    C->bar_count.store(0, std::memory_order_relaxed);
    C->bar_gen.store(gen + 1, std::memory_order_release);
    return verdict == 0;
  }
  const double t0 = now_s();
  while (C->bar_gen.load(std::memory_order_acquire) == gen) {
    if (now_s() - t0 > TIMEOUT_S) { C->failed.store(1); return false; }
    nap();
  }
  return C->bar_failed[gen & 1].load(std::memory_order_relaxed) == 0;
}

// device buffer [p, p + bytes) must be device memory and lie inside ONE allocation
static bool device_range_ok(const void* p, size_t bytes, const char* what) {
  if (bytes == 0) return true;
  if (!p) { fail(ncclInvalidArgument, "%s is NULL", what); return false; }
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeDevice) {
    (void)hipGetLastError();
    fail(ncclInvalidArgument, "%s is not device memory", what);
    return false;
  }
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
    (void)hipGetLastError();
    fail(ncclInvalidArgument, "%s: no allocation holds it", what);
    return false;
  }
  const size_t off = (size_t)((const char*)p - (const char*)base);
  if (off + bytes > size) {
    fail(ncclInvalidArgument, "%s: %lld bytes accessed, the allocation ends after %lld", what, (long long)bytes, (long long)(size - off));
    return false;
  }
  return true;
}

static bool grow_mine(mockComm* c, size_t bytes) {
  if (bytes < 4096) bytes = 4096;
  if (c->mine.p && c->mine.bytes >= bytes) return true;
  size_t want = c->mine.bytes ? c->mine.bytes : 4096;
  while (want < bytes) want *= 2;
  return map_file(c->mine, file_of(c->name, c->rank), want, true, true);
}

static const void* peer_payload(mockComm* c, int r, size_t bytes) {
  if (r == c->rank) return c->mine.p;
  Mapping& m = c->peer[(size_t)r];
  if (m.p && m.bytes >= bytes) return m.p;
  struct stat st;
  const std::string path = file_of(c->name, r);
  if (stat(path.c_str(), &st) != 0 || (size_t)st.st_size < bytes) return nullptr;
  if (!map_file(m, path, (size_t)st.st_size, false, false)) return nullptr;
  return m.p;
}

static mockComm* attach(const std::string& name, int nranks, int rank) {
  mockComm* c = new mockComm();
  c->nranks = nranks; c->rank = rank; c->name = name;
  c->peer.resize((size_t)nranks);
  if (!map_file(c->ctl, file_of(name, -1), sizeof(Control), true, true)) { delete c; return nullptr; }
  Control* C = c->C();  // a fresh file is all zeros: every field starts at 0
  C->nranks = nranks;
  if (!grow_mine(c, 4096)) { c->ctl.drop(); delete c; return nullptr; }
  C->arrived.fetch_add(1);
  return c;
}

template <typename T>
static void sum_into(T* acc, const T* src, size_t n) {
  for (size_t i = 0; i < n; i++) acc[i] += src[i];
}

static bool sum_typed(void* acc, const void* src, size_t n, ncclDataType_t t) {
  switch (t) {
    case ncclInt32: sum_into((int32_t*)acc, (const int32_t*)src, n); return true;
    case ncclUint32: sum_into((uint32_t*)acc, (const uint32_t*)src, n); return true;
    case ncclInt64: sum_into((int64_t*)acc, (const int64_t*)src, n); return true;
    case ncclUint64: sum_into((uint64_t*)acc, (const uint64_t*)src, n); return true;
    case ncclFloat32: sum_into((float*)acc, (const float*)src, n); return true;
    case ncclFloat64: sum_into((double*)acc, (const double*)src, n); return true;
    default: return false;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// One collective = a plan: which device pieces go into this rank's payload file, how the result is made on the host from
// the ranks' payloads, which device pieces receive it.  Two ways to run a plan:
//
//   synchronous (default): wait for the caller's stream, copy out, rendezvous, combine, copy in, return.  Arguments,
//     units, offsets and "every rank in the same collective" are checked and reported by the return code
//     (tests/test_gpu_multirank.py::test_mock_rccl_semantics_and_misuse_detection).
//
//   stream-ordered and adversarial (MOCK_RCCL_ASYNC=1): like RCCL, the call only ENQUEUES and returns.  On the caller's stream,
//     in order: the send pieces are copied to pinned memory; the receive pieces are filled with a poison pattern (0xA5);
//     a one-thread kernel waits -- bounded: MOCK_RCCL_KERNEL_TIMEOUT_S, default 30 -- for a word in pinned memory; the result
//     is copied from pinned memory into the receive pieces.  A helper thread per communicator does the rendezvous: it waits
//     for the event behind the staging copy, posts the payload, meets the other ranks, sleeps MOCK_RCCL_DELAY_MS (default 3),
//     combines on the host and only then raises the word.  So between the call and the stream reaching the end of the
//     collective the receive buffer holds poison: a consumer that is not ordered behind the collective ON THE STREAM (a read
//     from another stream or from the host without the event / the synchronisation it needs, a pool block handed to
//     somebody else while the collective still owns it) computes on poison and the parity tests fail.  Argument checks still
//     happen at the call; what only the rendezvous can see (a rank in another collective, mismatched all-to-all counts) makes
//     the communicator sticky-failed: the word is raised so that no kernel is left waiting, every later call returns the
//     error, and mock_rccl_async_errors() counts it for the tests.
//   What neither mode is: RCCL's kernels, its protocols, peer mappings, xGMI.
struct Piece {
  void* dev;
  size_t bytes;
  size_t off;  // offset in the staged payload / in the host result
};

struct Plan {
  Op op = OP_NONE;
  ncclDataType_t dtype = ncclInt8;
  uint64_t count = 0;
  size_t payload = 0, result = 0;
  std::vector<Piece> send, recv;
  uint64_t a2a_off[MAX_RANKS] = {}, a2a_cnt[MAX_RANKS] = {};
  std::vector<size_t> a2a_recvcounts;  // all-to-all: what this rank expects from each peer
  size_t es = 0;
};

// the host side of a plan, identical in both modes: post, barrier, same-operation check, combine into `result`, barrier
static ncclResult_t rendezvous(mockComm* c, const Plan& pl, const uint8_t* staged, uint8_t* result, int delay_ms) {
  Control* C = c->C();
  ncclResult_t rc = ncclSuccess;
  if (!grow_mine(c, pl.payload)) rc = fail(ncclSystemError, "cannot grow the payload file of rank %s%lld", "", c->rank);
  if (rc == ncclSuccess && pl.payload) memcpy(c->mine.p, staged, pl.payload);
  RankRecord& me = C->rec[c->rank];
  me.op = pl.op; me.dtype = (int32_t)pl.dtype; me.count = pl.count; me.bytes = pl.payload;
  memcpy(me.a2a_off, pl.a2a_off, sizeof me.a2a_off);
  memcpy(me.a2a_cnt, pl.a2a_cnt, sizeof me.a2a_cnt);
  if (rc != ncclSuccess) C->failed.store(1);
  if (!barrier(c)) return rc != ncclSuccess ? rc : fail(ncclSystemError, "a rank failed or did not arrive (operation %s%lld)", "", pl.op);
  // every rank must have posted the same operation (a rank in another collective is the classic multi-GPU hang)
  for (int r = 0; r < c->nranks && rc == ncclSuccess; r++) {
    const RankRecord& o = C->rec[r];
    if (o.op != pl.op || o.dtype != (int32_t)pl.dtype || (pl.op != OP_ALLTOALLV && o.count != pl.count))
      rc = fail(ncclInvalidUsage, "rank %s%lld posted another operation / count / type than this rank", "", r);
  }
  if (rc == ncclSuccess && delay_ms > 0) {
    timespec ts{delay_ms / 1000, (long)(delay_ms % 1000) * 1000000L};
    nanosleep(&ts, nullptr);
  }
  if (rc == ncclSuccess) {
    const int R = c->nranks;
    if (pl.op == OP_ALLREDUCE || pl.op == OP_REDUCESCATTER) {
      const size_t chunk = pl.result, skip = pl.op == OP_REDUCESCATTER ? (size_t)c->rank * chunk : 0;
      memset(result, 0, chunk ? chunk : 1);
      for (int r = 0; r < R && rc == ncclSuccess; r++) {  // rank order: the same bits on every rank
        const uint8_t* p = (const uint8_t*)peer_payload(c, r, pl.payload);
        if (!p) rc = fail(ncclSystemError, "cannot map the payload of rank %s%lld", "", r);
        else if (!sum_typed(result, p + skip, chunk / pl.es, pl.dtype)) rc = fail(ncclInvalidArgument, "datatype cannot be summed");
      }
      if (rc == ncclSuccess) { (pl.op == OP_ALLREDUCE ? c->n_allreduce : c->n_reducescatter)++; c->bytes_moved += pl.payload; }
    } else {
      size_t o_res = 0;
      for (int s = 0; s < R && rc == ncclSuccess; s++) {
        const RankRecord& o = C->rec[s];
        if (o.a2a_cnt[c->rank] != pl.a2a_recvcounts[(size_t)s]) {
          rc = fail(ncclInvalidUsage, "ncclAllToAllv: rank %s%lld sends %lld elements here, another count is expected", "", s,
                    (long long)o.a2a_cnt[c->rank]);
          break;
        }
        const size_t bts = pl.a2a_recvcounts[(size_t)s] * pl.es;
        if (bts) {
          const uint8_t* p = (const uint8_t*)peer_payload(c, s, o.bytes);
          if (!p) { rc = fail(ncclSystemError, "cannot map the payload of rank %s%lld", "", s); break; }
          memcpy(result + o_res, p + o.a2a_off[c->rank], bts);
          c->bytes_moved += bts;
        }
        o_res += bts;
      }
      if (rc == ncclSuccess) c->n_alltoallv++;
    }
  }
  if (rc != ncclSuccess) C->failed.store(1);
  if (!barrier(c)) return rc != ncclSuccess ? rc : fail(ncclSystemError, "a rank failed inside operation %s%lld", "", pl.op);
  if (c->rank == 0) C->ops++;
  g_total_ops++;
  g_counts[pl.op]++;
  return rc;
}

static ncclResult_t run_sync(mockComm* c, hipStream_t stream, const Plan& pl) {
  if (hipStreamSynchronize(stream) != hipSuccess) return fail(ncclUnhandledCudaError, "hipStreamSynchronize failed before the collective");
  std::vector<uint8_t> staged(pl.payload ? pl.payload : 1), result(pl.result ? pl.result : 1);
  for (const Piece& p : pl.send)
    if (p.bytes && hipMemcpy(staged.data() + p.off, p.dev, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) {
      c->C()->failed.store(1);
      (void)barrier(c);
      return fail(ncclUnhandledCudaError, "device -> host copy failed");
    }
  ncclResult_t rc = rendezvous(c, pl, staged.data(), result.data(), 0);
  if (rc != ncclSuccess) return rc;
  for (const Piece& p : pl.recv)
    if (p.bytes && hipMemcpy(p.dev, result.data() + p.off, p.bytes, hipMemcpyHostToDevice) != hipSuccess)
      return fail(ncclUnhandledCudaError, "host -> device copy failed");
  return ncclSuccess;
}

// ---- the stream-ordered mode ------------------------------------------------------------------------------------------
static bool async_mode() {
  static const bool on = getenv("MOCK_RCCL_ASYNC") && atoi(getenv("MOCK_RCCL_ASYNC")) != 0;
  return on;
}
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static std::atomic<uint64_t> g_async_errors{0}, g_async_ops{0}, g_kernel_timeouts{0};

__global__ void mock_rccl_wait_kernel(const uint32_t* flag, uint32_t seq, unsigned long long timeout_ticks, uint32_t* timed_out) {
  // one thread; the 100 MHz wall clock bounds the wait, so the grid drains whatever the host does
  const unsigned long long t0 = wall_clock64();
  while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
    if (wall_clock64() - t0 > timeout_ticks) {
      __hip_atomic_store(timed_out, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    __builtin_amdgcn_s_sleep(64);
  }
}

struct AsyncTask {
  Plan plan;
  uint32_t seq = 0;
  hipEvent_t staged = nullptr, done = nullptr;
  uint8_t *h_send = nullptr, *h_recv = nullptr;  // pinned
  size_t cap_send = 0, cap_recv = 0;
};

struct AsyncState {
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<AsyncTask*> queue, retired;
  bool stop = false;
  uint32_t seq = 0;
  uint32_t* flags = nullptr;  // pinned, coherent: [0] = sequence number of the last completed collective, [1] = a kernel timed out
  ncclResult_t sticky = ncclSuccess;
  std::string sticky_msg;
  int device = 0;
};

static void async_worker(mockComm* c, AsyncState* st) {
  (void)hipSetDevice(st->device);
  const int delay = env_int("MOCK_RCCL_DELAY_MS", 3);
  for (;;) {
    AsyncTask* t = nullptr;
    {
      std::unique_lock<std::mutex> lk(st->mu);
      st->cv.wait(lk, [&] { return st->stop || !st->queue.empty(); });
      if (st->queue.empty()) return;
      t = st->queue.front();
      st->queue.pop_front();
    }
    ncclResult_t rc = ncclSuccess;
    if (hipEventSynchronize(t->staged) != hipSuccess) rc = fail(ncclUnhandledCudaError, "the staging copy of a collective failed");
    if (rc == ncclSuccess) rc = rendezvous(c, t->plan, t->h_send, t->h_recv, delay);
    else { c->C()->failed.store(1); (void)barrier(c); }
    if (rc != ncclSuccess) {
      g_async_errors++;
      fprintf(stderr, "[mock_rccl] rank %d: collective %u failed behind the call: %s\n", c->rank, t->seq, t_err.c_str());
      std::lock_guard<std::mutex> lk(st->mu);
      if (st->sticky == ncclSuccess) { st->sticky = rc; st->sticky_msg = t_err; }
      // (the poison stays in the pinned result: whoever consumes it sees it)
      memset(t->h_recv, 0xA5, t->plan.result ? t->plan.result : 1);
    }
    g_async_ops++;
    __atomic_store_n(&st->flags[0], t->seq, __ATOMIC_RELEASE);  // the wait kernel of this collective may go on
    {
      std::lock_guard<std::mutex> lk(st->mu);
      st->retired.push_back(t);
    }
  }
}

static AsyncState* async_of(mockComm* c);

static ncclResult_t run_async(mockComm* c, hipStream_t stream, const Plan& pl) {
  AsyncState* st = async_of(c);
  if (!st) return fail(ncclSystemError, "cannot start the helper thread / pinned memory of the stream-ordered mode");
  {
    std::lock_guard<std::mutex> lk(st->mu);
    if (st->sticky != ncclSuccess) { t_err = st->sticky_msg; return st->sticky; }
  }
  // a task whose copies have all run is recycled (its pinned buffers with it): hipEventQuery never blocks, and nothing here
  // frees pinned memory while work is in flight -- a hipHostFree would synchronise the device and hide what this mode is for
  AsyncTask* t = nullptr;
  {
    std::lock_guard<std::mutex> lk(st->mu);
    for (auto it = st->retired.begin(); it != st->retired.end(); ++it)
      if (hipEventQuery((*it)->done) == hipSuccess) { t = *it; st->retired.erase(it); break; }
  }
  (void)hipGetLastError();
  if (!t) {
    t = new AsyncTask();
    if (hipEventCreateWithFlags(&t->staged, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&t->done, hipEventDisableTiming) != hipSuccess)
      return fail(ncclUnhandledCudaError, "cannot create events");
  }
  auto need = [](uint8_t*& p, size_t& cap, size_t bytes) -> bool {
    if (bytes <= cap) return true;
    size_t want = cap ? cap : 4096;
    while (want < bytes) want *= 2;
    uint8_t* q = nullptr;
    if (hipHostMalloc((void**)&q, want, hipHostMallocDefault) != hipSuccess) return false;
    // (the smaller buffer is leaked on purpose: freeing pinned memory synchronises the device)
    p = q;
    cap = want;
    return true;
  };
  if (!need(t->h_send, t->cap_send, pl.payload ? pl.payload : 1) || !need(t->h_recv, t->cap_recv, pl.result ? pl.result : 1))
    return fail(ncclSystemError, "cannot pin staging memory");
  t->plan = pl;
  {
    std::lock_guard<std::mutex> lk(st->mu);
    t->seq = ++st->seq;
  }
  const unsigned long long ticks = 100000000ull * (unsigned long long)env_int("MOCK_RCCL_KERNEL_TIMEOUT_S", 30);
  hipError_t e = hipSuccess;
  for (const Piece& p : pl.send)
    if (p.bytes && e == hipSuccess) e = hipMemcpyAsync(t->h_send + p.off, p.dev, p.bytes, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipEventRecord(t->staged, stream);
  for (const Piece& p : pl.recv)
    if (p.bytes && e == hipSuccess) e = hipMemsetAsync(p.dev, 0xA5, p.bytes, stream);  // poison until the collective completes
  if (e == hipSuccess) {
    hipLaunchKernelGGL(mock_rccl_wait_kernel, dim3(1), dim3(1), 0, stream, (const uint32_t*)st->flags, t->seq, ticks, st->flags + 1);
    e = hipGetLastError();
  }
  for (const Piece& p : pl.recv)
    if (p.bytes && e == hipSuccess) e = hipMemcpyAsync(p.dev, t->h_recv + p.off, p.bytes, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipEventRecord(t->done, stream);
  if (e != hipSuccess) {
    // nothing may be left waiting: run the rendezvous as a failure and raise the word
    c->C()->failed.store(1);
    std::lock_guard<std::mutex> lk(st->mu);
    st->sticky = ncclUnhandledCudaError;
    st->sticky_msg = std::string("mock rccl: enqueueing a collective failed: ") + hipGetErrorString(e);
    __atomic_store_n(&st->flags[0], t->seq, __ATOMIC_RELEASE);
    t_err = st->sticky_msg;
    return ncclUnhandledCudaError;
  }
  {
    std::lock_guard<std::mutex> lk(st->mu);
    st->queue.push_back(t);
  }
  st->cv.notify_all();
  return ncclSuccess;
}

static AsyncState* async_of(mockComm* c) {
  if (c->async) return (AsyncState*)c->async;
  AsyncState* st = new AsyncState();
  if (hipGetDevice(&st->device) != hipSuccess) { delete st; return nullptr; }
  if (hipHostMalloc((void**)&st->flags, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { delete st; return nullptr; }
  memset(st->flags, 0, 64);
  st->worker = std::thread(async_worker, c, st);
  c->async = st;
  return st;
}

static ncclResult_t run_plan(mockComm* c, hipStream_t stream, const Plan& pl) {
  if (!c) return fail(ncclInvalidArgument, "null communicator");
  return async_mode() ? run_async(c, stream, pl) : run_sync(c, stream, pl);
}

extern "C" {

const char* ncclGetErrorString(ncclResult_t r) {
  if (r == ncclSuccess) return "no error";
  return t_err.c_str();
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return fail(ncclInvalidArgument, "null id");
  static std::atomic<uint32_t> seq{0};
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "mockrccl-%d-%u-%llx", (int)getpid(), seq.fetch_add(1),
           (unsigned long long)(now_s() * 1e6));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "bad rank / size");
  id.internal[127] = 0;
  if (strncmp(id.internal, "mockrccl-", 9) != 0) return fail(ncclInvalidArgument, "the unique id was not made by this library");
  mockComm* c = attach(id.internal, nranks, rank);
  if (!c) return fail(ncclSystemError, "cannot create the shared files under /dev/shm (%s)", strerror(errno));
  const double t0 = now_s();
  while (c->C()->arrived.load() < nranks) {  // like RCCL: returns once every rank has joined
    if (now_s() - t0 > TIMEOUT_S) return fail(ncclSystemError, "only %s%lld of %lld ranks joined", "", c->C()->arrived.load(), nranks);
    nap();
  }
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  (void)devlist;  // a device listed twice is the point of this library
  if (!comms || ndev < 1 || ndev > MAX_RANKS) return fail(ncclInvalidArgument, "bad device count");
  ncclUniqueId id;
  ncclGetUniqueId(&id);
  for (int r = 0; r < ndev; r++) {
    comms[r] = attach(id.internal, ndev, r);
    if (!comms[r]) return fail(ncclSystemError, "cannot create the shared files under /dev/shm (%s)", strerror(errno));
  }
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->async) {  // every collective enqueued on this communicator has been handed to the helper: let it finish them, then stop it
    AsyncState* st = (AsyncState*)c->async;
    {
      std::lock_guard<std::mutex> lk(st->mu);
      st->stop = true;
    }
    st->cv.notify_all();
    if (st->worker.joinable()) st->worker.join();
    if (st->flags && st->flags[1]) {
      g_kernel_timeouts++;
      fprintf(stderr, "[mock_rccl] rank %d: a wait kernel gave up after its timeout: its collective handed back poison\n", c->rank);
    }
    // (tasks, events and pinned buffers are left to the process: their copies may still be in flight on the caller's stream)
    c->async = nullptr;
  }
  Control* C = c->C();
  const bool last = C->detached.fetch_add(1) + 1 == c->nranks;
  for (auto& m : c->peer) m.drop();
  c->mine.drop();
  unlink(file_of(c->name, c->rank).c_str());
  c->ctl.drop();
  if (last) unlink(file_of(c->name, -1).c_str());
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t stream) {
  const size_t es = dtype_size(dtype), bytes = count * es;
  if (op != ncclSum) return fail(ncclInvalidArgument, "only ncclSum");
  if (!es) return fail(ncclInvalidArgument, "datatype");
  if (!c) return fail(ncclInvalidArgument, "null communicator");
  if (!device_range_ok(sendbuff, bytes, "ncclAllReduce sendbuff") || !device_range_ok(recvbuff, bytes, "ncclAllReduce recvbuff")) {
    if (!async_mode()) { c->C()->failed.store(1); (void)barrier(c); }
    return ncclInvalidArgument;
  }
  Plan pl;
  pl.op = OP_ALLREDUCE; pl.dtype = dtype; pl.count = count; pl.payload = bytes; pl.result = bytes; pl.es = es;
  pl.send.push_back(Piece{(void*)sendbuff, bytes, 0});
  pl.recv.push_back(Piece{recvbuff, bytes, 0});
  return run_plan(c, stream, pl);
}

ncclResult_t ncclReduceScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t dtype, ncclRedOp_t op,
                               ncclComm_t c, hipStream_t stream) {
  const size_t es = dtype_size(dtype);
  if (op != ncclSum) return fail(ncclInvalidArgument, "only ncclSum");
  if (!es || !c) return fail(ncclInvalidArgument, "datatype / communicator");
  const size_t chunk = recvcount * es, bytes = chunk * (size_t)c->nranks;
  ncclResult_t bad = ncclSuccess;
  if (!device_range_ok(sendbuff, bytes, "ncclReduceScatter sendbuff") || !device_range_ok(recvbuff, chunk, "ncclReduceScatter recvbuff"))
    bad = ncclInvalidArgument;
  if (bad == ncclSuccess) {
    // rccl.h: in place when recvbuff == sendbuff + rank * recvcount; any OTHER overlap of the two is undefined
    const char *s = (const char*)sendbuff, *d = (const char*)recvbuff;
    const bool overlap = d < s + bytes && s < d + chunk;
    if (overlap && d != s + (size_t)c->rank * chunk)
      bad = fail(ncclInvalidUsage, "ncclReduceScatter: recvbuff overlaps sendbuff but is not sendbuff + rank * recvcount");
  }
  if (bad != ncclSuccess) {
    if (!async_mode()) { c->C()->failed.store(1); (void)barrier(c); }
    return bad;
  }
  Plan pl;
  pl.op = OP_REDUCESCATTER; pl.dtype = dtype; pl.count = recvcount; pl.payload = bytes; pl.result = chunk; pl.es = es;
  pl.send.push_back(Piece{(void*)sendbuff, bytes, 0});
  pl.recv.push_back(Piece{recvbuff, chunk, 0});
  return run_plan(c, stream, pl);
}

ncclResult_t ncclAllToAllv(const void* sendbuff, const size_t sendcounts[], const size_t sdispls[], void* recvbuff,
                           const size_t recvcounts[], const size_t rdispls[], ncclDataType_t dtype, ncclComm_t c, hipStream_t stream) {
  const size_t es = dtype_size(dtype);
  if (!es || !c || !sendcounts || !sdispls || !recvcounts || !rdispls) return fail(ncclInvalidArgument, "null argument / datatype");
  const int R = c->nranks;
  Plan pl;
  pl.op = OP_ALLTOALLV; pl.dtype = dtype; pl.count = 0; pl.es = es;
  ncclResult_t bad = ncclSuccess;
  size_t o = 0, ores = 0;
  for (int d = 0; d < R && bad == ncclSuccess; d++) {
    const size_t b = sendcounts[d] * es, rb = recvcounts[d] * es;
    if (!device_range_ok((const char*)sendbuff + sdispls[d] * es, b, "ncclAllToAllv send piece")) bad = ncclInvalidArgument;
    else if (!device_range_ok((char*)recvbuff + rdispls[d] * es, rb, "ncclAllToAllv receive piece")) bad = ncclInvalidArgument;
    pl.a2a_off[d] = o; pl.a2a_cnt[d] = sendcounts[d];
    pl.send.push_back(Piece{(void*)((const char*)sendbuff + sdispls[d] * es), b, o});
    pl.recv.push_back(Piece{(char*)recvbuff + rdispls[d] * es, rb, ores});
    pl.a2a_recvcounts.push_back(recvcounts[d]);
    o += b;
    ores += rb;
  }
  // receive pieces must not overlap each other (a displacement in bytes instead of elements makes them collide or run off the end)
  for (int a = 0; a < R && bad == ncclSuccess; a++)
    for (int b = a + 1; b < R; b++) {
      const size_t a0 = rdispls[a], a1 = a0 + recvcounts[a], b0 = rdispls[b], b1 = b0 + recvcounts[b];
      if (recvcounts[a] && recvcounts[b] && a0 < b1 && b0 < a1) { bad = fail(ncclInvalidUsage, "ncclAllToAllv: receive pieces overlap"); break; }
    }
  if (bad != ncclSuccess) {
    if (!async_mode()) { c->C()->failed.store(1); (void)barrier(c); }
    return bad;
  }
  pl.payload = o;
  pl.result = ores;
  return run_plan(c, stream, pl);
}

// for the tests: collectives this PROCESS has completed over the mock, by kind {all, all-reduce, reduce-scatter, all-to-all}
void mock_rccl_stats(uint64_t out[4]) {
  out[0] = g_total_ops.load();
  out[1] = g_counts[OP_ALLREDUCE].load();
  out[2] = g_counts[OP_REDUCESCATTER].load();
  out[3] = g_counts[OP_ALLTOALLV].load();
}
// the stream-ordered mode: {1 if it is on, collectives completed behind their calls, of those failed, wait kernels that timed out}
void mock_rccl_async_stats(uint64_t out[4]) {
  out[0] = async_mode() ? 1 : 0;
  out[1] = g_async_ops.load();
  out[2] = g_async_errors.load();
  out[3] = g_kernel_timeouts.load();
}

}  // extern "C"
