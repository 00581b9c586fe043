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
// communicator (arrival counter, barrier, per-rank operation records) and one payload file per rank.  Every operation is
// synchronous -- wait for the caller's stream, copy the send buffer to the rank's payload file, barrier, check that all
// ranks posted the SAME operation with the same count and type, combine on the host in rank order, copy the result to the
// receive buffer, barrier -- so what this proves is the call sites' arguments (units, offsets, in-place use, buffer
// extents, the order of collectives on every rank); what it does not prove is stream ordering against later kernels,
// RCCL's own kernels, or anything about xGMI.  Every wait has a timeout: a rank that never arrives is an error
// (ncclSystemError), not a hang.  Every buffer is checked to be device memory and the accessed range to lie inside its
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
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

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
    C->bar_count.store(0, std::memory_order_relaxed);
    C->bar_gen.store(gen + 1, std::memory_order_release);
    return C->failed.load() == 0;
  }
  const double t0 = now_s();
  while (C->bar_gen.load(std::memory_order_acquire) == gen) {
    if (now_s() - t0 > TIMEOUT_S) { C->failed.store(1); return false; }
    nap();
  }
  return C->failed.load() == 0;
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

// common frame of a collective: post the record + payload, barrier, run `combine` (reads the peers), barrier
template <typename Post, typename Combine>
static ncclResult_t collective(mockComm* c, hipStream_t stream, Op op, ncclDataType_t dtype, uint64_t count, size_t payload, Post post,
                               Combine combine) {
  if (!c) return fail(ncclInvalidArgument, "null communicator");
  Control* C = c->C();
  ncclResult_t rc = ncclSuccess;
  if (hipStreamSynchronize(stream) != hipSuccess) rc = fail(ncclUnhandledCudaError, "hipStreamSynchronize failed before the collective");
  if (rc == ncclSuccess && !grow_mine(c, payload)) rc = fail(ncclSystemError, "cannot grow the payload file of rank %s%lld", "", c->rank);
  RankRecord& me = C->rec[c->rank];
  me.op = op; me.dtype = (int32_t)dtype; me.count = count; me.bytes = payload;
  if (rc == ncclSuccess) rc = post(me);
  if (rc != ncclSuccess) C->failed.store(1);
  if (!barrier(c)) return rc != ncclSuccess ? rc : fail(ncclSystemError, "a rank failed or did not arrive (operation %s%lld)", "", op);
  // every rank must have posted the same operation (a rank in another collective is the classic multi-GPU hang)
  for (int r = 0; r < c->nranks && rc == ncclSuccess; r++) {
    const RankRecord& o = C->rec[r];
    if (o.op != op || o.dtype != (int32_t)dtype || (op != OP_ALLTOALLV && o.count != count))
      rc = fail(ncclInvalidUsage, "rank %s%lld posted another operation / count / type than this rank", "", r);
  }
  if (rc == ncclSuccess) rc = combine();
  if (rc != ncclSuccess) C->failed.store(1);
  if (!barrier(c)) return rc != ncclSuccess ? rc : fail(ncclSystemError, "a rank failed inside operation %s%lld", "", op);
  if (c->rank == 0) C->ops++;
  g_total_ops++;
  g_counts[op]++;
  return rc;
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
  return collective(
      c, stream, OP_ALLREDUCE, dtype, count, bytes,
      [&](RankRecord&) -> ncclResult_t {
        if (!device_range_ok(sendbuff, bytes, "ncclAllReduce sendbuff") || !device_range_ok(recvbuff, bytes, "ncclAllReduce recvbuff"))
          return ncclInvalidArgument;
        if (bytes && hipMemcpy(c->mine.p, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
          return fail(ncclUnhandledCudaError, "device -> host copy failed");
        return ncclSuccess;
      },
      [&]() -> ncclResult_t {
        std::vector<uint8_t> acc(bytes ? bytes : 1, 0);
        for (int r = 0; r < c->nranks; r++) {  // rank order: the same bits on every rank
          const void* p = peer_payload(c, r, bytes);
          if (!p) return fail(ncclSystemError, "cannot map the payload of rank %s%lld", "", r);
          if (!sum_typed(acc.data(), p, count, dtype)) return fail(ncclInvalidArgument, "datatype cannot be summed");
        }
        if (bytes && hipMemcpy(recvbuff, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
          return fail(ncclUnhandledCudaError, "host -> device copy failed");
        c->n_allreduce++; c->bytes_moved += bytes;
        return ncclSuccess;
      });
}

ncclResult_t ncclReduceScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t dtype, ncclRedOp_t op,
                               ncclComm_t c, hipStream_t stream) {
  const size_t es = dtype_size(dtype);
  if (op != ncclSum) return fail(ncclInvalidArgument, "only ncclSum");
  if (!es || !c) return fail(ncclInvalidArgument, "datatype / communicator");
  const size_t chunk = recvcount * es, bytes = chunk * (size_t)c->nranks;
  return collective(
      c, stream, OP_REDUCESCATTER, dtype, recvcount, bytes,
      [&](RankRecord&) -> ncclResult_t {
        if (!device_range_ok(sendbuff, bytes, "ncclReduceScatter sendbuff") || !device_range_ok(recvbuff, chunk, "ncclReduceScatter recvbuff"))
          return ncclInvalidArgument;
        // rccl.h: in place when recvbuff == sendbuff + rank * recvcount; any OTHER overlap of the two is undefined
        const char *s = (const char*)sendbuff, *d = (const char*)recvbuff;
        const bool overlap = d < s + bytes && s < d + chunk;
        if (overlap && d != s + (size_t)c->rank * chunk)
          return fail(ncclInvalidUsage, "ncclReduceScatter: recvbuff overlaps sendbuff but is not sendbuff + rank * recvcount");
        if (bytes && hipMemcpy(c->mine.p, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
          return fail(ncclUnhandledCudaError, "device -> host copy failed");
        return ncclSuccess;
      },
      [&]() -> ncclResult_t {
        std::vector<uint8_t> acc(chunk ? chunk : 1, 0);
        for (int r = 0; r < c->nranks; r++) {
          const uint8_t* p = (const uint8_t*)peer_payload(c, r, bytes);
          if (!p) return fail(ncclSystemError, "cannot map the payload of rank %s%lld", "", r);
          if (!sum_typed(acc.data(), p + (size_t)c->rank * chunk, recvcount, dtype)) return fail(ncclInvalidArgument, "datatype cannot be summed");
        }
        if (chunk && hipMemcpy(recvbuff, acc.data(), chunk, hipMemcpyHostToDevice) != hipSuccess)
          return fail(ncclUnhandledCudaError, "host -> device copy failed");
        c->n_reducescatter++; c->bytes_moved += bytes;
        return ncclSuccess;
      });
}

ncclResult_t ncclAllToAllv(const void* sendbuff, const size_t sendcounts[], const size_t sdispls[], void* recvbuff,
                           const size_t recvcounts[], const size_t rdispls[], ncclDataType_t dtype, ncclComm_t c, hipStream_t stream) {
  const size_t es = dtype_size(dtype);
  if (!es || !c || !sendcounts || !sdispls || !recvcounts || !rdispls) return fail(ncclInvalidArgument, "null argument / datatype");
  const int R = c->nranks;
  size_t total = 0;
  for (int d = 0; d < R; d++) total += sendcounts[d] * es;
  return collective(
      c, stream, OP_ALLTOALLV, dtype, 0, total,
      [&](RankRecord& me) -> ncclResult_t {
        size_t o = 0;
        for (int d = 0; d < R; d++) {
          const size_t b = sendcounts[d] * es;
          if (!device_range_ok((const char*)sendbuff + sdispls[d] * es, b, "ncclAllToAllv send piece")) return ncclInvalidArgument;
          if (!device_range_ok((char*)recvbuff + rdispls[d] * es, recvcounts[d] * es, "ncclAllToAllv receive piece")) return ncclInvalidArgument;
          me.a2a_off[d] = o; me.a2a_cnt[d] = sendcounts[d];
          if (b && hipMemcpy((char*)c->mine.p + o, (const char*)sendbuff + sdispls[d] * es, b, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ncclUnhandledCudaError, "device -> host copy failed");
          o += b;
        }
        // receive pieces must not overlap each other (a displacement in bytes instead of elements makes them collide or
        // run off the end)
        for (int a = 0; a < R; a++)
          for (int b = a + 1; b < R; b++) {
            const size_t a0 = rdispls[a], a1 = a0 + recvcounts[a], b0 = rdispls[b], b1 = b0 + recvcounts[b];
            if (recvcounts[a] && recvcounts[b] && a0 < b1 && b0 < a1) return fail(ncclInvalidUsage, "ncclAllToAllv: receive pieces overlap");
          }
        return ncclSuccess;
      },
      [&]() -> ncclResult_t {
        Control* C = c->C();
        for (int s = 0; s < R; s++) {
          const RankRecord& o = C->rec[s];
          if (o.a2a_cnt[c->rank] != recvcounts[s])
            return fail(ncclInvalidUsage, "ncclAllToAllv: rank %s%lld sends %lld elements here, another count is expected", "", s,
                        (long long)o.a2a_cnt[c->rank]);
          const size_t b = recvcounts[s] * es;
          if (!b) continue;
          const uint8_t* p = (const uint8_t*)peer_payload(c, s, o.bytes);
          if (!p) return fail(ncclSystemError, "cannot map the payload of rank %s%lld", "", s);
          if (hipMemcpy((char*)recvbuff + rdispls[s] * es, p + o.a2a_off[c->rank], b, hipMemcpyHostToDevice) != hipSuccess)
            return fail(ncclUnhandledCudaError, "host -> device copy failed");
          c->bytes_moved += b;
        }
        c->n_alltoallv++;
        return ncclSuccess;
      });
}

// for the tests: collectives this PROCESS has completed over the mock, by kind {all, all-reduce, reduce-scatter, all-to-all}
void mock_rccl_stats(uint64_t out[4]) {
  out[0] = g_total_ops.load();
  out[1] = g_counts[OP_ALLREDUCE].load();
  out[2] = g_counts[OP_REDUCESCATTER].load();
  out[3] = g_counts[OP_ALLTOALLV].load();
}

}  // extern "C"
