// runtime.hip -- context, error state, HIP-event profiling, FBM residency, view creation.
#include <ctype.h>
#include <fcntl.h>
#include <math.h>
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "common.h"
#include "host/host_bedpack.h"
#include "host/host_nibpack.h"

#include <mutex>
#include <thread>
#include <unordered_map>

static thread_local char g_err[1024] = "";

// ---------------------------------------------------------------------------
// device-memory pools, one per context (see common.h)
struct Pool {
  std::multimap<size_t, void*> free_blocks;  // rounded size -> block
  size_t cached_bytes = 0;
  int device = 0;
  bool alive = true;  // false once its context is destroyed: blocks still out are hipFree'd when they come back
};
struct LiveRec {
  size_t rb;
  int pool;
};
static std::mutex g_pool_mu;
static std::unordered_map<int, Pool> g_pools;        // pool id -> pool (id 0: calls made outside any context)
static std::unordered_map<void*, LiveRec> g_live;    // block handed out -> its size and home pool
static int g_next_pool_id = 1;
static const size_t POOL_MAX_CACHED = (size_t)48 << 30;
static thread_local tpg_ctx* g_cur_ctx = nullptr;

TpgEnter::TpgEnter(tpg_ctx* ctx) : prev(g_cur_ctx) {
  if (!ctx) return;
  g_cur_ctx = ctx;
  (void)hipSetDevice(ctx->device);  // HIP's current device is per host thread
}
TpgEnter::~TpgEnter() {
  if (prev && prev != g_cur_ctx) (void)hipSetDevice(prev->device);
  g_cur_ctx = prev;
}
tpg_ctx* tpg_current_ctx() { return g_cur_ctx; }

static size_t pool_round(size_t b) {
  if (b < 256) return 256;
  if (b < ((size_t)1 << 20)) return (b + 255) & ~(size_t)255;
  return (b + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
}

hipError_t tpg_pmalloc(void** p, size_t bytes) {
  const size_t rb = pool_round(bytes);
  const int pid = g_cur_ctx ? g_cur_ctx->pool_id : 0;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    Pool& pool = g_pools[pid];
    auto it = pool.free_blocks.find(rb);
    if (it != pool.free_blocks.end()) {
      *p = it->second;
      pool.free_blocks.erase(it);
      pool.cached_bytes -= rb;
      g_live[*p] = LiveRec{rb, pid};
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, rb);
  if (e != hipSuccess) {  // out of memory: drop this pool's cache and retry once
    (void)hipGetLastError();
    tpg_pool_trim(pid);
    e = hipMalloc(p, rb);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_live[*p] = LiveRec{rb, pid};
  return hipSuccess;
}

void tpg_pfree(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_live.find(p);
    if (it != g_live.end()) {
      const LiveRec rec = it->second;
      g_live.erase(it);
      auto pit = g_pools.find(rec.pool);
      if (pit != g_pools.end() && pit->second.alive && pit->second.cached_bytes + rec.rb <= POOL_MAX_CACHED) {
        pit->second.free_blocks.emplace(rec.rb, p);
        pit->second.cached_bytes += rec.rb;
        return;
      }
    }
  }
  (void)hipFree(p);
}

void tpg_pool_trim(int pool_id) {
  std::vector<void*> blocks;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto pit = g_pools.find(pool_id);
    if (pit == g_pools.end()) return;
    for (auto& kv : pit->second.free_blocks) blocks.push_back(kv.second);
    pit->second.free_blocks.clear();
    pit->second.cached_bytes = 0;
  }
  for (void* b : blocks) (void)hipFree(b);
}

static int pool_open(int device) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  const int id = g_next_pool_id++;
  g_pools[id].device = device;
  return id;
}

static void pool_close(int pool_id) {  // cached blocks are released now, blocks still out when they are freed
  tpg_pool_trim(pool_id);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto pit = g_pools.find(pool_id);
  if (pit != g_pools.end()) pit->second.alive = false;
}

void tpg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tpg_last_error(void) { return g_err; }
extern "C" const char* tpg_version(void) { return "tidypopgen_amd 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------
ProfScope::ProfScope(tpg_ctx* c, const char* name) : ctx(c), on(c->prof) {
  if (on && !c->prof_only.empty() && !c->prof_only.count(name)) on = false;
  if (!on) return;
  rec.name = name;
  auto get = [&]() {
    hipEvent_t e;
    if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  };
  rec.start = get();
  rec.stop = get();
  (void)hipEventRecord(rec.start, ctx->stream);
}
ProfScope::~ProfScope() {
  if (!on) return;
  (void)hipEventRecord(rec.stop, ctx->stream);
  ctx->prof_pending.push_back(rec);
}

int tpg_prof_resolve(tpg_ctx* ctx) {
  if (ctx->prof_pending.empty()) return TPG_OK;
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->prof_pending) {
    float ms = 0;
    TPG_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
    auto& a = ctx->prof_acc[r.name];
    a.first += ms;
    a.second += 1;
    ctx->event_pool.push_back(r.start);
    ctx->event_pool.push_back(r.stop);
  }
  ctx->prof_pending.clear();
  return TPG_OK;
}

extern "C" int tpg_prof_enable(tpg_ctx* ctx, int on) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx, TPG_EINVAL, "null ctx");
  TPG_TRY(tpg_prof_resolve(ctx));
  ctx->prof = on != 0;
  return TPG_OK;
}
extern "C" int tpg_prof_only(tpg_ctx* ctx, const char* names_csv) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx, TPG_EINVAL, "null ctx");
  TPG_TRY(tpg_prof_resolve(ctx));
  ctx->prof_only.clear();
  for (const char* p = names_csv; p && *p;) {
    const char* q = strchr(p, ',');
    const size_t len = q ? (size_t)(q - p) : strlen(p);
    if (len) ctx->prof_only.insert(std::string(p, len));
    p = q ? q + 1 : p + len;
  }
  return TPG_OK;
}
extern "C" int tpg_prof_reset(tpg_ctx* ctx) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx, TPG_EINVAL, "null ctx");
  TPG_TRY(tpg_prof_resolve(ctx));
  ctx->prof_acc.clear();
  return TPG_OK;
}
extern "C" int tpg_prof_get(tpg_ctx* ctx, const char* prefix, double* total_ms, int64_t* launches) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && prefix, TPG_EINVAL, "null argument");
  TPG_TRY(tpg_prof_resolve(ctx));
  double ms = 0;
  int64_t n = 0;
  size_t pl = strlen(prefix);
  for (auto& kv : ctx->prof_acc)
    if (kv.first.compare(0, pl, prefix) == 0) { ms += kv.second.first; n += kv.second.second; }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  return TPG_OK;
}
extern "C" int tpg_prof_dump(tpg_ctx* ctx, char* buf, size_t cap) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && buf && cap > 0, TPG_EINVAL, "null argument");
  TPG_TRY(tpg_prof_resolve(ctx));
  size_t off = 0;
  buf[0] = 0;
  for (auto& kv : ctx->prof_acc) {
    int w = snprintf(buf + off, cap - off, "%s\t%lld\t%.6f\n", kv.first.c_str(), (long long)kv.second.second,
                     kv.second.first);
    if (w < 0 || (size_t)w >= cap - off) break;
    off += (size_t)w;
  }
  return TPG_OK;
}

// ---------------------------------------------------------------------------
extern "C" int tpg_device_count(int* count) {
  TPG_REQUIRE(count, TPG_EINVAL, "null argument");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  *count = e == hipSuccess ? c : 0;
  return TPG_OK;
}

static constexpr int XFER_THREADS_MIN_CPUS = 16;  // a node must offer the teams' sixteen threads a CPU each

extern "C" int tpg_ctx_create(int device, tpg_ctx** out) {
  TPG_REQUIRE(out, TPG_EINVAL, "null out");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    tpg_set_error("no HIP device available (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return TPG_EHIP;
  }
  TPG_REQUIRE(device >= 0 && device < count, TPG_EINVAL, "device %d out of range [0,%d)", device, count);
  TPG_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  TPG_HIP(hipGetDeviceProperties(&prop, device));
  tpg_ctx* c = new tpg_ctx();
  c->device = device;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  hipError_t es = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (es != hipSuccess) { delete c; tpg_set_error("hipStreamCreate: %s", hipGetErrorString(es)); return TPG_EHIP; }
  c->own_stream = true;
  c->pool_id = pool_open(device);
  *out = c;
  return TPG_OK;
}

// "0-63,128-191" -> the CPUs of the list that are also in `allowed`
static void parse_cpulist(const char* line, const cpu_set_t* allowed, cpu_set_t* out) {
  CPU_ZERO(out);
  for (const char* q = line; *q;) {
    char* end = nullptr;
    const long a = strtol(q, &end, 10);
    if (end == q) break;
    long b = a;
    if (*end == '-') { q = end + 1; b = strtol(q, &end, 10); }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++)
      if (c >= 0 && CPU_ISSET((int)c, allowed)) CPU_SET((int)c, out);
    if (*end != ',') break;
    q = end + 1;
  }
}

extern "C" int tpg_host_bind_near_device(int device, int* node_out) {
  if (node_out) *node_out = -1;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    tpg_set_error("no HIP device available (%s)", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return TPG_EHIP;
  }
  TPG_REQUIRE(device >= 0 && device < count, TPG_EINVAL, "device %d out of range [0,%d)", device, count);
  char bus[64] = {0};
  TPG_HIP(hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, device));  // "0000:c1:00.0"
  for (char* q = bus; *q; q++) *q = (char)tolower((unsigned char)*q);
  char path[160], line[4096];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  FILE* f = fopen(path, "r");
  int node = -1;
  if (f) {
    if (fgets(line, sizeof(line), f)) node = atoi(line);
    fclose(f);
  }
  if (node < 0) return TPG_OK;  // unknown (or a one-node host that says -1)
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* g = fopen("/sys/devices/system/node/node1/cpulist", "r");  // a second node at all?
  if (!g) return TPG_OK;
  fclose(g);
  f = fopen(path, "r");
  if (!f) return TPG_OK;
  cpu_set_t allowed, mine;
  CPU_ZERO(&mine);
  const bool ok = fgets(line, sizeof(line), f) && sched_getaffinity(0, sizeof(allowed), &allowed) == 0;
  fclose(f);
  if (!ok) return TPG_OK;
  parse_cpulist(line, &allowed, &mine);
  if (CPU_COUNT(&mine) < XFER_THREADS_MIN_CPUS) return TPG_OK;
  if (sched_setaffinity(0, sizeof(mine), &mine) != 0) return TPG_OK;  // (0: the calling thread; threads started later inherit)
  if (node_out) *node_out = node;
  return TPG_OK;
}

extern "C" void tpg_ctx_destroy(tpg_ctx* ctx) {
  if (!ctx) return;
  {
    TpgEnter _enter(ctx);
    (void)tpg_prof_resolve(ctx);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->h2d_pinned) {
      for (int k = 0; k < tpg_ctx::H2D_SLOTS; k++)
        if (ctx->h2d_done[k]) (void)hipEventDestroy(ctx->h2d_done[k]);
      (void)hipHostFree(ctx->h2d_pinned);
    }
    tpg_resident_release(ctx);
    if (ctx->mail_host) (void)hipHostFree(ctx->mail_host);
    pool_close(ctx->pool_id);  // only this context's blocks
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  }
  if (g_cur_ctx == ctx) g_cur_ctx = nullptr;
  delete ctx;
}

extern "C" int tpg_ctx_set_stream(tpg_ctx* ctx, void* hip_stream) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx, TPG_EINVAL, "null ctx");
  TPG_TRY(tpg_prof_resolve(ctx));
  // the pool's blocks are ordered on the old stream: drain it, whoever owns it, before work moves to the new one
  if (ctx->stream) TPG_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->stream) TPG_HIP(hipStreamDestroy(ctx->stream));
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
  } else {
    TPG_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  return TPG_OK;
}

extern "C" int tpg_ctx_sync(tpg_ctx* ctx) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx, TPG_EINVAL, "null ctx");
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  return TPG_OK;
}

extern "C" int tpg_dev_alloc(tpg_ctx* ctx, size_t bytes, void** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && out, TPG_EINVAL, "null argument");
  TPG_HIP(hipSetDevice(ctx->device));
  TPG_HIP(tpg_pmalloc(out, bytes > 0 ? bytes : 16));
  return TPG_OK;
}
extern "C" void tpg_dev_free(void* p) {
  if (p) tpg_pfree(p);
}
extern "C" int tpg_dev_to_host(tpg_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && host_dst && dev_src, TPG_EINVAL, "null argument");
  TPG_HIP(tpg_download(ctx, host_dst, dev_src, bytes));
  return TPG_OK;
}

extern "C" int tpg_dev_from_host(tpg_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && dev_dst && host_src, TPG_EINVAL, "null argument");
  // public entry point: COMPLETE at return for every size (a caller may consume the buffer from a stream of its own, and a
  // copy error belongs to this call); the library's own small inputs take tpg_h2d_async, which is only stream-ordered
  if (bytes <= tpg_ctx::MAIL_PUSH_MAX) TPG_HIP(tpg_push_small(ctx, dev_dst, host_src, bytes));
  else TPG_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  return TPG_OK;
}

// ---------------------------------------------------------------------------
bool tpg_is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // clear the sticky "invalid value" for plain host memory
    return false;
  }
  return attr.type == hipMemoryTypeDevice;
}

int OutBuf::init(void* user_ptr, size_t nbytes) {
  user = user_ptr;
  bytes = nbytes;
  if (tpg_is_device_ptr(user_ptr)) {
    d = user_ptr;
    owned = false;
  } else {
    TPG_HIP(tpg_pmalloc(&d, nbytes > 0 ? nbytes : 16));
    owned = true;
  }
  return TPG_OK;
}
int OutBuf::commit(tpg_ctx* ctx) {
  if (owned && user && bytes) TPG_HIP(tpg_download(ctx, user, d, bytes));
  return TPG_OK;
}
OutBuf::~OutBuf() {
  if (owned && d) tpg_pfree(d);
}

static hipError_t h2d_engine(tpg_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  if (bytes > tpg_ctx::H2D_SLOT_BYTES) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    return e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
  }
  if (!ctx->h2d_pinned) {
    hipError_t e = hipHostMalloc((void**)&ctx->h2d_pinned, tpg_ctx::H2D_SLOT_BYTES * tpg_ctx::H2D_SLOTS, hipHostMallocDefault);
    if (e != hipSuccess) { ctx->h2d_pinned = nullptr; return e; }
    for (int k = 0; k < tpg_ctx::H2D_SLOTS; k++) {
      e = hipEventCreateWithFlags(&ctx->h2d_done[k], hipEventDisableTiming);
      if (e != hipSuccess) return e;
    }
  }
  const int k = ctx->h2d_next;
  ctx->h2d_next = (k + 1) % tpg_ctx::H2D_SLOTS;
  if (ctx->h2d_used[k]) {  // the copy that last used this slot (8 copies ago) has long finished; make sure
    hipError_t e = hipEventSynchronize(ctx->h2d_done[k]);
    if (e != hipSuccess) return e;
  }
  uint8_t* slot = ctx->h2d_pinned + (size_t)k * tpg_ctx::H2D_SLOT_BYTES;
  memcpy(slot, src, bytes);
  hipError_t e = hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipEventRecord(ctx->h2d_done[k], ctx->stream);
  ctx->h2d_used[k] = true;
  return e;
}

// ---------------------------------------------------------------------------
// The mailbox (common.h): small transfers by kernels through coherent pinned memory.
__global__ __launch_bounds__(1024) void tpg_mail_fetch_kernel(const uint32_t* __restrict__ src, uint32_t* dst, size_t nwords,
                                                               uint32_t* flag, uint32_t seq) {
  for (size_t i = threadIdx.x; i < nwords; i += 1024) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(1024) void tpg_mail_push_kernel(const uint32_t* src, uint32_t* __restrict__ dst, size_t nwords,
                                                              uint32_t* flag, uint32_t seq) {
  for (size_t i = threadIdx.x; i < nwords; i += 1024) dst[i] = __builtin_nontemporal_load(src + i);
  __syncthreads();  // every word has been READ: the host may write the ring slot again
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(256) void tpg_copy_dev_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16,
                                                           const uint8_t* __restrict__ srcb, uint8_t* __restrict__ dstb, size_t tail) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x < tail) dstb[n16 * 16 + threadIdx.x] = srcb[n16 * 16 + threadIdx.x];
}

static hipError_t mail_init(tpg_ctx* ctx) {
  if (ctx->mail_host) return hipSuccess;
  const size_t total = 64 + tpg_ctx::MAIL_FETCH_BYTES + tpg_ctx::MAIL_PUSH_BYTES;
  void* h = nullptr;
  hipError_t e = hipHostMalloc(&h, total, hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) return e;
  void* d = nullptr;
  e = hipHostGetDevicePointer(&d, h, 0);
  if (e != hipSuccess) { (void)hipHostFree(h); return e; }
  memset(h, 0, 64);
  ctx->mail_host = (uint8_t*)h;
  ctx->mail_dev = (uint8_t*)d;
  return hipSuccess;
}

// wait until the 32-bit word at `flag` has reached `seq` (sequence numbers wrap: compare as a signed difference); the
// stream is looked at now and then, so that a failed launch does not leave the host spinning for ever
static hipError_t mail_wait(tpg_ctx* ctx, const volatile uint32_t* flag, uint32_t seq) {
  int drained = 0;
  for (uint64_t spins = 1;; spins++) {
    if ((int32_t)(__atomic_load_n(flag, __ATOMIC_ACQUIRE) - seq) >= 0) return hipSuccess;
    __builtin_ia32_pause();
    if ((spins & 0x3FFF) == 0) {
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q == hipErrorNotReady) continue;
      if (q != hipSuccess) return q;
      if (++drained > 2) return hipErrorUnknown;  // the stream is empty and the kernel never wrote: a lost launch
    }
  }
}

hipError_t tpg_fetch_small(tpg_ctx* ctx, void* host_dst, const void* d_src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  static const bool off = getenv("TPG_NO_MAILBOX") != nullptr;  // (A/B: the copy engine + a stream synchronisation)
  if (off || bytes > tpg_ctx::MAIL_FETCH_BYTES || (bytes & 3) || ((uintptr_t)d_src & 3) || mail_init(ctx) != hipSuccess) {
    hipError_t e = hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    return e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
  }
  const uint32_t seq = ++ctx->mail_fetch_seq;
  hipLaunchKernelGGL(tpg_mail_fetch_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)d_src,
                     (uint32_t*)(ctx->mail_dev + 64), bytes / 4, (uint32_t*)ctx->mail_dev, seq);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = mail_wait(ctx, (const volatile uint32_t*)ctx->mail_host, seq);
  if (e != hipSuccess) return e;
  memcpy(host_dst, ctx->mail_host + 64, bytes);
  return hipSuccess;
}

hipError_t tpg_push_small(tpg_ctx* ctx, void* d_dst, const void* host_src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  static const bool off = getenv("TPG_NO_MAILBOX") != nullptr;
  if (off || bytes > tpg_ctx::MAIL_PUSH_MAX || (bytes & 3) || ((uintptr_t)d_dst & 3) || mail_init(ctx) != hipSuccess)
    return h2d_engine(ctx, d_dst, host_src, bytes);
  const size_t need = (bytes + 63) & ~(size_t)63;
  volatile uint32_t* done = (volatile uint32_t*)(ctx->mail_host + 4);  // sequence number of the last push that has been read
  if (ctx->mail_push_off + need > tpg_ctx::MAIL_PUSH_BYTES) {
    // a new lap of the ring: every push of the lap before must have been read (it has, long ago, unless the host has run
    // MAIL_PUSH_BYTES ahead of the device)
    ctx->mail_push_off = 0;
    ctx->mail_push_wrap_seq = ctx->mail_push_seq;
  }
  if (ctx->mail_push_wrap_seq != 0) {
    const hipError_t e = mail_wait(ctx, done, ctx->mail_push_wrap_seq);
    if (e != hipSuccess) return e;
  }
  const size_t off_ring = 64 + tpg_ctx::MAIL_FETCH_BYTES + ctx->mail_push_off;
  memcpy(ctx->mail_host + off_ring, host_src, bytes);
  __atomic_thread_fence(__ATOMIC_RELEASE);
  ctx->mail_push_off += need;
  const uint32_t seq = ++ctx->mail_push_seq;
  hipLaunchKernelGGL(tpg_mail_push_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)(ctx->mail_dev + off_ring),
                     (uint32_t*)d_dst, bytes / 4, (uint32_t*)(ctx->mail_dev + 4), seq);
  return hipGetLastError();
}

// (small inputs take the mailbox, the others the copy engine from a pinned slot)
hipError_t tpg_h2d_async(tpg_ctx* ctx, void* dst, const void* src, size_t bytes) { return tpg_push_small(ctx, dst, src, bytes); }

hipError_t tpg_copy_dev(tpg_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
  if (bytes == 0 || d_dst == d_src) return hipSuccess;
  if (((uintptr_t)d_dst | (uintptr_t)d_src) & 15) return hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream);
  const size_t n16 = bytes / 16;
  const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>(1, (n16 + 1023) / 1024), (size_t)ctx->num_cu * 8);
  hipLaunchKernelGGL(tpg_copy_dev_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)d_src, (uint4*)d_dst, n16,
                     (const uint8_t*)d_src, (uint8_t*)d_dst, bytes & 15);
  return hipGetLastError();
}

int InBuf::init(tpg_ctx* ctx, const void* user_ptr, size_t nbytes) {
  if (tpg_is_device_ptr(user_ptr)) {
    d = user_ptr;
    return TPG_OK;
  }
  TPG_HIP(tpg_pmalloc(&owned_ptr, nbytes > 0 ? nbytes : 16));
  TPG_HIP(tpg_h2d_async(ctx, owned_ptr, user_ptr, nbytes));  // the caller's buffer is free once this returns
  d = owned_ptr;
  return TPG_OK;
}
InBuf::~InBuf() {
  if (owned_ptr) tpg_pfree(owned_ptr);
}

// ---------------------------------------------------------------------------
// Host <-> HBM bulk transfers, the cold-start cost of every analysis (5 GB for a 5 000 x 1 000 000 FBM).  On the
// MI355X boxes one hipMemcpy between device memory and pageable host memory whose pages are present runs at the
// PCIe rate (52-57 GB/s measured, tools/xfer_probe.py) -- faster than anything this library staged by hand through
// pinned slots (30-41 GB/s).  What makes a transfer slow is page faults inside it: a result matrix the caller has
// just allocated, or an FBM file mapping touched for the first time (19 and 13 GB/s measured).  So:
// large host buffers -- a backing file is simply mapped -- are first touched by XFER_THREADS threads in parallel (one
// access per 4-KiB page), then moved with ONE hipMemcpy.  (Reading the file with pread() into pinned or pageable staging
// buffers, piece by piece beside the DMA, measured 7-41 GB/s, and pinned staging adds a set-up cost to the first call.)
static constexpr int XFER_THREADS = 16;
static constexpr size_t XFER_BIG = 64u << 20;

template <typename F>  // f(thread, lo, hi) over [0, bytes) cut into page-aligned stripes
static void xfer_parallel(size_t bytes, F f) {
  const size_t pages = (bytes + 4095) / 4096;
  const int nth = (int)std::min<size_t>(XFER_THREADS, std::max<size_t>(1, pages / 256));
  if (nth == 1) { f(0, 0, bytes); return; }  // below 2 MiB: on the calling thread
  std::vector<std::thread> th;
  for (int t = 0; t < nth; t++) {
    const size_t lo = std::min(bytes, pages * (size_t)t / (size_t)nth * 4096), hi = std::min(bytes, pages * ((size_t)t + 1) / (size_t)nth * 4096);
    if (lo < hi) th.emplace_back([=]() { f(t, lo, hi); });
  }
  for (auto& t : th) t.join();
}

static uint8_t* nib_stage_acquire(bool may_pin = true);
static void nib_stage_release(uint8_t* p);
static constexpr size_t XFER_PIECE = 64u << 20;  // a quarter of the process's pinned staging buffer (NIB_CHUNK)

// device memory -> host memory the caller owns (pageable), large: through the process's pinned staging buffer in four pieces
// of 64 MiB -- the copy engine fills piece i + 1 ... i + 3 (55 GB/s into pinned memory) while the team of threads moves piece
// i into the caller's pages, which is also what faults those pages in (no separate pass over them).  The runtime's own path
// for a pageable destination (round 4: parallel page touch, then one hipMemcpy) moved the bytes at 45 - 53 GB/s but left the
// runtime with something to undo: the NEXT device -> host copy of the process, or the exit of the thread, stood still for
// 45 - 48 ms after 1.4 GB had gone down that way (seen with the end-to-end routes' downloads beside the step; bench.py
// TPG_E2E_TRACE=1).  TPG_DOWNLOAD_PINNED=0: the old path (A/B).
// (width / dpitch: the destination as `bytes / width` pieces of `width` bytes, `dpitch` bytes apart -- a block of rows of a
// column-major host matrix, tpg_download_rows; width == bytes: one contiguous piece)
static hipError_t tpg_download_pinned(tpg_ctx* ctx, uint8_t* dst, const uint8_t* src, size_t bytes, bool* done,
                                      size_t width = 0, size_t dpitch = 0) {
  *done = false;
  if (width == 0 || width >= bytes) { width = bytes; dpitch = bytes; }
  static const bool off = getenv("TPG_DOWNLOAD_PINNED") && atoi(getenv("TPG_DOWNLOAD_PINNED")) == 0;
  // (a copy below 64 MiB takes the staging buffer only if the process already has one: pinning 256 MiB costs 35 - 40 ms)
  uint8_t* const pinned = off ? nullptr : nib_stage_acquire(bytes >= XFER_BIG);
  if (!pinned) return hipSuccess;
  struct Back { uint8_t* p; ~Back() { nib_stage_release(p); } } back{pinned};
  const size_t H = XFER_PIECE, np = (bytes + H - 1) / H;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int k = 0; k < 4 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
  auto issue = [&](size_t i) -> hipError_t {
    const size_t a = i * H, len = std::min(bytes - a, H);
    hipError_t r = hipMemcpyAsync(pinned + (i & 3) * H, src + a, len, hipMemcpyDeviceToHost, ctx->stream);
    return r == hipSuccess ? hipEventRecord(ev[i & 3], ctx->stream) : r;
  };
  for (size_t i = 0; i < np && i < 4 && e == hipSuccess; i++) e = issue(i);
  for (size_t i = 0; i < np && e == hipSuccess; i++) {
    e = hipEventSynchronize(ev[i & 3]);
    if (e != hipSuccess) break;
    const size_t a = i * H, len = std::min(bytes - a, H);
    const uint8_t* pin = pinned + (i & 3) * H;
    if (width == bytes) {
      xfer_parallel(len, [=](int, size_t lo, size_t hi) { memcpy(dst + a + lo, pin + lo, hi - lo); });
    } else {
      xfer_parallel(len, [=](int, size_t lo, size_t hi) {
        for (size_t o = lo; o < hi;) {  // source offset a + o = piece (a + o) / width, byte (a + o) % width of it
          const size_t col = (a + o) / width, within = (a + o) % width, run = std::min(hi - o, width - within);
          memcpy(dst + col * dpitch + within, pin + o, run);
          o += run;
        }
      });
    }
    if (i + 4 < np) e = issue(i + 4);
  }
  if (e != hipSuccess) (void)hipStreamSynchronize(ctx->stream);
  for (int q = 0; q < 4; q++)
    if (ev[q]) (void)hipEventDestroy(ev[q]);
  *done = e == hipSuccess;
  return e;
}

// device memory -> host memory the caller owns (pageable)
// A large result buffer the caller has just allocated is first touched by the download: with 4-KiB pages a team of sixteen
// threads fills such a buffer at 16 GB/s on the pool's hosts (one fault per page), with transparent huge pages at 120
// (tools/thp_probe.cpp; the hosts run THP in `madvise` mode, so somebody has to ask: numpy does for its own arrays, R and malloc
// do not).  Advisory and harmless where it does not apply (pages already present, file mappings, THP off).
// TPG_DOWNLOAD_THP=0: do not ask (A/B).
static void advise_huge_pages(void* dst, size_t extent) {
#ifdef MADV_HUGEPAGE
  static const bool off = getenv("TPG_DOWNLOAD_THP") && atoi(getenv("TPG_DOWNLOAD_THP")) == 0;
  const uintptr_t H = 2u << 20, lo = ((uintptr_t)dst + H - 1) & ~(H - 1), hi = ((uintptr_t)dst + extent) & ~(H - 1);
  if (!off && extent >= (4u << 20) && hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
#else
  (void)dst; (void)extent;
#endif
}

hipError_t tpg_download(tpg_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes <= tpg_ctx::MAIL_FETCH_BYTES) return tpg_fetch_small(ctx, dst, src, bytes);  // (falls back to the copy engine by itself)
  advise_huge_pages(dst, bytes);
  if (bytes >= (256u << 10)) {
    bool done = false;
    const hipError_t e = tpg_download_pinned(ctx, (uint8_t*)dst, (const uint8_t*)src, bytes, &done);
    if (e != hipSuccess || done) return e;
  }
  if (bytes >= XFER_BIG) {  // make the pages present (the buffer is about to be overwritten anyway)
    volatile uint8_t* d = (volatile uint8_t*)dst;
    xfer_parallel(bytes, [=](int, size_t lo, size_t hi) { for (size_t o = lo; o < hi; o += 4096) d[o] = 0; });
  }
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
  return e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
}

// device memory (height contiguous pieces of `width` bytes) -> the same pieces `dpitch` bytes apart in host memory the caller
// owns: rows [j0, j0 + ml) of an m-row column-major matrix that is filled block of loci by block of loci (stream.hip,
// comm.hip).  Through the pinned staging like any large download; hipMemcpy2D otherwise.
hipError_t tpg_download_rows(tpg_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t width, size_t height) {
  if (width == 0 || height == 0) return hipSuccess;
  if (height == 1 || dpitch == width) return tpg_download(ctx, dst, src, width * height);
  advise_huge_pages(dst, (height - 1) * dpitch + width);
  if (width * height >= (256u << 10)) {
    bool done = false;
    const hipError_t e = tpg_download_pinned(ctx, (uint8_t*)dst, (const uint8_t*)src, width * height, &done, width, dpitch);
    if (e != hipSuccess || done) return e;
  }
  hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, width, width, height, hipMemcpyDeviceToHost, ctx->stream);
  return e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
}

// host memory -> device memory.  A large source (an FBM file mapping an R session has not read yet) is moved in chunks:
// while one chunk is on its way (one hipMemcpy), the team of threads faults in the next one, so that the page-fault time
// (14 - 30 ms for the 1.2 million pages of a 5 GB mapping) hides behind the copy instead of preceding it.
// What the box gives (tools/upload_pipeline_probe.hip, 5 GB, page cache warm): the copy out of a touched file mapping runs
// at 39 - 46 GB/s (55 out of anonymous memory); pread() by 16 - 32 threads into ordinary staging buffers beside the DMA
// 33 - 38 GB/s; into a ring of pinned buffers 41 - 45 GB/s after 20 - 40 ms of pinning -- nothing beats the mapping.
hipError_t tpg_upload(tpg_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes < XFER_BIG) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    return e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
  }
  const volatile uint8_t* sp = (const volatile uint8_t*)src;
  auto touch = [sp](size_t a, size_t b) {
    xfer_parallel(b - a, [=](int, size_t lo, size_t hi) { uint8_t acc = 0; for (size_t o = a + lo; o < a + hi; o += 4096) acc ^= sp[o]; (void)acc; });
  };
  const size_t CH = 512u << 20;
  touch(0, std::min(CH, bytes));
  for (size_t a = 0; a < bytes; a += CH) {
    const size_t b = std::min(bytes, a + CH), c = std::min(bytes, b + CH);
    std::thread ahead;
    if (b < bytes) ahead = std::thread([=]() { touch(b, c); });
    hipError_t e = hipMemcpyAsync((uint8_t*)dst + a, (const uint8_t*)src + a, b - a, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (ahead.joinable()) ahead.join();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Nibbles -> bytes: dst[2 i] = src[i] & 15, dst[2 i + 1] = src[i] >> 4 (the other half of host_nibpack.h).  Sixteen packed
// bytes in, thirty-two bytes out per thread and step, coalesced both ways.
__global__ __launch_bounds__(256) void tpg_nib_expand_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 w = src[i];
    const uint32_t in[4] = {w.x, w.y, w.z, w.w};
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t lo = in[k] & 0x0F0F0F0Fu, hi = (in[k] >> 4) & 0x0F0F0F0Fu;  // genotypes 0, 2, 4, 6 / 1, 3, 5, 7 of the dword
      o[2 * k] = __builtin_amdgcn_perm(hi, lo, 0x05010400u);
      o[2 * k + 1] = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
    }
    dst[2 * i] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[2 * i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
  }
}

// The same host -> device copy with the bytes PACKED to nibbles on the way (host_nibpack.h): the team of threads that
// tpg_upload uses to fault the next chunk in packs it instead -- into one of two pinned buffers the context keeps -- while
// the chunk before it is on its way and being expanded; half the bytes cross PCIe, out of pinned memory.  A chunk that
// holds a byte >= 16 goes as it is.  TPG_UPLOAD_PACKED=0 switches it off.
static constexpr size_t NIB_CHUNK = 256u << 20;  // input bytes per chunk: 128 MiB packed
// Pinned staging (two halves of NIB_CHUNK / 2) belongs to the PROCESS, not to a context: pinning 256 MiB costs 35 - 40 ms,
// which a context that lives for one call (the uploader thread of a pipeline, the device threads of tpg_multi_*) would pay
// every time.  A buffer is taken for the length of one upload and handed back; concurrent uploads get one each.
static std::mutex g_nib_mu;
static std::vector<uint8_t*> g_nib_free;
static uint8_t* nib_stage_acquire(bool may_pin) {
  {
    std::lock_guard<std::mutex> lk(g_nib_mu);
    if (!g_nib_free.empty()) { uint8_t* p = g_nib_free.back(); g_nib_free.pop_back(); return p; }
  }
  if (!may_pin) return nullptr;
  uint8_t* p = nullptr;
  if (hipHostMalloc((void**)&p, NIB_CHUNK, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
// at most two buffers stay pinned between uploads (the main context's and a pipeline's uploader thread's); a burst of
// concurrent uploads hands the others back to the system (the two are the process's, like the HIP runtime's own pools: freeing
// them from a static destructor would race the runtime's teardown)
static size_t g_nib_keep = 3;  // (a streamed run has an uploader, a downloader and its main thread; tpg_multi one per device: tpg_stage_keep)
void tpg_stage_keep(int buffers) {
  std::lock_guard<std::mutex> lk(g_nib_mu);
  if ((size_t)buffers > g_nib_keep) g_nib_keep = std::min<size_t>((size_t)buffers, 16);
}
static void nib_stage_release(uint8_t* p) {
  {
    std::lock_guard<std::mutex> lk(g_nib_mu);
    static const size_t env_keep = getenv("TPG_PINNED_KEEP") ? (size_t)atoi(getenv("TPG_PINNED_KEEP")) : 0;
    if (g_nib_free.size() < std::max(g_nib_keep, env_keep)) { g_nib_free.push_back(p); return; }
  }
  (void)hipHostFree(p);  // (device-synchronising: only a burst beyond what the process keeps gets here)
}

static hipError_t tpg_upload_packed(tpg_ctx* ctx, uint8_t* dst, const uint8_t* src, size_t bytes) {
  uint8_t* const pinned = nib_stage_acquire();
  if (!pinned) return tpg_upload(ctx, dst, src, bytes);  // no pinned memory to be had: the plain copy
  struct Back { uint8_t* p; ~Back() { nib_stage_release(p); } } back{pinned};
  uint8_t* d_stage = nullptr;
  // two halves, like the pinned buffer; a payload of one chunk needs its own packed size only
  hipError_t e = tpg_pmalloc((void**)&d_stage, bytes <= NIB_CHUNK ? std::max<size_t>(bytes / 2, 16) : NIB_CHUNK);
  if (e != hipSuccess) { (void)hipGetLastError(); return tpg_upload(ctx, dst, src, bytes); }  // HBM is that full: the plain copy
  hipEvent_t ev[2] = {nullptr, nullptr};
  for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
  bool used[2] = {false, false};
  auto pack_chunk = [&](size_t a, size_t b, uint8_t* out) -> uint8_t {  // bytes [a, b) of src, b - a even
    const size_t n = b - a, pages = (n + 4095) / 4096;
    const int nth = (int)std::min<size_t>(XFER_THREADS, std::max<size_t>(1, pages / 256));
    std::vector<uint8_t> seen((size_t)nth, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < nth; t++) {
      const size_t lo = std::min(n, pages * (size_t)t / (size_t)nth * 4096), hi = std::min(n, pages * ((size_t)t + 1) / (size_t)nth * 4096);
      if (lo < hi) th.emplace_back([=, &seen]() { seen[(size_t)t] = tpg_nibpack(src + a + lo, out + lo / 2, hi - lo); });
    }
    for (auto& t : th) t.join();
    uint8_t r = 0;
    for (uint8_t x : seen) r |= x;
    return r;
  };
  const size_t even = bytes & ~(size_t)31;  // whole 32-byte outputs of the expansion kernel; the tail goes as it is
  int k = 0;
  for (size_t a = 0; a < even && e == hipSuccess; a += NIB_CHUNK, k ^= 1) {
    const size_t b = std::min(even, a + NIB_CHUNK);
    uint8_t* pin = pinned + (size_t)k * (NIB_CHUNK / 2);
    uint8_t* stg = d_stage + (size_t)k * (NIB_CHUNK / 2);
    if (used[k]) e = hipEventSynchronize(ev[k]);  // the copy that last read this half of the pinned buffer is done
    if (e != hipSuccess) break;
    const uint8_t seen = pack_chunk(a, b, pin);   // ... while the other half is on its way
    if (seen < 16) {
      e = hipMemcpyAsync(stg, pin, (b - a) / 2, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipEventRecord(ev[k], ctx->stream);
      used[k] = true;
      if (e == hipSuccess) {
        const int64_t n16 = (int64_t)((b - a) / 32);
        const unsigned grid = (unsigned)std::min<int64_t>((n16 + 255) / 256, (int64_t)ctx->num_cu * 16);
        hipLaunchKernelGGL(tpg_nib_expand_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)stg, (uint4*)(dst + a), n16);
        e = hipGetLastError();
      }
    } else {  // bytes that do not fit a nibble: this chunk as it is
      e = hipMemcpyAsync(dst + a, src + a, b - a, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
  }
  if (e == hipSuccess && even < bytes) e = hipMemcpyAsync(dst + even, src + even, bytes - even, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  else (void)hipStreamSynchronize(ctx->stream);
  for (int q = 0; q < 2; q++)
    if (ev[q]) (void)hipEventDestroy(ev[q]);
  tpg_pfree(d_stage);
  return e;
}

// FBM bytes (nrow x ncols, column-major) -> a .bed payload in HBM (ceil(nrow / 4) bytes per column), packed to 2 bits per
// genotype on the host by the upload team (host_bedpack.h) through lut16 (byte < 16 -> .bed code): for a streamed run that
// needs ONE code table (stream.hip) a quarter of the store's bytes cross PCIe.  Same pipeline as tpg_upload_packed: the team
// packs chunk c + 1 into one half of the pinned buffer while chunk c leaves the other.  *ok = false (nothing usable in dst)
// when there is no pinned staging to be had or a byte >= 16 turned up: the caller sends the block as bytes.
hipError_t tpg_upload_bedpacked(tpg_ctx* ctx, uint8_t* dst, const uint8_t* src, int64_t nrow, int64_t ncols, const uint8_t* lut16, bool* ok) {
  *ok = false;
  const size_t bpl = (size_t)((nrow + 3) / 4), half = NIB_CHUNK / 2;
  if ((size_t)nrow * (size_t)ncols < XFER_BIG) {
    // a small block (a tight budget, a short panel): packed into ordinary memory, one copy -- pinning the 256-MiB staging
    // buffer for it would cost more than the whole transfer
    std::vector<uint8_t> tmp(bpl * (size_t)ncols + 32);
    uint8_t seen = 0;
    if (nrow % 4 == 0) seen = tpg_bedpack(src, tmp.data(), (size_t)nrow * (size_t)ncols, lut16);
    else for (int64_t c = 0; c < ncols; c++) seen |= tpg_bedpack(src + (size_t)c * (size_t)nrow, tmp.data() + (size_t)c * bpl, (size_t)nrow, lut16);
    if (seen >= 16) return hipSuccess;
    hipError_t e = hipMemcpyAsync(dst, tmp.data(), bpl * (size_t)ncols, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    *ok = e == hipSuccess;
    return e;
  }
  uint8_t* const pinned = nib_stage_acquire();
  if (!pinned) return hipSuccess;
  struct Back { uint8_t* p; ~Back() { nib_stage_release(p); } } back{pinned};
  if (bpl > half) return hipSuccess;
  // columns per chunk.  (A block that fits ONE chunk -- the 134-MB blocks of an R driver loop -- is NOT cut further so that its
  // copy could run beside its packing: four chunks measured 2.3 ms against 1.9, a team start and join per chunk.)
  const int64_t cpc = std::max<int64_t>(1, (int64_t)(half / bpl));
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
  bool used[2] = {false, false};
  bool clean = true;
  int k = 0;
  for (int64_t c0 = 0; c0 < ncols && e == hipSuccess && clean; c0 += cpc, k ^= 1) {
    const int64_t c1 = std::min(ncols, c0 + cpc), nc = c1 - c0;
    uint8_t* pin = pinned + (size_t)k * half;
    if (used[k]) e = hipEventSynchronize(ev[k]);  // the copy that last read this half of the pinned buffer is done
    if (e != hipSuccess) break;
    const uint8_t* in = src + (size_t)c0 * (size_t)nrow;
    const size_t nbytes = (size_t)nc * (size_t)nrow;
    const int nth = (int)std::min<size_t>(XFER_THREADS, std::max<size_t>(1, nbytes >> 20));
    std::vector<uint8_t> seen((size_t)nth, 0);
    auto work = [&](int t) {
      if (nrow % 4 == 0) {  // the run of columns is contiguous in both layouts: stripes on 128-byte boundaries
        const size_t units = nbytes / 128, lo = units * (size_t)t / (size_t)nth * 128, hi = t + 1 == nth ? nbytes : units * ((size_t)t + 1) / (size_t)nth * 128;
        seen[(size_t)t] = tpg_bedpack(in + lo, pin + lo / 4, hi - lo, lut16);
      } else {
        uint8_t s = 0;
        for (int64_t c = nc * t / nth; c < nc * (t + 1) / nth; c++) s |= tpg_bedpack(in + (size_t)c * (size_t)nrow, pin + (size_t)c * bpl, (size_t)nrow, lut16);
        seen[(size_t)t] = s;
      }
    };
    if (nth == 1) {
      work(0);
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < nth; t++) th.emplace_back(work, t);
      for (auto& t : th) t.join();
    }
    for (uint8_t x : seen)
      if (x >= 16) clean = false;
    if (!clean) break;
    e = hipMemcpyAsync(dst + (size_t)c0 * bpl, pin, (size_t)nc * bpl, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ev[k], ctx->stream);
    used[k] = true;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  else (void)hipStreamSynchronize(ctx->stream);
  for (int q = 0; q < 2; q++)
    if (ev[q]) (void)hipEventDestroy(ev[q]);
  *ok = e == hipSuccess && clean;
  return e;
}

// The same pipeline WITHOUT the packing, for bytes that are already dense (a PLINK .bed payload: 2 bits per genotype): the
// team copies chunk c + 1 of the source (a file mapping: 39 - 46 GB/s when the DMA reads it directly, page faults included)
// into one half of the pinned buffer while chunk c leaves the other half at the rate of pinned memory (55 - 57 GB/s).
// TPG_UPLOAD_PINNED=0: the plain chunked copy (A/B).
static hipError_t tpg_upload_pinned(tpg_ctx* ctx, uint8_t* dst, const uint8_t* src, size_t bytes) {
  static const bool off = getenv("TPG_UPLOAD_PINNED") && atoi(getenv("TPG_UPLOAD_PINNED")) == 0;
  uint8_t* const pinned = off ? nullptr : nib_stage_acquire();
  if (!pinned) return tpg_upload(ctx, dst, src, bytes);
  struct Back { uint8_t* p; ~Back() { nib_stage_release(p); } } back{pinned};
  const size_t H = NIB_CHUNK / 4;  // 64 MiB pieces, four in the buffer: two copies may be in flight while two are being filled
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int k = 0; k < 4 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
  bool used[4] = {false, false, false, false};
  int k = 0;
  for (size_t a = 0; a < bytes && e == hipSuccess; a += H, k = (k + 1) & 3) {
    const size_t b = std::min(bytes, a + H);
    uint8_t* pin = pinned + (size_t)k * H;
    if (used[k]) e = hipEventSynchronize(ev[k]);  // the copy that last read this piece of the pinned buffer is done
    if (e != hipSuccess) break;
    xfer_parallel(b - a, [=](int, size_t lo, size_t hi) { memcpy(pin + lo, src + a + lo, hi - lo); });
    e = hipMemcpyAsync(dst + a, pin, b - a, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ev[k], ctx->stream);
    used[k] = true;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  else (void)hipStreamSynchronize(ctx->stream);
  for (int q = 0; q < 4; q++)
    if (ev[q]) (void)hipEventDestroy(ev[q]);
  return e;
}

// bulk FBM bytes: packed on the way when that can pay (large, 16-byte aligned destination) and is not switched off
static hipError_t tpg_upload_fbm_bytes(tpg_ctx* ctx, uint8_t* dst, const uint8_t* src, size_t bytes) {
  static const bool off = getenv("TPG_UPLOAD_PACKED") && atoi(getenv("TPG_UPLOAD_PACKED")) == 0;
  if (!off && bytes >= XFER_BIG && (((uintptr_t)dst) & 15) == 0) {
    ProfScope ps(ctx, "upload_packed");
    return tpg_upload_packed(ctx, dst, src, bytes);
  }
  return tpg_upload(ctx, dst, src, bytes);
}

extern "C" int tpg_fbm_from_host(tpg_ctx* ctx, const uint8_t* bytes, int64_t nrow, int64_t ncol, tpg_fbm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && bytes && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(nrow > 0 && ncol > 0, TPG_EINVAL, "empty FBM (%lld x %lld)", (long long)nrow, (long long)ncol);
  TPG_HIP(hipSetDevice(ctx->device));
  tpg_fbm* f = new tpg_fbm{ctx, nullptr, nrow, ncol};
  size_t sz = (size_t)nrow * (size_t)ncol;
  // the blocks of a driver loop (131 MB at 5 000 x 26 843) come from the context's pool: hipMalloc / hipFree cost
  // milliseconds each and synchronise the device; whole panels keep their own allocation
  f->pooled = sz <= ((size_t)1 << 30);
  hipError_t e = f->pooled ? tpg_pmalloc((void**)&f->d_bytes, sz) : hipMalloc((void**)&f->d_bytes, sz);
  if (e != hipSuccess) { delete f; tpg_set_error("device allocation of %zu bytes failed: %s", sz, hipGetErrorString(e)); return TPG_EHIP; }
  e = tpg_upload_fbm_bytes(ctx, f->d_bytes, bytes, sz);
  if (e != hipSuccess) { tpg_fbm_free(f); tpg_set_error("FBM upload failed: %s", hipGetErrorString(e)); return TPG_EHIP; }
  *out = f;
  return TPG_OK;
}

// Host FBM bytes that will be read through ONE code table (code256; NULL = raw bytes): when every entry of the table below 16
// is a code and no byte >= 16 turns up, they go up as 2 bits per genotype (host_bedpack.h: a quarter of the bytes over PCIe,
// half of the nibble pack's) into a store in the layout of a .bed payload, and *view_table is the table to make the views
// through (raw where the caller's was raw -- a lone pairwise view then gets its FP4 layout from the pack --, else 0 / 1 / 2 /
// NA); otherwise the byte store of tpg_fbm_from_host and *view_table = code256.  TPG_UPLOAD_BEDPACK=0: always bytes (A/B).
static void make_lut(const double* code256, uint8_t* lut);
int tpg_fbm_from_host_for_table(tpg_ctx* ctx, const uint8_t* bytes, int64_t nrow, int64_t ncol, const double* code256, tpg_fbm** out,
                                const double** view_table) {
  *view_table = code256;
  const char* sw = getenv("TPG_UPLOAD_BEDPACK");
  uint8_t lut[256], l16[16];
  make_lut(code256, lut);
  static const uint8_t bedcode[4] = {3, 2, 0, 1};  // code 0, 1, 2, missing -> .bed 11, 10, 00, 01 = bigsnpr's bytes 0, 1, 2, 3
  bool can = !(sw && atoi(sw) == 0) && nrow > 0 && ncol > 0 && (size_t)nrow * (size_t)ncol >= (64u << 10);
  for (int b = 0; b < 16 && can; b++) {
    if (lut[b] > 3) can = false;
    else l16[b] = bedcode[lut[b]];
  }
  if (can) {
    TPG_HIP(hipSetDevice(ctx->device));
    tpg_fbm* f = new tpg_fbm{ctx, nullptr, nrow, ncol};
    f->bed_bpl = (nrow + 3) / 4;
    const size_t sz = (size_t)f->bed_bpl * (size_t)ncol;
    f->pooled = true;
    hipError_t e = tpg_pmalloc((void**)&f->d_bytes, sz);
    if (e != hipSuccess) { (void)hipGetLastError(); delete f; return tpg_fbm_from_host(ctx, bytes, nrow, ncol, out); }
    bool ok = false;
    {
      ProfScope ps(ctx, "upload_bedpacked");
      e = tpg_upload_bedpacked(ctx, f->d_bytes, bytes, nrow, ncol, l16, &ok);
    }
    if (e != hipSuccess) { tpg_fbm_free(f); tpg_set_error("FBM upload failed: %s", hipGetErrorString(e)); return TPG_EHIP; }
    if (ok) {
      static double code012[256];
      static const bool init = [] { for (int b = 0; b < 256; b++) code012[b] = b < 3 ? (double)b : __builtin_nan(""); return true; }();
      (void)init;
      *view_table = code256 ? code012 : nullptr;
      *out = f;
      return TPG_OK;
    }
    tpg_fbm_free(f);  // a byte >= 16 somewhere: as bytes
  }
  return tpg_fbm_from_host(ctx, bytes, nrow, ncol, out);
}

// a view straight from host bytes: upload (for this one table: tpg_fbm_from_host_for_table), pack, release the store
extern "C" int tpg_view_create_from_host(tpg_ctx* ctx, const uint8_t* bytes, int64_t nrow, int64_t ncol, const int32_t* rowInd1, int64_t n,
                                         const int32_t* colInd1, int64_t m, const double* code256, tpg_view** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && bytes && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(nrow > 0 && ncol > 0, TPG_EINVAL, "empty FBM (%lld x %lld)", (long long)nrow, (long long)ncol);
  tpg_fbm* f = nullptr;
  const double* table = code256;
  TPG_TRY(tpg_fbm_from_host_for_table(ctx, bytes, nrow, ncol, code256, &f, &table));
  const int rc = tpg_view_create(ctx, f, rowInd1, n, colInd1, m, table, out);
  tpg_fbm_free(f);  // stream-ordered: the pack kernel has run (a view creation ends with a host round trip)
  return rc;
}

// An FBM whose bytes arrive block of columns by block of columns (the reference's own block loop, R/snp_ibs.R:59-82):
// tpg_fbm_alloc reserves the HBM, tpg_fbm_upload_cols fills columns [col0, col0 + ncols) from host memory.  Called
// from a second host thread with a context of its own (its own stream) the upload of block b + 1 runs beside the
// pack / accumulate kernels of block b, which only read columns already uploaded (tpg_view_create with that colInd).
extern "C" int tpg_fbm_alloc(tpg_ctx* ctx, int64_t nrow, int64_t ncol, tpg_fbm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(nrow > 0 && ncol > 0, TPG_EINVAL, "empty FBM (%lld x %lld)", (long long)nrow, (long long)ncol);
  tpg_fbm* f = new tpg_fbm{ctx, nullptr, nrow, ncol};
  hipError_t e = hipMalloc((void**)&f->d_bytes, (size_t)nrow * (size_t)ncol);
  if (e != hipSuccess) { delete f; tpg_set_error("hipMalloc failed: %s", hipGetErrorString(e)); return TPG_EHIP; }
  *out = f;
  return TPG_OK;
}

extern "C" int tpg_fbm_upload_cols(tpg_ctx* ctx, tpg_fbm* fbm, const uint8_t* host_cols, int64_t col0, int64_t ncols) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && fbm && host_cols, TPG_EINVAL, "null argument");
  TPG_REQUIRE(fbm->bed_bpl == 0, TPG_EUNSUPPORTED, "a .bed store is uploaded as a whole");
  TPG_REQUIRE(col0 >= 0 && ncols >= 0 && col0 + ncols <= fbm->ncol, TPG_EINVAL, "columns [%lld, %lld) outside the FBM",
              (long long)col0, (long long)(col0 + ncols));
  TPG_HIP(tpg_upload_fbm_bytes(ctx, fbm->d_bytes + (size_t)col0 * (size_t)fbm->nrow, host_cols, (size_t)ncols * (size_t)fbm->nrow));
  return TPG_OK;
}

extern "C" int tpg_fbm_open_bk(tpg_ctx* ctx, const char* path, int64_t nrow, int64_t ncol, tpg_fbm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && path && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(nrow > 0 && ncol > 0, TPG_EINVAL, "empty FBM (%lld x %lld)", (long long)nrow, (long long)ncol);
  int fd = open(path, O_RDONLY);
  TPG_REQUIRE(fd >= 0, TPG_EINVAL, "cannot open backing file %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0 || (int64_t)st.st_size < nrow * ncol) {
    close(fd);
    tpg_set_error("backing file %s is smaller than %lld x %lld bytes", path, (long long)nrow, (long long)ncol);
    return TPG_EINVAL;
  }
  const size_t sz = (size_t)nrow * (size_t)ncol;
  void* p = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  TPG_REQUIRE(p != MAP_FAILED, TPG_EINVAL, "mmap of %s failed", path);
  int rc = tpg_fbm_from_host(ctx, (const uint8_t*)p, nrow, ncol, out);  // parallel page touch + one copy (tpg_upload)
  munmap(p, sz);
  return rc;
}

extern "C" int tpg_fbm_synth(tpg_ctx* ctx, uint64_t seed, int64_t nrow, int64_t ncol, int64_t j0, int npop,
                             uint32_t miss_thresh, int imputed_bytes, tpg_fbm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(nrow > 0 && ncol > 0 && npop > 0 && npop <= 1024, TPG_EINVAL, "bad synth shape");
  TPG_HIP(hipSetDevice(ctx->device));
  tpg_fbm* f = new tpg_fbm{ctx, nullptr, nrow, ncol};
  size_t sz = (size_t)nrow * (size_t)ncol;
  hipError_t e = hipMalloc((void**)&f->d_bytes, sz);
  if (e != hipSuccess) { delete f; tpg_set_error("tpg_pmalloc(%zu) failed: %s", sz, hipGetErrorString(e)); return TPG_EHIP; }
  int rc = tpg_launch_synth(ctx, f->d_bytes, seed, nrow, ncol, j0, npop, miss_thresh, imputed_bytes);
  if (rc != TPG_OK) { (void)hipFree(f->d_bytes); delete f; return rc; }
  *out = f;
  return TPG_OK;
}

extern "C" int tpg_fbm_from_bed_host(tpg_ctx* ctx, const uint8_t* bytes, int64_t n, int64_t m, tpg_fbm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && bytes && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(n > 0 && m > 0, TPG_EINVAL, "empty .bed (%lld x %lld)", (long long)n, (long long)m);
  TPG_HIP(hipSetDevice(ctx->device));
  const int64_t bpl = (n + 3) / 4;
  tpg_fbm* f = new tpg_fbm{ctx, nullptr, n, m};
  f->bed_bpl = bpl;
  const size_t sz = (size_t)bpl * (size_t)m;
  hipError_t e = hipMalloc((void**)&f->d_bytes, sz);
  if (e != hipSuccess) { delete f; tpg_set_error("hipMalloc(%zu) failed: %s", sz, hipGetErrorString(e)); return TPG_EHIP; }
  e = sz >= XFER_BIG ? tpg_upload_pinned(ctx, f->d_bytes, bytes, sz) : tpg_upload(ctx, f->d_bytes, bytes, sz);
  if (e != hipSuccess) { (void)hipFree(f->d_bytes); delete f; tpg_set_error(".bed upload failed: %s", hipGetErrorString(e)); return TPG_EHIP; }
  *out = f;
  return TPG_OK;
}

// A .bed store whose SNPs arrive block by block (the pipeline of tpg_fbm_alloc / tpg_fbm_upload_cols for the 2-bit payload:
// a SNP is ceil(n / 4) contiguous bytes, so a block of SNPs is one contiguous piece of the file behind its 3-byte magic)
extern "C" int tpg_fbm_alloc_bed(tpg_ctx* ctx, int64_t n, int64_t m, tpg_fbm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(n > 0 && m > 0, TPG_EINVAL, "empty .bed (%lld x %lld)", (long long)n, (long long)m);
  tpg_fbm* f = new tpg_fbm{ctx, nullptr, n, m};
  f->bed_bpl = (n + 3) / 4;
  hipError_t e = hipMalloc((void**)&f->d_bytes, (size_t)f->bed_bpl * (size_t)m);
  if (e != hipSuccess) { delete f; tpg_set_error("hipMalloc failed: %s", hipGetErrorString(e)); return TPG_EHIP; }
  *out = f;
  return TPG_OK;
}

extern "C" int tpg_fbm_upload_bed_snps(tpg_ctx* ctx, tpg_fbm* fbm, const uint8_t* host_snps, int64_t snp0, int64_t nsnps) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && fbm && host_snps, TPG_EINVAL, "null argument");
  TPG_REQUIRE(fbm->bed_bpl > 0, TPG_EUNSUPPORTED, "not a .bed store (tpg_fbm_upload_cols fills a byte FBM)");
  TPG_REQUIRE(snp0 >= 0 && nsnps >= 0 && snp0 + nsnps <= fbm->ncol, TPG_EINVAL, "SNPs [%lld, %lld) outside the store",
              (long long)snp0, (long long)(snp0 + nsnps));
  const size_t sz = (size_t)nsnps * (size_t)fbm->bed_bpl;
  uint8_t* dst = fbm->d_bytes + (size_t)snp0 * (size_t)fbm->bed_bpl;
  TPG_HIP(sz >= XFER_BIG ? tpg_upload_pinned(ctx, dst, host_snps, sz) : tpg_upload(ctx, dst, host_snps, sz));
  return TPG_OK;
}

extern "C" int tpg_fbm_open_bed(tpg_ctx* ctx, const char* path, int64_t n, int64_t m, tpg_fbm** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && path && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(n > 0 && m > 0, TPG_EINVAL, "empty .bed (%lld x %lld)", (long long)n, (long long)m);
  int fd = open(path, O_RDONLY);
  TPG_REQUIRE(fd >= 0, TPG_EINVAL, "cannot open %s", path);
  const size_t sz = 3 + (size_t)((n + 3) / 4) * (size_t)m;
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < sz) {
    close(fd);
    tpg_set_error("%s is smaller than a %lld x %lld .bed", path, (long long)n, (long long)m);
    return TPG_EINVAL;
  }
  void* p = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  TPG_REQUIRE(p != MAP_FAILED, TPG_EINVAL, "mmap of %s failed", path);
  const uint8_t* b = (const uint8_t*)p;
  int rc;
  if (b[0] != 0x6C || b[1] != 0x1B || b[2] != 0x01) {
    tpg_set_error("%s is not a SNP-major PLINK .bed (magic %02x %02x %02x)", path, b[0], b[1], b[2]);
    rc = TPG_EINVAL;
  } else {
    rc = tpg_fbm_from_bed_host(ctx, b + 3, n, m, out);
  }
  munmap(p, sz);
  return rc;
}

extern "C" int tpg_fbm_to_host(tpg_ctx* ctx, const tpg_fbm* fbm, uint8_t* bytes) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && fbm && bytes, TPG_EINVAL, "null argument");
  TPG_REQUIRE(fbm->bed_bpl == 0, TPG_EUNSUPPORTED, "a .bed store has no FBM bytes; unpack a view instead");
  TPG_HIP(hipMemcpyAsync(bytes, fbm->d_bytes, (size_t)fbm->nrow * (size_t)fbm->ncol, hipMemcpyDeviceToHost, ctx->stream));
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  return TPG_OK;
}

extern "C" void tpg_fbm_free(tpg_fbm* fbm) {
  if (!fbm) return;
  if (fbm->d_bytes) {
    if (fbm->pooled) tpg_pfree(fbm->d_bytes);  // stream-ordered: behind the kernels that read it on its context's stream
    else (void)hipFree(fbm->d_bytes);
  }
  delete fbm;
}

// ---------------------------------------------------------------------------
// byte -> 2-bit code table from a code256 (NULL = raw bytes: 0 / 1 / 2 valid, everything else missing)
static void make_lut(const double* code256, uint8_t* lut) {
  for (int b = 0; b < 256; b++) {
    if (!code256) { lut[b] = b < 3 ? (uint8_t)b : 3; continue; }
    double x = code256[b];
    if (!(x > -1)) lut[b] = 3;  // NA (NaN), same test as the reference
    else if (x == 0.0) lut[b] = 0;
    else if (x == 1.0) lut[b] = 1;
    else if (x == 2.0) lut[b] = 2;
    else lut[b] = 0xFF;  // unsupported value: flagged by the pack kernel only if it occurs
  }
}

// one view, or two views of the same (rows, columns) through two code tables packed from one read of the FBM
static int view_create_impl(tpg_ctx* ctx, const tpg_fbm* fbm, const int32_t* rowInd1, int64_t n, const int32_t* colInd1,
                            int64_t m, const double* code256_a, const double* code256_b, bool two, tpg_view** out_a,
                            tpg_view** out_b) {
  TPG_REQUIRE(ctx && fbm && out_a && (!two || out_b), TPG_EINVAL, "null argument");
  if (!rowInd1) n = fbm->nrow;
  if (!colInd1) m = fbm->ncol;
  TPG_REQUIRE(n > 0 && m > 0, TPG_EINVAL, "empty view (%lld x %lld)", (long long)n, (long long)m);
  TPG_REQUIRE(n < (1ll << 24), TPG_EINVAL, "n = %lld too large", (long long)n);
  // index validation on the host (bigstatsr's accessors bounds-check too)
  if (rowInd1)
    for (int64_t i = 0; i < n; i++)
      TPG_REQUIRE(rowInd1[i] >= 1 && rowInd1[i] <= fbm->nrow, TPG_EINVAL, "rowInd[%lld] = %d out of [1,%lld]",
                  (long long)i, rowInd1[i], (long long)fbm->nrow);
  if (colInd1)
    for (int64_t j = 0; j < m; j++)
      TPG_REQUIRE(colInd1[j] >= 1 && colInd1[j] <= fbm->ncol, TPG_EINVAL, "colInd[%lld] = %d out of [1,%lld]",
                  (long long)j, colInd1[j], (long long)fbm->ncol);
  const int nv = two ? 2 : 1;
  uint8_t lut[2 * (256 + 16)];
  memset(lut, 0, sizeof(lut));  // table, then its 16-byte flag area (zero)
  make_lut(code256_a, lut);
  if (two) make_lut(code256_b, lut + 256 + 16);
  tpg_view* v[2] = {nullptr, nullptr};
  int32_t *d_rows = nullptr, *d_cols = nullptr;
  uint8_t* d_lut = nullptr;
  auto fail = [&](int code) {
    if (d_rows) tpg_pfree(d_rows);
    if (d_cols) tpg_pfree(d_cols);
    if (d_lut) tpg_pfree(d_lut);
    tpg_view_free(v[0]);
    tpg_view_free(v[1]);
    return code;
  };
#define VHIP(call)                                                                       \
  do {                                                                                   \
    hipError_t _e = (call);                                                              \
    if (_e != hipSuccess) {                                                              \
      tpg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      return fail(TPG_EHIP);                                                             \
    }                                                                                    \
  } while (0)
  for (int k = 0; k < nv; k++) {
    v[k] = new tpg_view{ctx, n, m, ceil_div(n, 128), ceil_div(m, 128), nullptr, nullptr, 0};
    v[k]->bytes_each = (size_t)v[k]->Q * (size_t)v[k]->KG * 4096;
    // A pair is "the raw view of the pairwise statistics + the imputed view of the PCA": neither reads the 2-bit T
    // layout (the pairwise kernel reads T4, the class Gram its own sorted layout), so the pair is packed as L + T4 and
    // L; whoever does want T gets it from L (tpg_view_need_T).
    // A single RAW-byte view (code256 == NULL) is what the pairwise statistics alone ask for (increment_{ibs,king,as}_counts
    // compare raw bytes, src/snp_ibs.cpp:47-54): it is packed as L + T4 too, which saves a stand-alone snp_ibs / snp_king /
    // pairwise_grm call, and every block of an R driver loop, the T -> T4 pass (0.7 ms at 5 000 x 1 000 000).
    if (!two && code256_a) VHIP(tpg_pmalloc((void**)&v[k]->T, v[k]->bytes_each));
    else if (k == 0) VHIP(tpg_pmalloc((void**)&v[k]->T4, 2 * v[k]->bytes_each));
    VHIP(tpg_pmalloc((void**)&v[k]->L, v[k]->bytes_each));
  }
  VHIP(tpg_pmalloc((void**)&d_lut, sizeof(lut)));
  VHIP(tpg_h2d_async(ctx, d_lut, lut, (size_t)nv * (256 + 16)));
  if (rowInd1) {
    VHIP(tpg_pmalloc((void**)&d_rows, sizeof(int32_t) * (size_t)n));
    VHIP(tpg_h2d_async(ctx, d_rows, rowInd1, sizeof(int32_t) * (size_t)n));
  }
  if (colInd1) {
    VHIP(tpg_pmalloc((void**)&d_cols, sizeof(int32_t) * (size_t)m));
    VHIP(tpg_h2d_async(ctx, d_cols, colInd1, sizeof(int32_t) * (size_t)m));
  }
  int rc = tpg_launch_pack(ctx, fbm, d_rows, d_cols, d_lut, v[0], v[1]);
  if (rc != TPG_OK) return fail(rc);
  // the "a byte that occurs maps to no 2-bit code" flags: the one host round trip of a view creation
  uint8_t back[2 * (256 + 16)];
  VHIP(tpg_fetch_small(ctx, back, d_lut, (size_t)nv * (256 + 16)));
#undef VHIP
  for (int k = 0; k < nv; k++) {
    int32_t bad;
    memcpy(&bad, back + k * (256 + 16) + 256, sizeof(bad));
    if (bad) {
      tpg_set_error("code256 maps an occurring FBM byte to a value outside {0,1,2,NA}; the 2-bit device path cannot represent it");
      return fail(TPG_EUNSUPPORTED);
    }
  }
  if (d_rows) tpg_pfree(d_rows);
  if (d_cols) tpg_pfree(d_cols);
  tpg_pfree(d_lut);
  *out_a = v[0];
  if (two) *out_b = v[1];
  return TPG_OK;
}

extern "C" int tpg_view_create(tpg_ctx* ctx, const tpg_fbm* fbm, const int32_t* rowInd1, int64_t n,
                               const int32_t* colInd1, int64_t m, const double* code256, tpg_view** out) {
  TpgEnter _enter(ctx);
  return view_create_impl(ctx, fbm, rowInd1, n, colInd1, m, code256, nullptr, false, out, nullptr);
}

// Two views of the same (rowInd, colInd) through two code tables from ONE read of the FBM bytes: the raw view of
// the pairwise statistics (code256_a = NULL or CODE_012) and the imputed view of the PCA (CODE_IMPUTE_PRED) that an
// analysis of one gen_tibble needs (R/gt_has_imputed.R:101-106 flips the FBM's code256 between exactly these).
extern "C" int tpg_view_create_pair(tpg_ctx* ctx, const tpg_fbm* fbm, const int32_t* rowInd1, int64_t n,
                                    const int32_t* colInd1, int64_t m, const double* code256_a, const double* code256_b,
                                    tpg_view** out_a, tpg_view** out_b) {
  TpgEnter _enter(ctx);
  return view_create_impl(ctx, fbm, rowInd1, n, colInd1, m, code256_a, code256_b, true, out_a, out_b);
}

extern "C" void tpg_view_free(tpg_view* v) {
  if (!v) return;
  if (v->T) tpg_pfree(v->T);
  if (v->L) tpg_pfree(v->L);
  if (v->T4) tpg_pfree(v->T4);
  if (v->lc_part) tpg_pfree(v->lc_part);
  delete v;
}
extern "C" int64_t tpg_view_n(const tpg_view* v) { return v ? v->n : 0; }
extern "C" int64_t tpg_view_m(const tpg_view* v) { return v ? v->m : 0; }

extern "C" int tpg_view_unpack(tpg_ctx* ctx, const tpg_view* v, uint8_t* codes) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && codes, TPG_EINVAL, "null argument");
  // codes from T, cross-checked against L on the device: both layouts must agree
  OutBuf o;
  TPG_TRY(o.init(codes, (size_t)v->n * (size_t)v->m));
  TPG_TRY(tpg_launch_unpack(ctx, v, o.dev<uint8_t>(), 0));
  uint8_t* d2 = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d2, (size_t)v->n * (size_t)v->m));
  int rc = tpg_launch_unpack(ctx, v, d2, 1);
  std::vector<uint8_t> a((size_t)v->n * (size_t)v->m), b((size_t)v->n * (size_t)v->m);
  if (rc == TPG_OK) {
    hipError_t e = hipMemcpyAsync(a.data(), o.dev<uint8_t>(), a.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b.data(), d2, b.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { tpg_set_error("unpack copy failed: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  }
  tpg_pfree(d2);
  TPG_TRY(rc);
  TPG_REQUIRE(a == b, TPG_EHIP, "internal error: T and L layouts of the view disagree");
  return o.commit(ctx);
}
