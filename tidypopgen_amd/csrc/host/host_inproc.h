// host_inproc.h -- the in-process all-reduce the device threads of tpg_multi_* use where RCCL cannot run (comm.hip).  Plain
// C++ threads (no HIP): tests/test_host_sanitizers.py builds it with -fsanitize=thread and with address,undefined.
#pragma once
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <vector>

// Rehearsal transport between the device threads of ONE process (a device listed twice in tpg_multi_create, which RCCL
// refuses, or TPG_MULTI_HOST_TRANSPORT=1): an all-reduce through host memory -- the last thread to arrive adds the
// ranks' buffers in rank order (the same sums on every run), everybody copies the result.  It lets the tpg_multi_* paths
// (thread teams, phases, status agreement, band and row-range writes) run with several ranks on a one-GPU box; tests only.
struct InprocGroup {
  std::mutex mu;
  std::condition_variable cv;
  int n = 0, arrived = 0, left = 0;
  uint64_t gen = 0;
  std::vector<void*> slot;
  std::vector<int64_t> slot_count;
  std::vector<int> slot_dtype;
  bool bad = false;  // the ranks of the last exchange disagreed on count / type
  std::vector<uint8_t> acc;
};
struct InprocRank { InprocGroup* g; int rank; };

static int inproc_allreduce(void* user, void* buf, int64_t count, int dtype) {
  InprocRank* me = (InprocRank*)user;
  InprocGroup* g = me->g;
  std::unique_lock<std::mutex> lk(g->mu);
  g->cv.wait(lk, [&] { return g->left == 0; });  // the previous exchange has been read by everybody
  const uint64_t my_gen = g->gen;
  g->slot[(size_t)me->rank] = buf;
  g->slot_count[(size_t)me->rank] = count;
  g->slot_dtype[(size_t)me->rank] = dtype;
  if (++g->arrived == g->n) {
    // the ranks must have passed the same count and type: anything else is reported to ALL of them (nothing is summed or
    // copied past the end of a shorter buffer)
    g->bad = false;
    for (int r = 0; r < g->n; r++)
      if (g->slot_count[(size_t)r] != count || g->slot_dtype[(size_t)r] != dtype) g->bad = true;
    if (!g->bad) {
      const size_t es = dtype == 0 ? sizeof(int32_t) : sizeof(double);
      g->acc.assign((size_t)count * es, 0);
      for (int r = 0; r < g->n; r++) {
        if (dtype == 0) { int32_t* a = (int32_t*)g->acc.data(); const int32_t* b = (const int32_t*)g->slot[(size_t)r]; for (int64_t i = 0; i < count; i++) a[i] += b[i]; }
        else { double* a = (double*)g->acc.data(); const double* b = (const double*)g->slot[(size_t)r]; for (int64_t i = 0; i < count; i++) a[i] += b[i]; }
      }
    }
    g->arrived = 0;
    g->left = g->n;
    g->gen++;
    g->cv.notify_all();
  } else {
    g->cv.wait(lk, [&] { return g->gen != my_gen; });
  }
  const bool bad = g->bad;
  if (!bad) memcpy(buf, g->acc.data(), g->acc.size());
  if (--g->left == 0) g->cv.notify_all();
  return bad ? 1 : 0;
}
