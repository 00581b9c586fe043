// host_fsttiles.h -- pure-host piece of fst.hip: the pair list of pairwise_pop_fst cut into the tiles of populations that
// tpg_fst_wc84_tile_kernel's threads own.  No HIP here: built alone under the sanitizers by tests/test_host_sanitizers.py.
#pragma once
#include <stdint.h>

#include <map>
#include <utility>
#include <vector>

#define FSTW_TR 3
#define FSTW_TC 2
#define FSTW_TASK_INTS (2 + FSTW_TR * FSTW_TC)  // {first row population, first column population, pair index x 6}

// the pairs (0-based populations, row = first, column = second of a pair) cut into tiles of FSTW_TR x FSTW_TC populations
static inline void fst_wc84_tiles(const std::vector<int32_t>& p0, int P, std::vector<int32_t>& tasks) {
  std::map<std::pair<int, int>, std::vector<size_t>> where;  // tile -> its tasks (more than one if a pair is listed twice)
  tasks.clear();
  for (int pi = 0; pi < P; pi++) {
    const int g1 = p0[(size_t)2 * pi], g2 = p0[(size_t)2 * pi + 1];
    const int rb = g1 / FSTW_TR, cb = g2 / FSTW_TC, slot = (g1 % FSTW_TR) * FSTW_TC + g2 % FSTW_TC;
    auto& list = where[{rb, cb}];
    size_t t = (size_t)-1;
    for (size_t cand : list)
      if (tasks[cand * FSTW_TASK_INTS + 2 + slot] < 0) { t = cand; break; }
    if (t == (size_t)-1) {
      t = tasks.size() / FSTW_TASK_INTS;
      tasks.resize(tasks.size() + FSTW_TASK_INTS, -1);
      tasks[t * FSTW_TASK_INTS] = rb * FSTW_TR;
      tasks[t * FSTW_TASK_INTS + 1] = cb * FSTW_TC;
      list.push_back(t);
    }
    tasks[t * FSTW_TASK_INTS + 2 + slot] = pi;
  }
}

