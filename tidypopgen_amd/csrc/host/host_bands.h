// host_bands.h -- how the super-tile rows of the pairwise slabs are dealt to the ranks of a communicator (pairwise.hip).
// Plain C++ (no HIP): tests/test_host_sanitizers.py builds it with -fsanitize=address,undefined.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#ifndef TPG_PW_TA
#define TPG_PW_TA 3
#endif

// Bands: super-tile rows dealt to `nranks` ranks in contiguous runs of (nearly) equal unit counts (row I holds
// TA (nst - I) units).  -> band boundaries and the padded chunk size (units) every band gets in the buffer.
static inline void pw_bands(int64_t nst, int nranks, std::vector<int32_t>& band, int64_t& chunk) {
  const int64_t total = TPG_PW_TA * nst * (nst + 1) / 2;
  band.assign((size_t)nranks + 1, (int32_t)nst);
  band[0] = 0;
  // boundary r = the row boundary whose cumulative unit count is nearest to r / nranks of the total
  int64_t cum = 0;
  int r = 1;
  for (int64_t I = 0; I < nst && r < nranks; I++) {
    const int64_t before = cum;
    cum += TPG_PW_TA * (nst - I);
    while (r < nranks && cum * nranks >= total * r) {
      const bool cut_before = (total * r - before * nranks) < (cum * nranks - total * r) && (int32_t)I > band[(size_t)r - 1];
      band[(size_t)r] = (int32_t)(cut_before ? I : I + 1);
      r++;
    }
  }
  chunk = 0;
  auto off = [&](int64_t I) { return TPG_PW_TA * (I * nst - (I * (I - 1)) / 2); };
  for (int q = 0; q < nranks; q++) chunk = std::max(chunk, off(band[(size_t)q + 1]) - off(band[(size_t)q]));
  if (chunk < 1) chunk = 1;
}
