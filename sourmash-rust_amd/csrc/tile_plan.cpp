// tile_plan.cpp -- see tile_plan.hpp
#include "tile_plan.hpp"

#include <algorithm>

namespace smh {

void plan_order(const uint32_t* comp_r, uint32_t nrows, const uint32_t* comp_c, uint32_t ncols, uint32_t max_comp,
                TilePlan* plan) {
  plan->rperm.assign(nrows, 0);
  plan->cperm.assign(ncols, 0);
  std::vector<uint32_t> cnt((size_t)max_comp + 1);
  auto order_by = [&](const uint32_t* comp, uint32_t nsk, std::vector<uint32_t>& perm) {
    std::fill(cnt.begin(), cnt.end(), 0u);
    for (uint32_t i = 0; i < nsk; i++) cnt[(size_t)comp[i] + 1]++;
    for (size_t k = 1; k <= max_comp; k++) cnt[k] += cnt[k - 1];
    for (uint32_t i = 0; i < nsk; i++) perm[cnt[comp[i]]++] = i;
  };
  order_by(comp_r, nrows, plan->rperm);
  order_by(comp_c, ncols, plan->cperm);
}

void plan_tiles(const uint32_t* comp_r, uint32_t nrows, const uint32_t* comp_c, uint32_t ncols, uint32_t tr, uint32_t tc,
                bool symmetric, bool all_tiles_on, TilePlan* plan) {
  const std::vector<uint32_t>& rperm = plan->rperm;
  const std::vector<uint32_t>& cperm = plan->cperm;
  const uint32_t tiles_r = (nrows + tr - 1) / tr, tiles_c = (ncols + tc - 1) / tc;
  plan->all_tiles = (uint64_t)tiles_r * tiles_c;
  std::vector<uint32_t>& list = plan->tiles;
  list.clear();
  // a tile wholly below the diagonal: all its column slots < all its row slots
  auto below = [&](uint32_t ti, uint32_t tj) { return symmetric && (uint64_t)tj * tc + tc - 1 < (uint64_t)ti * tr; };
  if (all_tiles_on || plan->all_tiles > (1ull << 28)) {   // (too many to flag one by one: visit them all)
    for (uint32_t ti = 0; ti < tiles_r; ti++)
      for (uint32_t tj = 0; tj < tiles_c; tj++)
        if (!below(ti, tj)) { list.push_back(ti); list.push_back(tj); }
    return;
  }
  std::vector<uint8_t> flag((size_t)tiles_r * tiles_c, 0);
  // both slot sequences are sorted by component: walk them together
  uint32_t i = 0, j = 0;
  while (i < nrows && j < ncols) {
    const uint32_t cr = comp_r[rperm[i]], cc = comp_c[cperm[j]];
    if (cr < cc) { i++; continue; }
    if (cc < cr) { j++; continue; }
    uint32_t i1 = i, j1 = j;
    while (i1 < nrows && comp_r[rperm[i1]] == cr) i1++;
    while (j1 < ncols && comp_c[cperm[j1]] == cr) j1++;
    for (uint32_t ti = i / tr; ti <= (i1 - 1) / tr; ti++)
      for (uint32_t tj = j / tc; tj <= (j1 - 1) / tc; tj++) flag[(size_t)ti * tiles_c + tj] = 1;
    i = i1; j = j1;
  }
  for (uint32_t ti = 0; ti < tiles_r; ti++)
    for (uint32_t tj = 0; tj < tiles_c; tj++)
      if (flag[(size_t)ti * tiles_c + tj] && !below(ti, tj)) { list.push_back(ti); list.push_back(tj); }
}

}  // namespace smh
