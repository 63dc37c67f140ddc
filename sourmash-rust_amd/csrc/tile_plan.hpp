// tile_plan.hpp -- host-side planning of the N x M compare block (no device code):
// which (row tile, column tile) pairs can hold sketches that share a hash, given the connected
// component of every row and column (compare_kernels.hip computes them with a union-find over
// the sorted pooled hashes).  Kept separate so that the CPU test-suite can check it.
#pragma once
#include <cstdint>
#include <vector>

namespace smh {

struct TilePlan {
  std::vector<uint32_t> rperm, cperm;  // slot -> row / column: sketches of one component are adjacent
  std::vector<uint32_t> tiles;         // (row tile, column tile) pairs to launch
  uint64_t all_tiles = 0;              // tiles in the whole block
};

// rperm / cperm: stable counting sort of rows / columns by component id (ids < max_comp).
void plan_order(const uint32_t* comp_r, uint32_t nrows, const uint32_t* comp_c, uint32_t ncols, uint32_t max_comp,
                TilePlan* plan);
// tiles of tr x tc slots that contain a (row, column) pair of one component.  symmetric (rows and
// columns are the same list with one num): tiles wholly below the diagonal are left out, their
// pairs are written as the mirrors of the tiles above.  all_tiles_on: every tile (measurements).
void plan_tiles(const uint32_t* comp_r, uint32_t nrows, const uint32_t* comp_c, uint32_t ncols, uint32_t tr, uint32_t tc,
                bool symmetric, bool all_tiles_on, TilePlan* plan);

}  // namespace smh
