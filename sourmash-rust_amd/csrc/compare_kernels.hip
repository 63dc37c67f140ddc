// compare_kernels.hip -- gfx950 kernels for KmerMinHash::compare / intersection_size /
// count_common (reference src/lib.rs:428-436, 470-508) and Leaf::containment
// (reference src/index.rs:146-154).
//
// Single-pass formulation: walk the sorted union of A and B in ascending order for
// at most n = self.num union elements (unbounded when num == 0); `size` = union elements walked,
// `common` = those present in both.  This equals the reference's two merges + two intersections:
// combined = bottom_n(A u B), common = |(A ^ B) ^ combined|, size = |combined|.
//
//   k_compare_wave    one wavefront per ordered pair.  Both sketches are staged in LDS with
//                     coalesced loads; the merged sequence (ties: A first) is cut into 64 equal
//                     diagonals by a merge-path binary search, every lane walks its diagonal, and
//                     a wave prefix sum over the per-lane union counts places the truncation point.
//                     Serves the pairwise C ABI and small / ragged blocks.
//   k_compare_tiled   see below: the N x N matrix kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device.hpp"
#include "kernels.hpp"

namespace smh {
namespace {

struct PairCounts {
  uint32_t uni;  // new union elements in my diagonal
  uint32_t com;  // elements of B that duplicate an element of A in my diagonal
};

// A and B may live in LDS or global memory; the walk is identical.
template <bool Count>
__device__ __forceinline__ PairCounts walk(const uint64_t* A, uint32_t la, const uint64_t* B,
                                           uint32_t lb, uint32_t pa, uint32_t pb, uint32_t steps,
                                           uint64_t u0, uint64_t n) {
  // Count == false: plain census.  Count == true: `com` only counts duplicates whose union rank
  // (running union count, the duplicate itself adds none) is <= n.
  uint32_t uni = 0, com = 0;
  for (uint32_t t = 0; t < steps; t++) {
    bool takeA = (pb >= lb) || (pa < la && A[pa] <= B[pb]);
    if (takeA) { pa++; uni++; }
    else {
      bool dup = pa > 0 && A[pa - 1] == B[pb];
      if (dup) { if (!Count || u0 + uni <= n) com++; }
      else uni++;
      pb++;
    }
  }
  return {uni, com};
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ uint64_t wave_excl_scan64(uint64_t v, int lane) {
  uint64_t incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    uint64_t o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  return incl - v;
}

template <bool InLds>
__global__ __launch_bounds__(64) void k_compare_wave(SketchSet rows, SketchSet cols, uint32_t num,
                                                     const uint32_t* __restrict__ row_nums,
                                                     CompareOut out) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
  const int lane = threadIdx.x;
  const uint64_t npairs = (uint64_t)rows.n * cols.n;
  for (uint64_t pid = blockIdx.x; pid < npairs; pid += gridDim.x) {
    const uint32_t i = (uint32_t)(pid / cols.n), j = (uint32_t)(pid % cols.n);
    const uint64_t ao = rows.offsets[i], bo = cols.offsets[j];
    const uint32_t la = (uint32_t)(rows.offsets[i + 1] - ao), lb = (uint32_t)(cols.offsets[j + 1] - bo);
    const uint64_t n = row_nums ? row_nums[i] : num;
    const uint64_t* A = rows.hashes + ao;
    const uint64_t* B = cols.hashes + bo;
    if (InLds) {
      __syncthreads();
      for (uint32_t t = lane; t < la; t += 64) lds64[t] = A[t];
      for (uint32_t t = lane; t < lb; t += 64) lds64[la + t] = B[t];
      __syncthreads();
      A = lds64;
      B = lds64 + la;
    }
    const uint32_t total = la + lb;
    const uint32_t D = (total + 63) / 64;
    const uint32_t t0 = min((uint32_t)lane * D, total), t1 = min(t0 + D, total);
    // merge path: pa = how many of the first t0 merged elements come from A (ties: A first)
    uint32_t lo = t0 > lb ? t0 - lb : 0, hi = min(t0, la);
    while (lo < hi) {
      uint32_t mid = (lo + hi) >> 1;
      if (A[mid] <= B[t0 - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    const uint32_t pa = lo, pb = t0 - lo;
    PairCounts c = walk<false>(A, la, B, lb, pa, pb, t1 - t0, 0, 0);
    const uint64_t u0 = wave_excl_scan64(c.uni, lane);
    const uint64_t tot_u = wave_sum64(c.uni);
    const uint64_t tot_c = wave_sum64(c.com);
    uint64_t mine = c.com;
    if (n != 0 && tot_u > n) {
      if (u0 + c.uni <= n) mine = c.com;
      else if (u0 <= n) mine = walk<true>(A, la, B, lb, pa, pb, t1 - t0, u0, n).com;
      else mine = 0;
    }
    const uint64_t common = wave_sum64(mine);
    const uint64_t size = (n != 0 && tot_u > n) ? n : tot_u;
    if (lane == 0) {
      if (out.common) out.common[pid] = common;
      if (out.size) out.size[pid] = size;
      if (out.jaccard) out.jaccard[pid] = (double)common / (double)(size > 1 ? size : 1);
      if (out.count_common) out.count_common[pid] = tot_c;
      if (out.containment) out.containment[pid] = (double)tot_c / (double)la;
    }
  }
}

}  // namespace

void launch_compare_block(const SketchSet& rows, const SketchSet& cols, uint32_t num,
                          const uint32_t* row_nums, const CompareOut& out, Device& dev,
                          hipStream_t s, uint32_t max_row_len, uint32_t max_col_len) {
  const uint64_t npairs = (uint64_t)rows.n * cols.n;
  if (npairs == 0) return;
  const size_t need = ((size_t)max_row_len + max_col_len) * sizeof(uint64_t);
  const int grid = (int)(npairs < (uint64_t)dev.cu_count() * 32 ? npairs : (uint64_t)dev.cu_count() * 32);
  dev.prof_begin(s);
  if (need <= 64 * 1024) {
    hipLaunchKernelGGL(k_compare_wave<true>, dim3(grid), dim3(64), need ? need : 16, s, rows, cols, num,
                       row_nums, out);
  } else {
    hipLaunchKernelGGL(k_compare_wave<false>, dim3(grid), dim3(64), 16, s, rows, cols, num, row_nums,
                       out);
  }
  HIP_CHECK(hipGetLastError());
  dev.prof_end("compare_wave", s);
}

}  // namespace smh
