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
//   k_pair_block/grid one ordered pair from plain pointers (the pairwise reference API).
//   k_compare_few     a few sketches against many (LinearIndex::find, scaffold arg-max).
//   k_compare_comp    N x M block, few sharing pairs: one workgroup per (column, rows of its component).
//   k_compare_tiled   the N x M matrix kernel (see below), over the tiles a device-side plan selects;
//                     frequent hashes are set aside and decided by k_fill_disjoint from per-sketch records.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <memory>
#include <mutex>
#include <vector>
#include <cstdlib>

#include "block_sort.hpp"
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

// ---------------------------------------------------------------------------------
// One ordered pair from plain pointers (the pairwise reference API on mirrored sketches): no CSR
// tables to upload, the counts come back in one small struct and the host derives every output.
//   k_pair_block  one workgroup: merge path over all its lanes, workgroup scan of the per-lane
//                 union counts for the truncation point n = self.num.
//   k_pair_grid   no truncation (n == 0, scaled sketches of any size): the merged sequence is cut
//                 over the whole grid, every wave adds its census with two atomics.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_pair_block(const uint64_t* __restrict__ A, uint32_t la,
                                                        const uint64_t* __restrict__ B, uint32_t lb, uint64_t n,
                                                        PairOut* __restrict__ out) {
  constexpr int NW = THREADS / 64;
  __shared__ unsigned long long w_u[NW], w_c[NW], w_m[NW];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t total = la + lb;
  const uint32_t D = (total + THREADS - 1) / THREADS;
  const uint32_t t0 = min((uint32_t)tid * D, total), t1 = min(t0 + D, total);
  uint32_t lo = t0 > lb ? t0 - lb : 0, hi = min(t0, la);
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (A[mid] <= B[t0 - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  const uint32_t pa = lo, pb = t0 - lo;
  const PairCounts c = walk<false>(A, la, B, lb, pa, pb, t1 - t0, 0, 0);
  // workgroup exclusive scan of uni: wave scan + wave totals through LDS
  const uint64_t wave_excl = wave_excl_scan64(c.uni, lane);
  const uint64_t wave_u = wave_sum64(c.uni), wave_c = wave_sum64(c.com);
  if (lane == 0) { w_u[w] = wave_u; w_c[w] = wave_c; }
  __syncthreads();
  uint64_t before = 0, tot_u = 0, tot_c = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    if (i < w) before += w_u[i];
    tot_u += w_u[i]; tot_c += w_c[i];
  }
  const uint64_t u0 = before + wave_excl;
  uint64_t mine = c.com;
  if (n != 0 && tot_u > n) {
    if (u0 + c.uni <= n) mine = c.com;
    else if (u0 <= n) mine = walk<true>(A, la, B, lb, pa, pb, t1 - t0, u0, n).com;
    else mine = 0;
  }
  const uint64_t wave_m = wave_sum64(mine);
  if (lane == 0) w_m[w] = wave_m;
  __syncthreads();
  if (tid == 0) {
    uint64_t common = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) common += w_m[i];
    out->tot_u = tot_u; out->tot_c = tot_c; out->common = common;
  }
}

__global__ __launch_bounds__(256) void k_pair_grid(const uint64_t* __restrict__ A, uint32_t la,
                                                   const uint64_t* __restrict__ B, uint32_t lb,
                                                   PairOut* __restrict__ out) {
  const uint64_t T = (uint64_t)gridDim.x * 256;
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t total = (uint64_t)la + lb;
  const uint64_t D = (total + T - 1) / T;
  const uint64_t t0 = min(g * D, total), t1 = min(t0 + D, total);
  uint32_t lo = t0 > lb ? (uint32_t)(t0 - lb) : 0, hi = (uint32_t)min(t0, (uint64_t)la);
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (A[mid] <= B[t0 - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  const PairCounts c = walk<false>(A, la, B, lb, lo, (uint32_t)(t0 - lo), (uint32_t)(t1 - t0), 0, 0);
  const uint64_t wu = wave_sum64(c.uni), wc = wave_sum64(c.com);
  if ((threadIdx.x & 63) == 0) {
    if (wu) atomicAdd(&out->tot_u, (unsigned long long)wu);
    if (wc) atomicAdd(&out->tot_c, (unsigned long long)wc);
  }
}

// ---------------------------------------------------------------------------------
// k_compare_few: a few sketches against many (LinearIndex::find, the scaffold arg-max, a query
// against a resident index).  blockIdx.y picks the "few" sketch Q, kept in LDS by the workgroup;
// each wave streams whole "many" sketches A from HBM, 64 consecutive elements per step (one
// coalesced 512 B read).  Lane i binary-searches its element in Q: lb = #(Q < a), match = Q[lb]==a.
// Its position in the sorted union is  u = i + lb - (#matches before i)  (ballot prefix), so the
// truncated walk needs no serial merge:  common = #(match && u < n),  count_common = #match,
// |A u Q| = |A| + |Q| - count_common.  The search window's lower end is carried from step to step
// (elements ascend).  Unlike k_compare_wave the LDS footprint is one Q per workgroup, not A + B per
// wave, so occupancy stays high.
struct WavePair { uint32_t cm, cc; bool cut; };
// one wave, one ordered pair (A streamed from memory, Q searched): see k_compare_few
template <bool WantCC>
__device__ __forceinline__ WavePair wave_pair(const uint64_t* __restrict__ A, uint32_t la, const uint64_t* Q, uint32_t lq,
                                              uint32_t n, int lane) {
  uint32_t base = 0, cc = 0, cm = 0;
  bool cut = false;
  for (uint32_t i0 = 0; i0 < la; i0 += 64) {
    const uint32_t i = i0 + lane;
    const bool ok = i < la;
    const uint64_t a = ok ? A[i] : ~0ull;
    uint32_t lo = base, len = lq - base;
    while (len > 0) {
      const uint32_t half = len >> 1, mid = lo + half;
      const bool lt = Q[mid] < a;
      lo = lt ? mid + 1 : lo;
      len = lt ? len - half - 1 : half;
    }
    const bool match = ok && lo < lq && Q[lo] == a;
    const uint64_t mm = __ballot(match);
    const uint32_t u = i + lo - (cc + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull)));
    cm += (uint32_t)__popcll(__ballot(match && u < n));
    cc += (uint32_t)__popcll(mm);
    base = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
    // union rank of the step's last element already past the cut: nothing later can count
    if (!WantCC && (uint32_t)__builtin_amdgcn_readlane((int)u, 63) >= n && i0 + 64 <= la) { cut = true; break; }
  }
  return {cm, cc, cut};
}

template <bool QLds, bool WantCC>
__global__ __launch_bounds__(256) void k_compare_few(SketchSet many, SketchSet few, uint32_t many_is_row, uint32_t num,
                                                     const uint32_t* __restrict__ row_nums, CompareOut out) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t y = blockIdx.y;
  const uint64_t qo = few.offsets[y];
  const uint32_t lq = (uint32_t)(few.offsets[y + 1] - qo);
  const uint64_t* Q = few.hashes + qo;
  if (QLds) {
    for (uint32_t t = tid; t < lq; t += 256) lds64[t] = Q[t];
    __syncthreads();
    Q = lds64;
  }
  for (uint32_t node = blockIdx.x * 4 + w; node < many.n; node += gridDim.x * 4) {
    const uint64_t ao = many.offsets[node];
    const uint32_t la = (uint32_t)(many.offsets[node + 1] - ao);
    uint32_t n = row_nums ? row_nums[many_is_row ? node : y] : num;
    n = n ? n : 0xffffffffu;
    const WavePair r = wave_pair<WantCC>(many.hashes + ao, la, Q, lq, n, lane);
    if (lane == 0) {
      const uint64_t tot_u = (uint64_t)la + lq - r.cc;
      const uint64_t size = (r.cut || tot_u > n) ? n : tot_u;
      const size_t pid = many_is_row ? (size_t)node * few.n + y : (size_t)y * many.n + node;
      if (out.common) out.common[pid] = r.cm;
      if (out.size) out.size[pid] = size;
      if (out.jaccard) out.jaccard[pid] = (double)r.cm / (double)(size > 1 ? size : 1);
      if (WantCC) {
        if (out.count_common) out.count_common[pid] = r.cc;
        if (out.containment) out.containment[pid] = (double)r.cc / (double)(many_is_row ? la : lq);
      }
    }
  }
}

// k_compare_comp: the N x M block when only a modest number of pairs can share a hash (small
// components): one workgroup per (column, its component's rows).  The column sketch sits in LDS,
// each wave takes rows of the same component -- slots [r0, r1) of the row order -- and computes
// the pair exactly like k_compare_few.  No rank encoding is needed on this route, and a pair
// keeps 64 lanes busy instead of one, which is what a launch of a few thousand pairs needs.
// Which pairs of a rows x cols block a launch is responsible for, and where a computed pair is ALSO written.
//   own_mode 0  every pair of the block.
//   own_mode 1  rows and columns are the same sketches in the same slot order (one process, all-vs-all, one num): tiles
//               wholly below the diagonal of slot space are not launched; every pair writes its mirror.
//   own_mode 2  the rows are a block of an all-vs-all matrix that several ranks share (columns = the whole collection,
//               one num): row i owns the pairs (i, j) with (j - i) mod N < N/2 (ties: the smaller index) -- every unordered
//               pair has exactly one owner, every row owns N/2 pairs.  Only owned pairs HAVE to be computed here (the
//               other rank sends the rest, distributed.py); pairs whose column is one of this block's rows also write
//               their mirror, so the diagonal block needs no exchange.
// A pair (i, j) whose column j lies in [mir_lo, mir_hi) -- global indices of the local rows -- also writes (j, i).
struct PairScope {
  uint32_t own_mode;
  uint32_t ntotal;      // N of the collection (own_mode 2)
  uint32_t row_base;    // global index of local row 0
  uint32_t col_base;    // global index of local column 0
  uint32_t mir_lo, mir_hi;
};
__device__ __forceinline__ bool owns_pair(uint32_t i, uint32_t j, uint32_t N) {
  const uint64_t d = j >= i ? (uint64_t)(j - i) : (uint64_t)j + N - i;
  return 2 * d < N || (2 * d == N && i < j);
}

struct CompWork { uint32_t col, r0, r1; };
// The work list and its length are produced on the device (see "device-side plan" below): a
// persistent grid walks it.  rkey[slot] = (component << 32 | row): rows in component order.
template <bool QLds, bool WantCC>
__global__ __launch_bounds__(256) void k_compare_comp(SketchSet rows, SketchSet cols, const CompWork* __restrict__ work,
                                                      const uint32_t* __restrict__ nwork_dev, uint32_t work_cap,
                                                      const uint64_t* __restrict__ rkey, uint32_t num,
                                                      const uint32_t* __restrict__ row_nums, PairScope sc,
                                                      CompareOut out) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t nwork = min(*nwork_dev, work_cap);
  for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const CompWork wk = work[wi];
    const uint32_t col = wk.col;
    const uint64_t qo = cols.offsets[col];
    const uint32_t lq = (uint32_t)(cols.offsets[col + 1] - qo);
    const uint64_t* Q = cols.hashes + qo;
    if (QLds) {
      __syncthreads();                       // the previous item's column is no longer being read
      for (uint32_t t = tid; t < lq; t += 256) lds64[t] = Q[t];
      __syncthreads();
      Q = lds64;
    }
    for (uint32_t slot = wk.r0 + w; slot < wk.r1; slot += 4) {
      const uint32_t row = (uint32_t)rkey[slot];
      const uint32_t gi = sc.row_base + row, gj = sc.col_base + col;
      if (sc.own_mode == 2 && !owns_pair(gi, gj, sc.ntotal)) continue;   // the owner's rank (or this one, as (j, i)) computes it
      const uint64_t ao = rows.offsets[row];
      const uint32_t la = (uint32_t)(rows.offsets[row + 1] - ao);
      uint32_t n = row_nums ? row_nums[row] : num;
      n = n ? n : 0xffffffffu;
      const WavePair r = wave_pair<WantCC>(rows.hashes + ao, la, Q, lq, n, lane);
      if (lane == 0) {
        const uint64_t tot_u = (uint64_t)la + lq - r.cc;
        const uint64_t size = (r.cut || tot_u > n) ? n : tot_u;
        const double jac = (double)r.cm / (double)(size > 1 ? size : 1);
        const size_t pid = (size_t)row * cols.n + col;
        if (out.common) out.common[pid] = r.cm;
        if (out.size) out.size[pid] = size;
        if (out.jaccard) out.jaccard[pid] = jac;
        if (WantCC) {
          if (out.count_common) out.count_common[pid] = r.cc;
          if (out.containment) out.containment[pid] = (double)r.cc / (double)la;
        }
        if (gj >= sc.mir_lo && gj < sc.mir_hi && gi != gj) {   // the column is one of the local rows: also pair (col, row)
          const size_t pid2 = (size_t)(gj - sc.mir_lo) * cols.n + (gi - sc.col_base);
          if (out.common) out.common[pid2] = r.cm;
          if (out.size) out.size[pid2] = size;
          if (out.jaccard) out.jaccard[pid2] = jac;
          if (WantCC) {
            if (out.count_common) out.count_common[pid2] = r.cc;
            if (out.containment) out.containment[pid2] = (double)r.cc / (double)lq;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// k_compare_tiled: the N x M matrix kernel.
//
// Pre-pass (host driver below): all hashes of rows and columns are dictionary-encoded to dense,
// order-preserving u32 ranks (radix sort + run ids), so every comparison below is a 32-bit
// compare and the result is unchanged.  Rank space is cut into R ranges holding equal shares of
// the pooled elements, and part[s][r] = first element of sketch s that is >= bound[r].
//
// One workgroup owns a 64 x 64 tile of pairs: lane = column, each wave walks 16 rows.  Ranges are
// visited in ascending order; for each, the 64 row segments are packed into an LDS pool (a segment
// is at most a few dozen consecutive dwords: distinct banks, equal addresses broadcast) and the 64
// column segments are stored transposed (element e of column j at e*64+j: lane j always hits bank
// j mod 32).  Every lane merges its pair's two segments (sentinel-terminated), carrying the running
// union and common counts across ranges; the union walk stops counting at n = the row's num.
// A tile whose segments do not fit LDS for some range merges that range straight from global memory.
constexpr int kTB = 64;          // columns per tile (= lanes of a wave); rows per tile = WPB * RPW
// Sentinels behind every staged segment, above every rank (a block holds fewer than 2^31 - 2 hashes) and positive as ints.
// A row's differs from a column's, so that when BOTH sides of a pair are exhausted the walk sees A < B: it keeps stepping
// over A's padding without finding a match or moving B (the walks look for the end only every 4 steps).
constexpr uint32_t kSentA = 0x7ffffffeu, kSent = 0x7fffffffu;
// four consecutive ranks read with one load from ANY element position (dword-aligned; the arrays are padded by 16 bytes)
struct __attribute__((aligned(4))) Rank4 { uint32_t x, y, z, w; };
constexpr uint32_t kPad = 4;
using LdsU32 = const __attribute__((address_space(3))) uint32_t*;

// What the device-side plan of a block compare decides (see "device-side plan" below); the kernels
// read it, the host reads it back once, at the end of the call.
// Read-only data written by EARLIER launches, read with an address that is the same for all lanes: through the constant
// address space the compiler uses scalar loads (s_load into scalar registers, the scalar cache) instead of vector loads.
#define SMH_CONSTANT __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const SMH_CONSTANT T* as_constant(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (const SMH_CONSTANT T*)p;
#pragma clang diagnostic pop
}
struct PlanState {
  unsigned long long pairs;      // sum over the components of rows x columns: pairs that CAN share a hash
  unsigned long long ovf_steps;  // tiled: (tile, range) steps that did not fit the LDS stage and merged from global memory
  uint32_t route;                // kRouteComponents or kRouteTiled
  uint32_t skip_tiled;           // 1: the tiled kernels do nothing
  uint32_t skip_comp;            // 1: the per-component pair kernel does nothing
  uint32_t rpw;                  // rows per wave of the tiled instantiation that runs (4, 2 or 1)
  uint32_t ntiles;               // tiles in the list
  uint32_t nwork;                // work items of the per-component kernel
  uint32_t count16;              // tiles that hold sharing pairs at the 16-row geometry
  uint32_t next_tile[8];         // tiled: tiles handed out so far, per XCD stretch of the list
  uint32_t pf;                   // tiled: the software-pipelined kernel walks the tiles (k_compare_tiled_pf), not the plain one
  uint32_t halvings;             // pipelined kernel: stretches whose speculative span did not fit and was rebuilt, halved, with plain loads
  uint32_t pf_after_halving;     // ... tiles in which a table was built from PREFETCHED boundary crossings after such a rebuild
  uint32_t bad_tables;           // a segment that would end before it starts (never, unless a table was built from crossings of the wrong boundary)
};
// Workgroups are dealt to the 8 XCDs round-robin.  A kernel that goes through the SORTED positions and touches arrays in
// collection order (by origin) touches, at any time, one line per sketch -- consecutive elements of a sketch are ~n sorted
// positions apart -- and every such line is touched again by the positions that follow.  Give every XCD ONE contiguous
// stretch of the sorted positions and those lines are completed inside its own L2 instead of travelling to memory in
// pieces from all eight.
__device__ __forceinline__ uint32_t xcd_chunked_block() {
  const uint32_t b = blockIdx.x, per = gridDim.x >> 3;
  return b < per * 8u ? (b & 7u) * per + (b >> 3) : b;
}
// What the owner of one slice of hash space finds in it (see "collection dictionary" below)
struct RangeState {
  uint32_t nruns;                // distinct hashes of the slice (local dense ranks)
  uint32_t nfreq_seen;           // runs longer than the frequency threshold
  uint32_t nfreq;                // frequent hashes set aside (0 when there were more than the slice's share of kMaxFreq)
  uint32_t freq_run[64];         // their run ids, ascending (= ascending hash) once k_freq_finalize has run
  uint32_t overflow;             // the four-pass sort gave up (keys that tie in the sorted bits and could not be put in order by
                                 // k_tie_fix / k_tie_sort): the slice is built again with all eight passes
  uint32_t sort_shift;           // the four-pass sort: its lowest bit (k_key_span)
  uint32_t span_done;            // k_key_span: workgroups that have added their part
  unsigned long long span_or;    // k_key_span: OR of (segment end ^ one reference key): its highest bit is the highest bit that varies
};
constexpr uint32_t kMaxFreq = 64;

// what k_mask_layout decided about the range masks of a collection (see "range masks" further down)
struct MaskInfo {
  uint32_t ok;          // the masks exist (few enough words)
  uint32_t wtot;        // words per sketch
  uint32_t shared;      // shared hashes in all
  uint32_t pad;
};

struct TiledArgs {
  const uint32_t* rrank; const uint64_t* roff; const uint32_t* rpart; uint32_t nrows;
  const uint32_t* crank; const uint64_t* coff; const uint32_t* cpart; uint32_t ncols;
  uint32_t R, num;
  const uint32_t* row_nums;
  const uint32_t* tiles;   // (row tile, column tile) pairs: only tiles that can hold sharing pairs
  uint32_t tiles_cap;
  const uint64_t* rkey;    // row slot -> (component << 32 | row); column slot -> ...: sketches of one component are adjacent
  const uint64_t* ckey;
  PlanState* st;
  uint32_t use_xcd;        // give every XCD a contiguous stretch of the tile list
  PairScope scope;         // pair (i, j) also writes (j, i) when j is one of the local rows
  uint32_t capA, capBt;  // LDS dwords for the row pool / the transposed column tile
  unsigned long long* ovf_steps;   // = &st->ovf_steps
  CompareOut out;
  // range masks (nullptr: none -- the walk starts at the first range): see "range masks" further down.  Word-major /
  // range-major tables over ALL nsk sketches of the collection; a row or column of the block is sketch scope.*_base + index
  const unsigned long long* masks; const uint32_t* partT; const uint32_t* woff; const uint32_t* wn; const MaskInfo* minfo; uint32_t nsk;
};

// One row against the 64 staged columns (lane = column) over one staged stretch of rank space: A = the row's la ranks +
// sentinels, Bl = this lane's column (elements kTB dwords apart, lb of them + sentinels), n = the row's cut (bottom-n of the
// union), ucount / common / cc = the pair's running union size, matches inside the cut, all matches.
template <bool WantCC>
__device__ __forceinline__ void tiled_walk_row(const uint32_t* A, const uint32_t* Bl, uint32_t la, uint32_t lb, uint32_t n,
                                               uint32_t& ucount, uint32_t& common, uint32_t& cc) {
  if (!WantCC && ucount >= n) { ucount += la + lb; return; }  // past the cut: nothing can count
  const uint32_t u0 = ucount;
  // No lane's cut can fall inside this range (even with no match at all the union stays within n), or every lane is
  // past it already (count_common wanted): only the NUMBER of matches of the range matters, and the walk needs no
  // counters at all -- a step advances A, B or both, so A's and B's final positions say how many steps were matches.
  // Addresses advance by a constant per step (folded into the reads' immediate offsets) minus what the compares
  // take back, all in two-operand full-rate instructions; the end (both at a sentinel) is looked for every 4 steps.
  uint32_t sa = (uint32_t)(uintptr_t)A, sb = (uint32_t)(uintptr_t)Bl;   // LDS byte addresses
  const uint32_t sa0 = sa;
  uint32_t av = *(LdsU32)(uintptr_t)sa, bv = *(LdsU32)(uintptr_t)sb;
  uint32_t iters = 0;
  if (__all((u0 + la + lb <= n) || (WantCC && u0 >= n))) {
    while ((av & bv) != kSentA) {
#pragma unroll
      for (uint32_t j = 1; j <= 4; j++) {
        const uint32_t g = (bv - av) >> 31;      // 1: bv < av, only B's element is consumed (ranks are < 2^31)
        const uint32_t l = (av - bv) >> 31;      // 1: av < bv, only A's
        sa -= g << 2;
        sb -= l << 8;
        av = *(LdsU32)(uintptr_t)(sa + 4u * j);
        bv = *(LdsU32)(uintptr_t)(sb + (4u * kTB) * j);
      }
      sa += 16u; sb += 16u * kTB;
      iters += 1;
    }
    // A fell behind the unconditional 4 bytes per step once for every B-only step; B's elements are B-only or matches
    const uint32_t m = lb - ((sa0 + 16u * iters - sa) >> 2);
    ucount = u0 + la + lb - m;
    if (u0 + la + lb <= n) common += m;
    if (WantCC) cc += m;
    return;
  }
  // The cut may fall inside this range for some lane: the same walk with the matches counted while the union is
  // short of n (r = u - n is negative until then), still in two-operand arithmetic on VGPRs only.
  int32_t r = (int32_t)(u0 - n);
  uint32_t cmv = 0, ccv = 0;
  while ((av & bv) != kSentA) {
#pragma unroll
    for (uint32_t j = 1; j <= 4; j++) {
      const uint32_t t1 = bv - av, t2 = av - bv;
      const uint32_t eq = ((t1 | t2) >> 31) ^ 1u;     // neither is smaller
      cmv += eq & ((uint32_t)r >> 31);
      if (WantCC) ccv += eq;
      r += 1;
      sa -= (t1 >> 31) << 2;
      sb -= (t2 >> 31) << 8;
      av = *(LdsU32)(uintptr_t)(sa + 4u * j);
      bv = *(LdsU32)(uintptr_t)(sb + (4u * kTB) * j);
    }
    sa += 16u; sb += 16u * kTB;
  }
  // a lane that did not reach its cut has counted every match; one that did only needs ucount >= n from here on
  ucount = u0 + la + lb - (WantCC ? ccv : cmv);
  common += cmv;
  if (WantCC) cc += ccv;
}

// Wave-wide inclusive prefix sum / maximum in DPP steps (row shifts inside the 16-lane rows, then the last lane of a row
// broadcast to the following rows): VALU speed -- __shfl_up / __shfl_xor go through the LDS crossbar, ~100 cycles a step,
// and the two waves that build a stretch's table are what the other six wait for.
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {      // the maximum, in every lane's return value
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// The next tile of this workgroup (thread 0): a persistent grid pulls tiles off the list (tile times differ: a static split
// left the chip half idle at the end).  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: workgroup b
// first serves stretch b % 8 of the list, so that the tiles an XCD works on share rows and columns, and helps the other
// stretches out once its own is exhausted (steal = stretches given up so far).  0xffffffff: none left.
__device__ __forceinline__ uint32_t tiled_take_tile(PlanState* st, uint32_t ntiles, uint32_t nstretch, uint32_t chunk, uint32_t& steal) {
  while (steal < nstretch) {
    const uint32_t x = ((blockIdx.x & 7u) + steal) % nstretch;
    const uint32_t lo = x * chunk, hi = min(lo + chunk, ntiles);
    if (lo < hi) {
      const uint32_t k = atomicAdd(&st->next_tile[x], 1u);
      if (lo + k < hi) return lo + k;
    }
    steal++;
  }
  return 0xffffffffu;
}

// The same merge for one range that does not fit the LDS stage: A and B straight from global memory (rare)
template <bool WantCC>
__device__ __forceinline__ void tiled_merge_global_row(const uint32_t* __restrict__ A, uint32_t la, const uint32_t* __restrict__ B, uint32_t lb,
                                                       uint32_t n, uint32_t& ucount, uint32_t& common, uint32_t& cc) {
  uint32_t pa = 0, pb = 0, u = ucount, cm = common, c2 = 0;
  while (pa < la && pb < lb) {
    const uint32_t av = A[pa], bv = B[pb];
    const bool eq = av == bv;
    cm += (eq && u < n) ? 1u : 0u;
    if (WantCC) c2 += eq ? 1u : 0u;
    u += 1;
    pa += av <= bv ? 1u : 0u;
    pb += bv <= av ? 1u : 0u;
  }
  u += (la - pa) + (lb - pb);
  ucount = u; common = cm;
  if (WantCC) cc += c2;
}
// What a lane has found for pair (row, col) goes to the outputs -- and to the mirrored entry when the column is one of the
// local rows and there is one num: the walk is symmetric in its two inputs, so this is also pair (col, row); tiles that
// hold no owned pair are not launched.  nq: the row's cut.
template <bool WantCC>
__device__ __forceinline__ void tiled_write_pair(const TiledArgs& a, uint32_t row, uint32_t col, uint32_t nq, uint32_t ucount, uint32_t common,
                                                 uint32_t cc) {
  const size_t pid = (size_t)row * a.ncols + col;
  const uint64_t size = ucount < nq ? ucount : nq;
  const double jac = (double)common / (double)(size > 1 ? size : 1);
  if (a.out.common) a.out.common[pid] = common;
  if (a.out.size) a.out.size[pid] = size;
  if (a.out.jaccard) a.out.jaccard[pid] = jac;
  if (WantCC) {
    if (a.out.count_common) a.out.count_common[pid] = cc;
    if (a.out.containment) a.out.containment[pid] = (double)cc / (double)(a.roff[row + 1] - a.roff[row]);
  }
  const uint32_t gi = a.scope.row_base + row, gj = a.scope.col_base + col;
  if (gj >= a.scope.mir_lo && gj < a.scope.mir_hi && gi != gj) {
    const size_t pid2 = (size_t)(gj - a.scope.mir_lo) * a.ncols + (gi - a.scope.col_base);
    if (a.out.common) a.out.common[pid2] = common;
    if (a.out.size) a.out.size[pid2] = size;
    if (a.out.jaccard) a.out.jaccard[pid2] = jac;
    if (WantCC) {
      if (a.out.count_common) a.out.count_common[pid2] = cc;
      if (a.out.containment) a.out.containment[pid2] = (double)cc / (double)(a.coff[col + 1] - a.coff[col]);
    }
  }
}

// RPW rows per wave, WPB waves per workgroup (they share the staged column tile), MINW = waves per
// SIMD the register allocator must leave room for
template <bool WantCC, int RPW, int WPB, int MINW>
__global__ __launch_bounds__(64 * WPB, MINW) void k_compare_tiled(TiledArgs a) {
  constexpr int kRowsPerWave = RPW;
  constexpr int kTR = WPB * RPW;   // rows per tile (<= 64)
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  uint32_t* lenA = sm;            // [64]
  uint32_t* offA = sm + 64;       // [64]
  uint32_t* gA = sm + 128;        // [64] global index of the segment start
  uint32_t* lenB = sm + 192;      // [64]
  uint32_t* gB = sm + 256;        // [64]
  uint32_t* ctl = sm + 320;       // [0], [1] overflow flags
  uint32_t* nrowL = sm + 328;     // [64] truncation length of each row (0xffffffff = none, 0 = no row)
  uint32_t* rowid = sm + 392;     // [64] row of each row slot of the tile (0xffffffff = none)
  uint32_t* colid = sm + 456;     // [64] column of each column slot
  uint32_t* poolA = sm + 520;
  uint32_t* Bt = poolA + a.capA;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // the plan picks ONE of the launched instantiations (rows per wave) -- or none of them
  if (a.st->skip_tiled || a.st->rpw != (uint32_t)RPW || a.st->pf) return;
  const uint32_t ntiles = min(a.st->ntiles, a.tiles_cap);
  // (tiles are pulled off the list dynamically, one stretch of it per XCD first: tiled_take_tile)
  const uint32_t nstretch = (a.use_xcd && ntiles >= 64) ? 8u : 1u;
  const uint32_t chunk = (ntiles + nstretch - 1) / nstretch;
  uint32_t steal = 0;     // (thread 0) stretches given up so far
  while (true) {
  __syncthreads();                 // the previous tile's tables and stage are no longer being read
  if (tid == 0) ctl[2] = tiled_take_tile(a.st, ntiles, nstretch, chunk, steal);
  __syncthreads();
  const uint32_t tix = ctl[2];
  if (tix == 0xffffffffu) break;
  const uint32_t bi = a.tiles[2 * tix], bj = a.tiles[2 * tix + 1];
  if (tid < 64) {
    const uint32_t rs = bi * kTR + tid, cs = bj * kTB + tid;
    rowid[tid] = (tid < kTR && rs < a.nrows) ? (uint32_t)a.rkey[rs] : 0xffffffffu;
    colid[tid] = cs < a.ncols ? (uint32_t)a.ckey[cs] : 0xffffffffu;
  }
  __syncthreads();
  const uint32_t col = colid[lane];
  const bool col_ok = col != 0xffffffffu;

  // per-pair running counts live in registers: every loop over q below is fully unrolled so
  // that the indices are static
  uint32_t ucount[kRowsPerWave], common[kRowsPerWave], cc[WantCC ? kRowsPerWave : 1];
#pragma unroll
  for (int q = 0; q < kRowsPerWave; q++) {
    ucount[q] = 0; common[q] = 0;
    if (WantCC) cc[q] = 0;
  }
  if (tid < 64) {
    const uint32_t row = rowid[tid];
    uint32_t n = 0;
    if (row != 0xffffffffu) { n = a.row_nums ? a.row_nums[row] : a.num; n = n ? n : 0xffffffffu; }
    nrowL[tid] = n;
  }

  // Ranges are fine-grained (sized for the longest sketch); a tile whose sketches are short in
  // this part of rank space walks several of them at once: span m doubles while the segments
  // still fit the LDS stage and halves when they do not.
  uint32_t r = 0, m = 1, cool = 0;
  while (r < a.R) {
    uint32_t mt = m;
    if (cool == 0 && m < 64) mt = m * 2; else if (cool) cool--;
    if (mt > a.R - r) mt = a.R - r;
    bool overflow;
    while (true) {
    // ---- segment table of ranges [r, r + mt)
    if (tid < 64) {
      const uint32_t row = rowid[tid];
      uint32_t lo = 0, hi = 0, g = 0;
      if (row != 0xffffffffu) {
        lo = a.rpart[(size_t)row * (a.R + 1) + r];
        hi = a.rpart[(size_t)row * (a.R + 1) + r + mt];
        g = (uint32_t)a.roff[row] + lo;
      }
      lenA[tid] = hi - lo; gA[tid] = g;
      // exclusive scan of (len + 1 sentinel slot) over the tile's rows, by wave 0
      uint32_t v = tid < kTR ? hi - lo + kPad : 0, incl = v;
      for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      offA[tid] = incl - v;
      const uint32_t totA = __shfl(incl, 63);
      if (tid == 0) ctl[0] = totA > a.capA ? 1u : 0u;
    } else if (tid < 128) {
      const uint32_t c = colid[tid - 64];
      uint32_t lo = 0, hi = 0, g = 0;
      if (c != 0xffffffffu) {
        lo = a.cpart[(size_t)c * (a.R + 1) + r];
        hi = a.cpart[(size_t)c * (a.R + 1) + r + mt];
        g = (uint32_t)a.coff[c] + lo;
      }
      lenB[tid - 64] = hi - lo; gB[tid - 64] = g;
      uint32_t mx = hi - lo;
      for (int off = 32; off; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, off));
      if (tid == 64) ctl[1] = ((mx + kPad) * kTB > a.capBt) ? 1u : 0u;
    }
    __syncthreads();
    overflow = (ctl[0] | ctl[1]) != 0;
    if (!overflow || mt == 1) break;
    __syncthreads();           // everyone has read the flags before the table is rebuilt
    mt >>= 1;
    cool = 16;                 // do not try to grow again for a while
    }
    m = mt;

    if (!overflow) {
      // ---- stage: rows packed (wave w copies its 16 rows), columns transposed (lane = column)
#pragma unroll 4
      for (int q = 0; q < kRowsPerWave; q++) {
        const int t = w * kRowsPerWave + q;
        const uint32_t la = lenA[t], oa = offA[t], g = gA[t];
        for (uint32_t e = lane; e < la; e += 64) poolA[oa + e] = a.rrank[g + e];
        if (lane < (int)kPad) poolA[oa + la + lane] = kSentA;
      }
      {
        const uint32_t lb = lenB[lane], g = gB[lane];
        for (uint32_t e = w; e < lb + kPad; e += WPB) Bt[e * kTB + lane] = e < lb ? a.crank[g + e] : kSent;
      }
      __syncthreads();
      // ---- merge: one pair per lane per row
      const uint32_t lb = lenB[lane];
#pragma unroll
      for (int q = 0; q < kRowsPerWave; q++) {
        const int t = w * kRowsPerWave + q;
        tiled_walk_row<WantCC>(poolA + offA[t], Bt + lane, lenA[t], lb, nrowL[t], ucount[q], common[q], cc[WantCC ? q : 0]);
      }
    } else {
      // ---- rare: this range does not fit LDS for this tile; merge from global memory
      if (tid == 0) atomicAdd(a.ovf_steps, 1ull);
      const uint32_t lb = lenB[lane];
      const uint32_t* B = a.crank + gB[lane];
#pragma unroll
      for (int q = 0; q < kRowsPerWave; q++) {
        const int t = w * kRowsPerWave + q;
        tiled_merge_global_row<WantCC>(a.rrank + gA[t], lenA[t], B, lb, nrowL[t], ucount[q], common[q], cc[WantCC ? q : 0]);
      }
    }
    // ---- all pairs of the tile past their cut: the remaining ranges cannot change common or size
    bool done = !WantCC;
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) done = done && (ucount[q] >= nrowL[w * kRowsPerWave + q]);
    if (__syncthreads_and(done || !col_ok)) {
      // lanes of missing columns vote "done"; real lanes decide
      break;
    }
    r += m;
  }

#pragma unroll
  for (int q = 0; q < kRowsPerWave; q++) {
    const uint32_t row = rowid[w * kRowsPerWave + q];
    if (row != 0xffffffffu && col_ok)
      tiled_write_pair<WantCC>(a, row, col, nrowL[w * kRowsPerWave + q], ucount[q], common[q], WantCC ? cc[q] : 0u);
  }
  }   // tiles of this workgroup
}

// ---- the same tile walk, software-pipelined: while a stretch of rank space is walked, the next ones are on their way ----
// k_compare_tiled spends a third of a workgroup's time in three barriers and two dependent global loads per stretch (where the
// sketches cross the range boundaries, then their ranks) that nothing in the workgroup overlaps with the walk.  Here, while
// stretch i is walked: the ranks of stretch i+1 travel straight into a second LDS stage (global_load_lds: no registers,
// asynchronous), two waves build the table of stretch i+2 from boundary crossings that were prefetched the same way, and
// the crossings of stretch i+3 are requested.  ONE barrier per stretch (the "every pair past its cut?" vote), preceded by the
// wait for what was requested.  Twice the stage, three tables: a workgroup is 8 waves sharing one column stage (4 workgroups
// = 32 waves per CU, as before).  A stretch that does not fit the stage at the span it tried (the span is grown
// speculatively, see k_compare_tiled) is rebuilt with plain loads and halved spans before its ranks are requested.
constexpr uint32_t kPfTab = 5 * 64;                  // lenA, offA, endA, lenB, endB
constexpr uint32_t kPfHeader = 16 + 3 * 64 + 128 + 2 * 128 + 3 * kPfTab;
template <bool WantCC, int RPW, int WPB, int MINW>
__global__ __launch_bounds__(64 * WPB, MINW) void k_compare_tiled_pf(TiledArgs a) {
  constexpr int kRowsPerWave = RPW;
  constexpr int kTR = WPB * RPW;   // rows per tile (<= 64)
  static_assert(WPB >= (int)kPad && kTR <= 64, "waves 0..3 write the column sentinels");
  constexpr uint32_t kNone = 0xffffffffu;
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  uint32_t* ctl = sm;              // [2 s], [2 s + 1]: table slot s does not fit (rows, columns); [8] the tile taken; [9..11] votes; [12 + s] longest column segment
  uint32_t* nrowL = sm + 16;       // [64] truncation length of each row (0xffffffff = none, 0 = no row)
  uint32_t* rowid = sm + 80;       // [64]
  uint32_t* colid = sm + 144;      // [64]
  uint32_t* goff = sm + 208;       // [128] where each row (0..63) / column (64..127) starts in the rank array
  uint32_t* raw = sm + 336;        // [2][128] prefetched boundary crossings (rows, columns), alternating
  uint32_t* tab = sm + 592;        // three tables
  uint32_t* stage = sm + kPfHeader;   // two stages
  const uint32_t stage_dw = a.capA + a.capBt;

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (w: a scalar)
  // What only a tile's set-up and write-out need (the tile list, the keys, offsets, nums, outputs, the pair scope) is read
  // from the kernel-argument segment where it is used, not carried in scalar registers across the stretch loop (which
  // spills them into vector lanes: a v_readlane per use)
  const TiledArgs& ka = *(const TiledArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  // the plan picks a tile height; this instantiation serves one of them -- or none
  if (a.st->skip_tiled || a.st->rpw * 4u != (uint32_t)kTR || !a.st->pf) return;
  const uint32_t ntiles = min(a.st->ntiles, a.tiles_cap);
  const uint32_t nstretch = (a.use_xcd && ntiles >= 64) ? 8u : 1u;
  const uint32_t chunk = (ntiles + nstretch - 1) / nstretch;
  const uint32_t R = a.R;
  uint32_t steal = 0;     // (thread 0) stretches given up so far

  // table slot s for ranges [r, r + mt) (waves 0 and 1): lengths, LDS offsets, where the segments end.  They start where the
  // table in slot p ended (first: read), and end at prefetched crossings (rawp) or at ones read here.
  auto build_table = [&](uint32_t s, uint32_t p, uint32_t r, uint32_t mt, bool first, const uint32_t* rawp) {
    uint32_t* T = tab + s * kPfTab;
    const uint32_t* P = tab + p * kPfTab;
    if (tid < 64) {
      const uint32_t row = rowid[tid];
      uint32_t lo = 0, hi = 0;
      if (row != kNone) {
        lo = first ? a.rpart[(size_t)row * (R + 1) + r] : P[128 + tid];
        hi = rawp ? rawp[tid] : a.rpart[(size_t)row * (R + 1) + r + mt];
        if (hi < lo) { hi = lo; ka.st->bad_tables = 1; }      // (see "The fault of round 3" in DESIGN.md 3.4: reported, never followed)
      }
      T[tid] = hi - lo; T[128 + tid] = hi;
      const uint32_t v = tid < kTR ? hi - lo + kPad : 0, incl = wave_incl_scan_add(v);
      T[64 + tid] = incl - v;
      if (tid == 63) ctl[2 * s] = incl > a.capA ? 1u : 0u;
    } else if (tid < 128) {
      const uint32_t c = colid[tid - 64];
      uint32_t lo = 0, hi = 0;
      if (c != kNone) {
        lo = first ? a.cpart[(size_t)c * (R + 1) + r] : P[256 + tid - 64];
        hi = rawp ? rawp[tid] : a.cpart[(size_t)c * (R + 1) + r + mt];
        if (hi < lo) { hi = lo; ka.st->bad_tables = 1; }
      }
      T[192 + tid - 64] = hi - lo; T[256 + tid - 64] = hi;
      const uint32_t mx = wave_max_u32(hi - lo);
      if (tid == 64) { ctl[2 * s + 1] = ((mx + kPad) * kTB > a.capBt) ? 1u : 0u; ctl[12 + s] = mx; }   // (the longest column segment: issue_stage)
    }
  };
  // the ranks of table slot s's segments -> stage slot g, asynchronously (rows packed: wave w its rows; columns transposed:
  // lane = column); the sentinels behind them with ordinary LDS stores
  auto issue_stage = [&](uint32_t s, uint32_t g) {
    const uint32_t* T = tab + s * kPfTab;
    uint32_t* pA = stage + g * stage_dw;
    uint32_t* pB = pA + a.capA;
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) {
      const int t = w * kRowsPerWave + q;
      const uint32_t la = __builtin_amdgcn_readfirstlane(T[t]), oa = __builtin_amdgcn_readfirstlane(T[64 + t]);   // (scalars)
      const uint32_t idx = goff[t] + T[128 + t] - la + (uint32_t)lane;     // (element indices in 32 bits: one 64-bit add per request)
      if ((uint32_t)lane < la) __builtin_amdgcn_global_load_lds(a.rrank + idx, pA + oa, 4, 0, 0);
      for (uint32_t e0 = 64; e0 < la; e0 += 64)                             // (a segment of more than 64 ranks: rare)
        if (e0 + lane < la) __builtin_amdgcn_global_load_lds(a.rrank + (idx + e0), pA + oa + e0, 4, 0, 0);
      if (lane < (int)kPad) pA[oa + la + lane] = kSentA;
    }
    const uint32_t lb = T[192 + lane];
    const uint32_t idxc = goff[64 + lane] + T[256 + lane] - lb;
    const uint32_t lbmax = __builtin_amdgcn_readfirstlane(ctl[12 + s]);      // (a scalar: the loop is wave-uniform)
    for (uint32_t e = w; e < lbmax; e += WPB)
      if (e < lb) __builtin_amdgcn_global_load_lds(a.crank + (idxc + e), pB + e * kTB, 4, 0, 0);
    if (w < (int)kPad) pB[(lb + w) * kTB + lane] = kSent;      // (waves 0..3: one of the column's four sentinels each)
  };
  // where the rows and columns cross boundary `at`, asynchronously into rawp[] (waves 0 and 1)
  auto issue_raw = [&](uint32_t at, uint32_t* rawp) {
    if (tid < 64) {
      const uint32_t row = rowid[tid];
      if (row != kNone) __builtin_amdgcn_global_load_lds(a.rpart + (size_t)row * (R + 1) + at, rawp, 4, 0, 0);
    } else if (tid < 128) {
      const uint32_t c = colid[tid - 64];
      if (c != kNone) __builtin_amdgcn_global_load_lds(a.cpart + (size_t)c * (R + 1) + at, rawp + 64, 4, 0, 0);
    }
  };
  // the span of the stretch at r after one of span m: doubled while the segments have been fitting (see k_compare_tiled)
  auto next_span = [&](uint32_t m, uint32_t& cool, uint32_t r) {
    uint32_t mt = m;
    if (cool == 0 && m < 64) mt = m * 2; else if (cool) cool--;
    if (mt > R - r) mt = R - r;
    return mt;
  };
  auto flags = [&](uint32_t s) { return (ctl[2 * s] | ctl[2 * s + 1]) != 0; };
  // table slot s (segments start where slot p's ended, or are read when first) with plain loads, the span halved until it
  // fits or is one range; workgroup-wide, ends after a barrier.  Returns "does not fit".
  auto settle_table = [&](uint32_t s, uint32_t p, uint32_t r, uint32_t& mt, bool first, uint32_t& cool) {
    while (true) {
      build_table(s, p, r, mt, first, nullptr);
      __syncthreads();
      const bool o = flags(s);
      if (!o || mt == 1) return o;
      __syncthreads();           // everyone has read the flags before the table is rebuilt
      mt >>= 1; cool = 16;
    }
  };

  while (true) {
  __builtin_amdgcn_s_waitcnt(0x0f70);   // (vmcnt 0: nothing of the previous tile is still on its way into LDS)
  __syncthreads();
  if (tid == 0) {
    ctl[8] = tiled_take_tile(ka.st, ntiles, nstretch, chunk, steal);
    ctl[9] = 0; ctl[10] = 0; ctl[11] = 0;      // the "a pair is short of its cut" flags of three consecutive stretches
  }
  __syncthreads();
  const uint32_t tix = ctl[8];
  if (tix == kNone) break;
  const uint32_t bi = ka.tiles[2 * tix], bj = ka.tiles[2 * tix + 1];
  if (tid < 64) {
    const uint32_t rs = bi * kTR + tid, cs = bj * kTB + tid;
    const uint32_t row = (tid < kTR && rs < ka.nrows) ? (uint32_t)ka.rkey[rs] : kNone;
    const uint32_t c = cs < ka.ncols ? (uint32_t)ka.ckey[cs] : kNone;
    rowid[tid] = row; colid[tid] = c;
    uint32_t n = 0;
    if (row != kNone) { n = ka.row_nums ? ka.row_nums[row] : ka.num; n = n ? n : kNone; }
    nrowL[tid] = n;
    goff[tid] = row != kNone ? (uint32_t)ka.roff[row] : 0u;
    goff[64 + tid] = c != kNone ? (uint32_t)ka.coff[c] : 0u;
  }
  __syncthreads();
  const uint32_t col = colid[lane];
  const bool col_ok = col != kNone;

  uint32_t ucount[kRowsPerWave], common[kRowsPerWave], cc[WantCC ? kRowsPerWave : 1];
#pragma unroll
  for (int q = 0; q < kRowsPerWave; q++) {
    ucount[q] = 0; common[q] = 0;
    if (WantCC) cc[q] = 0;
  }

  // ---- range masks: where every pair's union reaches its cut, without walking (see "range masks" further down).
  // Pass 1, range by range: matches so far = popcounts of the ANDed words; union so far = the two sketches' crossings of the
  // range's upper boundary minus the matches; the first range in which that reaches the row's cut is the pair's.  Pass 2:
  // that ONE range is walked by the pair's lane alone, from the rank arrays (the staged stretch loop below is not entered
  // when the masks exist); pairs that never reach a cut are final as they are.
  const bool use_masks = ka.masks != nullptr && ka.minfo->ok != 0;      // (uniform)
  uint32_t first_r = 0;                 // the first range the tile walks (R: none)
  uint32_t mtot[kRowsPerWave];          // matches over all ranges
  uint32_t nocut = 0;                   // bit q: the pair (my q-th row, my column) needs no walk
  if (use_masks) {
    const uint32_t nsk = ka.nsk;
    const uint32_t gcol = col_ok ? ka.scope.col_base + col : 0u;
    uint32_t grow[kRowsPerWave], nq[kRowsPerWave], rstar[kRowsPerWave], ipre[kRowsPerWave];
    uint32_t self = 0;                  // bit q: the pair is a sketch with ITSELF -- everything matches, also what nobody else holds
    uint32_t samec[kRowsPerWave];       // all ones: row and column are of one component (bits are handed out per component: a pair
                                        // across components can only share FREQUENT hashes, which have words of their own)
    const uint32_t ccomp = (bj * kTB + (uint32_t)lane) < ka.ncols ? (uint32_t)(ka.ckey[bj * kTB + lane] >> 32) : kNone;
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) {
      const uint32_t row = rowid[w * kRowsPerWave + q];
      grow[q] = row != kNone ? ka.scope.row_base + row : 0u;
      nq[q] = (row != kNone && col_ok) ? nrowL[w * kRowsPerWave + q] : kNone;     // (no pair: never cut)
      mtot[q] = 0; rstar[q] = R; ipre[q] = 0;
      if (row != kNone && col_ok && grow[q] == gcol) self |= 1u << q;
      const uint32_t rs = bi * kTR + (uint32_t)(w * kRowsPerWave + q);
      samec[q] = (rs < ka.nrows && (uint32_t)(ka.rkey[rs] >> 32) == ccomp) ? 0xffffffffu : 0u;
    }
    const unsigned long long* mk = ka.masks;
    const uint32_t* pT = ka.partT;
    // What is the same for all lanes -- the ranges' word offsets, the ROWS' words and crossings -- is read through the scalar
    // cache into scalar registers (constant address space: written by earlier launches only), the column's words by vector
    // loads; everything a range needs (up to kMW words per sketch; more: the tail loop) is requested before anything is used.
    // (First version: every load a vector load followed by its own wait -- ~15 dependent round trips per range.)
    const SMH_CONSTANT unsigned long long* mkc = as_constant(ka.masks);
    const SMH_CONSTANT uint32_t* pTc = as_constant(ka.partT);
    const SMH_CONSTANT uint32_t* woffc = as_constant(ka.woff);
    const SMH_CONSTANT uint32_t* wnc = as_constant(ka.wn);
    constexpr int kMW = 3;
    uint32_t growS[kRowsPerWave];
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) growS[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)grow[q]);
    // The usual tile -- every pair of one component, no sketch paired with itself -- takes the short forms: every word counts
    // for every pair (two ANDs and two accumulating popcounts per word and row), the union so far is pa + pb - matches.
    // rstar counts the ranges whose upper boundary the union has NOT reached (the union only grows: those are the first ones);
    // ipre follows the matches while that lasts.
    const uint32_t nsk8 = nsk * 8u, nsk4 = nsk * 4u;         // (the masks stay below 4 GB: see has_masks / lazy_ready)
    const char* colb = reinterpret_cast<const char*>(mk + gcol);
    const char* colp = reinterpret_cast<const char*>(pT + gcol);
    const SMH_CONSTANT char* rowb[kRowsPerWave];
    const SMH_CONSTANT char* rowp[kRowsPerWave];
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) {
      rowb[q] = reinterpret_cast<const SMH_CONSTANT char*>(mkc + growS[q]);
      rowp[q] = reinterpret_cast<const SMH_CONSTANT char*>(pTc + growS[q]);
    }
    bool fastb = self == 0 && col_ok;
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) { fastb = fastb && samec[q] == 0xffffffffu; rstar[q] = 0; }
    const bool fast = __all(fastb);
    for (uint32_t r = 0; r < R; r++) {
      const uint32_t w0 = woffc[r], w1 = woffc[r + 1], wf = w0 + wnc[r];
      const uint32_t nw = w1 - w0;
      unsigned long long cw[kMW], rw[kRowsPerWave][kMW];
      uint32_t pa[kRowsPerWave];
      // The rows' loads are a constant base per row plus ONE 32-bit offset per word that all rows share -- the scalar unit,
      // one per CU, was the busiest part of the kernel when every load had its own 64-bit address arithmetic.  The first word
      // is read whether the range has it or not (the array is padded; what is not the range's is not counted), the others
      // only when it has them: a range of the family collection has one word, and three times the loads cost it 0.35 ms.
      const uint32_t ow = w0 * nsk8, op = (r + 1u) * nsk4;
#pragma unroll
      for (int k = 0; k < kMW; k++) {
        cw[k] = 0ull;
        if (k == 0 || (uint32_t)k < nw) cw[k] = *reinterpret_cast<const unsigned long long*>(colb + (ow + (uint32_t)k * nsk8));
      }
      const uint32_t pb = *reinterpret_cast<const uint32_t*>(colp + op);
#pragma unroll
      for (int k = 0; k < kMW; k++) {
        if (k == 0 || (uint32_t)k < nw) {
#pragma unroll
          for (int q = 0; q < kRowsPerWave; q++) rw[q][k] = *reinterpret_cast<const SMH_CONSTANT unsigned long long*>(rowb[q] + (ow + (uint32_t)k * nsk8));
        } else {
#pragma unroll
          for (int q = 0; q < kRowsPerWave; q++) rw[q][k] = 0ull;
        }
      }
#pragma unroll
      for (int q = 0; q < kRowsPerWave; q++) pa[q] = *reinterpret_cast<const SMH_CONSTANT uint32_t*>(rowp[q] + op);
      bool allfound = true;
      if (fast) {
#pragma unroll
        for (int k = 0; k < kMW; k++) {
          if ((uint32_t)k < nw) {
#pragma unroll
            for (int q = 0; q < kRowsPerWave; q++) {
              // (two accumulating popcounts: v_bcnt_u32_b32 adds its third operand)
              const unsigned long long x = rw[q][k] & cw[k];
              mtot[q] = (uint32_t)__popc((uint32_t)x) + mtot[q];
              mtot[q] = (uint32_t)__popc((uint32_t)(x >> 32)) + mtot[q];
            }
          }
        }
        for (uint32_t wi = w0 + kMW; wi < w1; wi++) {    // (rare: a range with more words)
          const unsigned long long mb = mk[(size_t)wi * nsk + gcol];
#pragma unroll
          for (int q = 0; q < kRowsPerWave; q++) mtot[q] += (uint32_t)__popcll(mkc[(size_t)wi * nsk + growS[q]] & mb);
        }
#pragma unroll
        for (int q = 0; q < kRowsPerWave; q++) {
          const bool below = pa[q] + pb - mtot[q] < nq[q];
          rstar[q] += below ? 1u : 0u;
          ipre[q] = below ? mtot[q] : ipre[q];
          allfound = allfound && !below;
        }
      } else {
#pragma unroll
        for (int k = 0; k < kMW; k++) {
          if ((uint32_t)k < nw) {
            // (a component word counts for pairs of one component only; the range's frequent words, behind them, for every pair)
            const uint32_t keep = w0 + (uint32_t)k < wf ? 0u : 0xffffffffu;
#pragma unroll
            for (int q = 0; q < kRowsPerWave; q++) mtot[q] += (uint32_t)__popcll(rw[q][k] & cw[k]) & (samec[q] | keep);
          }
        }
        for (uint32_t wi = w0 + kMW; wi < w1; wi++) {
          const unsigned long long mb = mk[(size_t)wi * nsk + gcol];
          const uint32_t keep = wi < wf ? 0u : 0xffffffffu;
#pragma unroll
          for (int q = 0; q < kRowsPerWave; q++) mtot[q] += (uint32_t)__popcll(mkc[(size_t)wi * nsk + growS[q]] & mb) & (samec[q] | keep);
        }
#pragma unroll
        for (int q = 0; q < kRowsPerWave; q++) {
          const bool below = pa[q] + pb - (((self >> q) & 1u) ? pa[q] : mtot[q]) < nq[q];
          rstar[q] += below ? 1u : 0u;
          ipre[q] = below ? mtot[q] : ipre[q];
          allfound = allfound && !below;
        }
      }
      // (count_common not wanted: a wave whose pairs have all found their range needs no more of the totals)
      if (!WantCC && (r & 7u) == 7u && __all(allfound)) break;
    }
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) {
      if (rstar[q] == R) nocut |= 1u << q;
      if ((self >> q) & 1u) mtot[q] = pT[(size_t)R * nsk + grow[q]];
    }
    // ---- the ONE range in which a pair's union reaches its cut is walked by the pair's lane alone, straight from the
    // rank arrays; no stage, no table, no barrier -- everything before that range is known from the masks.  The lanes of a
    // wave read 64 different places, and a load instruction costs the memory pipe a cycle per lane whatever its width: so a
    // lane reads FOUR ranks of either side per load (any alignment) and walks four steps out of registers -- the window of a
    // side is shifted down when its head is consumed -- before it asks again: a quarter of the loads and of the round trips of
    // an element-by-element walk.  Two rows of the wave at a time (two independent chains; four would not fit the registers).
    {
      const uint32_t* rr = ka.rrank;
      const uint32_t* cr = ka.crank;
      constexpr int kPair = kRowsPerWave >= 2 ? 2 : 1;
#pragma unroll
      for (int q0 = 0; q0 < kRowsPerWave; q0 += kPair) {
        uint32_t ai[kPair], bi2[kPair], aend[kPair], bend[kPair], left[kPair], mm[kPair];    // (aend, bend: the SKETCHES' ends)
        bool any = false;
#pragma unroll
        for (int j = 0; j < kPair; j++) {
          const int q = q0 + j;
          ai[j] = aend[j] = bi2[j] = bend[j] = left[j] = mm[j] = 0;
          if (!((nocut >> q) & 1u)) {
            const uint32_t row = rowid[w * kRowsPerWave + q], rs = rstar[q];
            const uint32_t pa = pT[(size_t)rs * nsk + grow[q]], pb = pT[(size_t)rs * nsk + gcol];
            const uint32_t before = ((self >> q) & 1u) ? pa : ipre[q];
            ai[j] = (uint32_t)ka.roff[row] + pa; aend[j] = (uint32_t)ka.roff[row + 1];
            bi2[j] = (uint32_t)ka.coff[col] + pb; bend[j] = (uint32_t)ka.coff[col + 1];
            left[j] = nq[q] - (pa + pb - before);          // union elements still to go (> 0: the cut lies in this range)
            mm[j] = before;
            any = true;
          }
        }
        while (__any(any)) {
          Rank4 va[kPair], vb[kPair];
#pragma unroll
          for (int j = 0; j < kPair; j++) {                // (a finished or absent pair reads the first ranks and ignores them)
            va[j] = *reinterpret_cast<const Rank4*>(rr + (left[j] ? ai[j] : 0u));
            vb[j] = *reinterpret_cast<const Rank4*>(cr + (left[j] ? bi2[j] : 0u));
          }
          // What lies past the end of a SEGMENT is the sketch's next range: larger than every rank of this range on either
          // side, so it acts as the sentinel it stands for (it is never matched: the cut comes before both sides run out).
          // Only past the end of the SKETCH do other ranks follow: there -- rarely -- the sentinels are put in by hand.
          bool inside = true;
#pragma unroll
          for (int j = 0; j < kPair; j++) inside = inside && (left[j] == 0 || (ai[j] + 3 < aend[j] && bi2[j] + 3 < bend[j]));
          if (!__all(inside)) {
#pragma unroll
            for (int j = 0; j < kPair; j++) {
              va[j].x = ai[j] < aend[j] ? va[j].x : kSentA; va[j].y = ai[j] + 1 < aend[j] ? va[j].y : kSentA;
              va[j].z = ai[j] + 2 < aend[j] ? va[j].z : kSentA; va[j].w = ai[j] + 3 < aend[j] ? va[j].w : kSentA;
              vb[j].x = bi2[j] < bend[j] ? vb[j].x : kSent; vb[j].y = bi2[j] + 1 < bend[j] ? vb[j].y : kSent;
              vb[j].z = bi2[j] + 2 < bend[j] ? vb[j].z : kSent; vb[j].w = bi2[j] + 3 < bend[j] ? vb[j].w : kSent;
            }
          }
          any = false;
#pragma unroll
          for (int j = 0; j < kPair; j++) {
            uint32_t a0 = va[j].x, a1 = va[j].y, a2 = va[j].z, b0 = vb[j].x, b1 = vb[j].y, b2 = vb[j].z;
            const uint32_t a3 = va[j].w, b3 = vb[j].w;
            // four steps; the window of a side moves down when its head is consumed (only the slots a later step can reach)
#define SMH_STEP(SHIFT_)                                                                  \
            {                                                                             \
              const bool go = left[j] != 0;                                               \
              const bool ta = go && a0 <= b0, tb = go && b0 <= a0;      /* both: a match */ \
              mm[j] += (ta && tb) ? 1u : 0u;                                              \
              ai[j] += ta ? 1u : 0u;                                                      \
              bi2[j] += tb ? 1u : 0u;                                                     \
              left[j] -= go ? 1u : 0u;                                                    \
              SHIFT_                                                                      \
            }
            SMH_STEP(a0 = ta ? a1 : a0; a1 = ta ? a2 : a1; a2 = ta ? a3 : a2; b0 = tb ? b1 : b0; b1 = tb ? b2 : b1; b2 = tb ? b3 : b2;)
            SMH_STEP(a0 = ta ? a1 : a0; a1 = ta ? a2 : a1; b0 = tb ? b1 : b0; b1 = tb ? b2 : b1;)
            SMH_STEP(a0 = ta ? a1 : a0; b0 = tb ? b1 : b0;)
            SMH_STEP(;)
#undef SMH_STEP
            any = any || left[j] != 0;
          }
        }
#pragma unroll
        for (int j = 0; j < kPair; j++)
          if (!((nocut >> (q0 + j)) & 1u)) { common[q0 + j] = mm[j]; ucount[q0 + j] = nq[q0 + j]; }
      }
    }
    first_r = R;            // (nothing is left for the staged walk)
  }

  // ---- stretches 0 and 1 with plain loads; the ranks of stretch 0 and the crossings at the end of stretch 2 requested
  // (s0, s1, s2: table slots of the stretch being walked, the next, the one after; stage slot = stretch number & 1)
  if (first_r < R) {        // (with masks: nothing to walk when no pair of the tile reaches a cut)
  uint32_t cool = 0, it = 0;
  uint32_t r0 = first_r, mt0 = next_span(1, cool, first_r);
  bool ovf0 = settle_table(0, 0, r0, mt0, true, cool);
  if (!ovf0) issue_stage(0, 0);
  uint32_t r1 = r0 + mt0, mt1 = 0;
  bool ovf1 = false;
  if (r1 < R) {
    mt1 = next_span(mt0, cool, r1);
    ovf1 = settle_table(1, 0, r1, mt1, false, cool);
  }
  uint32_t raw_at = 0, raw_slot = 0;   // the boundary whose crossings are (arriving) in raw[raw_slot] (0: none asked for yet)
  if (mt1 && r1 + mt1 < R) {
    uint32_t c2 = cool;
    raw_at = r1 + mt1 + next_span(mt1, c2, r1 + mt1);
    raw_slot ^= 1u;
    issue_raw(raw_at, raw + raw_slot * 128);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  uint32_t s0 = 0, s1 = 1, s2 = 2;
  uint32_t n_halved = 0;    // (stretches of this tile rebuilt with a halved span, and whether prefetched crossings were used
  bool pf_after = false;    //  after one: for the record only, added to the plan's counters once per tile)

  while (true) {
    // ---- here: table s0 = stretch `it` (its ranks in stage it & 1, unless ovf0), table s1 = the next one (mt1 != 0), the
    // crossings at raw_at are in raw[raw_slot]; everything visible to everybody
    if (mt1 && !ovf1) issue_stage(s1, (it & 1u) ^ 1u);
    uint32_t r2 = r1 + mt1, mt2 = 0;
    if (mt1 && r2 < R) {
      mt2 = next_span(mt1, cool, r2);
      // Prefetched crossings are used for the table that ENDS at the boundary they were requested for, and for no other:
      // where a row or column crosses boundary b is a function of b alone, so `raw_at == r2 + mt2` is exact whatever route
      // (spans grown, halved, rebuilt) led to r2 and mt2; a table that ends anywhere else reads its ends with plain loads
      const bool use_raw = raw_at == r2 + mt2;
      pf_after = pf_after || (n_halved && use_raw);
      build_table(s2, s1, r2, mt2, false, use_raw ? raw + raw_slot * 128 : nullptr);
      if (r2 + mt2 < R) {
        uint32_t c2 = cool;
        raw_at = r2 + mt2 + next_span(mt2, c2, r2 + mt2);     // (if table s2 turns out not to fit, this is not the boundary asked for later)
        raw_slot ^= 1u;                                       // (not the slot just read; the other one was last read an iteration ago)
        issue_raw(raw_at, raw + raw_slot * 128);
      }
    }
    // ---- walk stretch `it`
    const uint32_t* T = tab + s0 * kPfTab;
    if (!ovf0) {
      const uint32_t* pA = stage + (it & 1u) * stage_dw;
      const uint32_t* pB = pA + a.capA;
      const uint32_t lb = T[192 + lane];
#pragma unroll
      for (int q = 0; q < kRowsPerWave; q++) {
        const int t = w * kRowsPerWave + q;
        // (with masks count_common is their total: the walk only has to reach the cuts)
        if (WantCC && use_masks) tiled_walk_row<false>(pA + T[64 + t], pB + lane, T[t], lb, nrowL[t], ucount[q], common[q], cc[0]);
        else tiled_walk_row<WantCC>(pA + T[64 + t], pB + lane, T[t], lb, nrowL[t], ucount[q], common[q], cc[WantCC ? q : 0]);
      }
    } else {
      // ---- rare: one range that does not fit LDS for this tile; merge from global memory
      if (tid == 0) atomicAdd(ka.ovf_steps, 1ull);
      const uint32_t lb = T[192 + lane];
      const uint32_t* B = a.crank + goff[64 + lane] + (T[256 + lane] - lb);
#pragma unroll
      for (int q = 0; q < kRowsPerWave; q++) {
        const int t = w * kRowsPerWave + q;
        const uint32_t la = T[t];
        if (WantCC && use_masks) tiled_merge_global_row<false>(a.rrank + goff[t] + (T[128 + t] - la), la, B, lb, nrowL[t], ucount[q], common[q], cc[0]);
        else tiled_merge_global_row<WantCC>(a.rrank + goff[t] + (T[128 + t] - la), la, B, lb, nrowL[t], ucount[q], common[q], cc[WantCC ? q : 0]);
      }
    }
    // ---- all pairs of the tile past their cut: the remaining ranges cannot change common or size
    bool done = !WantCC || use_masks;
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) done = done && (ucount[q] >= nrowL[w * kRowsPerWave + q] || ((nocut >> q) & 1u));
    // the vote with ONE barrier (__syncthreads_and takes three): a wave with a pair still short of its cut raises the flag of
    // this stretch; the flag of the stretch after the next is cleared now (the last readers of that slot -- the stretch
    // before the previous one -- have all passed a barrier since)
    if (tid == 0) ctl[9 + (it + 1) % 3] = 0;
    if (!__all(done || !col_ok) && lane == 0) ctl[9 + it % 3] = 1;
    __builtin_amdgcn_s_waitcnt(0x0f70);        // what this wave requested has arrived
    __syncthreads();
    if (ctl[9 + it % 3] == 0) break;
    if (!mt1) break;
    // ---- the next stretch becomes the current one
    bool ovf2 = mt2 ? flags(s2) : false;
    if (ovf2 && mt2 > 1) {
      // it did not fit at the span it tried: halve it with plain loads until it does (a single range that does not fit stays
      // as it is: it is merged from global memory)
      __syncthreads();
      mt2 >>= 1; cool = 16;
      ovf2 = settle_table(s2, s1, r2, mt2, false, cool);
      n_halved++;
    }
    it++;
    r0 = r1; mt0 = mt1; ovf0 = ovf1;
    r1 = r2; mt1 = mt2; ovf1 = ovf2;
    const uint32_t sx = s0; s0 = s1; s1 = s2; s2 = sx;
  }
  if (tid == 0 && n_halved) {
    atomicAdd(&ka.st->halvings, n_halved);
    if (pf_after) atomicAdd(&ka.st->pf_after_halving, 1u);
  }
  }   // first_r < R

#pragma unroll
  for (int q = 0; q < kRowsPerWave; q++) {
    const uint32_t row = rowid[w * kRowsPerWave + q];
    if (row != kNone && col_ok) {
      uint32_t uc = ucount[q], cm = common[q];
      if (use_masks && ((nocut >> q) & 1u)) {
        // never walked (or walked along for nothing): the union is everything, the matches are the masks' total
        cm = mtot[q];
        uc = (uint32_t)(ka.roff[row + 1] - ka.roff[row]) + (uint32_t)(ka.coff[col + 1] - ka.coff[col]) - cm;
      }
      tiled_write_pair<WantCC>(ka, row, col, nrowL[w * kRowsPerWave + q], uc, cm, WantCC ? (use_masks ? mtot[q] : cc[q]) : 0u);
    }
  }
  }   // tiles of this workgroup
}

// ---- pre-pass kernels ---------------------------------------------------------------------
// Range boundaries of the tiled kernel, as HASH values: out[0] = first hash value of the slice, out[k] = the hash at
// sorted position k*n/Rg of the slice's pool -- Rg ranges with equal shares of the pooled elements.
__global__ void k_hbounds(const uint32_t* __restrict__ starts, const uint64_t* __restrict__ uniq, const RangeState* __restrict__ rs,
                          uint32_t n, uint32_t Rg, const uint64_t* __restrict__ slice_lo_p, uint64_t* __restrict__ out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Rg) return;
  const uint64_t slice_lo = *slice_lo_p;
  const uint32_t nruns = rs->nruns;
  if (k == 0 || nruns == 0) { out[k] = slice_lo; return; }
  const uint32_t pos = (uint32_t)(((uint64_t)k * n) / Rg);
  uint32_t lo = 0, hi = nruns;  // last run with starts[run] <= pos
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (starts[mid] <= pos) lo = mid; else hi = mid;
  }
  out[k] = max(uniq[lo], slice_lo);
}
// part[s][r] = first index in sketch s whose hash is >= hbound[r]  (r < R);  part[s][R] = |s|
// (built by the first block compare that may take the tiled route: skip = that call's plan does not, built = done already)
__global__ void k_partition(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ off, uint32_t nsk,
                            const uint64_t* __restrict__ hbound, uint32_t R, uint32_t* __restrict__ part,
                            const uint32_t* __restrict__ skip, const uint32_t* __restrict__ built) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)nsk * (R + 1) || *skip || *built) return;
  uint32_t s = (uint32_t)(g / (R + 1)), r = (uint32_t)(g % (R + 1));
  const uint64_t* v = hashes + off[s];
  uint32_t len = (uint32_t)(off[s + 1] - off[s]);
  if (r == R) { part[g] = len; return; }
  uint32_t lo = 0, hi = len;
  const uint64_t b = hbound[r];
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (v[mid] < b) lo = mid + 1; else hi = mid;
  }
  part[g] = lo;
}

// ---- collection dictionary: slices of hash space ----------------------------------------------
// The dictionary of a collection (dense order-preserving ranks of all its hashes, components of the
// "shares a hash" graph, frequent hashes, range boundaries) can be built by `world` cooperating owners:
// hash space is cut into `world` slices holding equal shares of the pooled elements (splitters from a
// sorted sample -- every owner computes the same ones from the same collection), owner g gathers slice g
// of EVERY sketch (a contiguous piece of it: sketches are sorted), sorts it, and publishes what it found;
// after one all-gather of those shares everybody assembles the whole dictionary.  world == 1: one slice.
__global__ __launch_bounds__(256) void k_sample_keys(const uint64_t* __restrict__ hashes, uint64_t total, uint32_t S,
                                                     uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < S) out[i] = hashes[(uint64_t)i * total / S];
}
__global__ void k_pick_splitters(const uint64_t* __restrict__ sorted, uint32_t S, uint32_t G, uint64_t* __restrict__ split) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  split[g] = (g == 0 || S == 0) ? 0ull : sorted[(uint64_t)g * S / G];
}
// spart[s][g] = first index of sketch s whose hash is >= split[g]  (g < G);  spart[s][G] = |s|
__global__ __launch_bounds__(256) void k_slice_parts(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ off, uint32_t nsk,
                                                     const uint64_t* __restrict__ split, uint32_t G, uint32_t* __restrict__ spart) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)nsk * (G + 1)) return;
  const uint32_t s = (uint32_t)(t / (G + 1)), g = (uint32_t)(t % (G + 1));
  const uint64_t* v = hashes + off[s];
  const uint32_t len = (uint32_t)(off[s + 1] - off[s]);
  uint32_t lo = 0, hi = len;
  if (g == G) lo = len;
  else if (g > 0) {
    const uint64_t b = split[g];
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (v[mid] < b) lo = mid + 1; else hi = mid;
    }
  }
  spart[t] = lo;
}
// segoff[g][s] = elements of slice g in the sketches before s (exclusive scan down column g; segoff[g][nsk] = slice size).
// One workgroup per slice: every lane sums a stretch of sketches, the stretch sums are scanned in LDS.
__global__ __launch_bounds__(1024) void k_slice_scan(const uint32_t* __restrict__ spart, uint32_t nsk, uint32_t G,
                                                     uint32_t* __restrict__ segoff, uint32_t* __restrict__ sizes) {
  __shared__ uint32_t part_sum[1024];
  const uint32_t g = blockIdx.x, tid = threadIdx.x;
  const uint32_t per = (nsk + 1023) / 1024;
  const uint32_t s0 = min(tid * per, nsk), s1 = min(s0 + per, nsk);
  uint32_t sum = 0;
  for (uint32_t s = s0; s < s1; s++) sum += spart[(size_t)s * (G + 1) + g + 1] - spart[(size_t)s * (G + 1) + g];
  part_sum[tid] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t v = tid >= off ? part_sum[tid - off] : 0u;
    __syncthreads();
    part_sum[tid] += v;
    __syncthreads();
  }
  uint32_t run = part_sum[tid] - sum;
  uint32_t* dst = segoff + (size_t)g * (nsk + 1);
  for (uint32_t s = s0; s < s1; s++) {
    dst[s] = run;
    run += spart[(size_t)s * (G + 1) + g + 1] - spart[(size_t)s * (G + 1) + g];
  }
  if (tid == 1023) { dst[nsk] = part_sum[1023]; sizes[g] = part_sum[1023]; }
}
// slice g of every sketch, sketch after sketch: keys[t], node[t] = its sketch, org[t] = t
__global__ __launch_bounds__(256) void k_slice_gather_pos(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ off, uint32_t nsk,
                                                          const uint32_t* __restrict__ spart, uint32_t G, uint32_t g,
                                                          const uint32_t* __restrict__ segoff_g, uint32_t n,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ node, uint32_t* __restrict__ parent) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nsk; i += gridDim.x * blockDim.x) parent[i] = i;   // (the union-find forest's start)
  // a lane takes 8 consecutive slice positions: one search for the sketch of the first, then a walk (the sort numbers the keys).
  // Indifferent to how long a sketch is: the form for collections with sketches far longer than the others.
  const uint32_t t0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8u;
  if (t0 >= n) return;
  uint32_t lo = 0, hi = nsk;   // last s with segoff_g[s] <= t0  (its segment is not empty: t0 < n)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (segoff_g[mid] <= t0) lo = mid; else hi = mid;
  }
  uint32_t seg0 = segoff_g[lo], next = segoff_g[lo + 1];
  const uint64_t* src = hashes + off[lo] + spart[(size_t)lo * (G + 1) + g];
  const uint32_t t1 = min(t0 + 8u, n);
  for (uint32_t t = t0; t < t1; t++) {
    while (t >= next) {          // (sketches with nothing in this slice are stepped over)
      lo++;
      seg0 = next; next = segoff_g[lo + 1];
      src = hashes + off[lo] + spart[(size_t)lo * (G + 1) + g];
    }
    keys[t] = src[t - seg0];
    node[t] = lo;
  }
}
__global__ __launch_bounds__(256) void k_slice_gather(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ off, uint32_t nsk,
                                                      const uint32_t* __restrict__ spart, uint32_t G, uint32_t g,
                                                      const uint32_t* __restrict__ segoff_g, uint32_t n,
                                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ node, uint32_t* __restrict__ parent) {
  // a wavefront per sketch: its piece of the slice is a contiguous stretch of the sketch, copied by consecutive lanes
  // (a lane per 8 slice positions -- a search and eight dependent loads each -- took 63 us for 2.5 M keys)
  const uint32_t sk = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (sk >= nsk) return;
  if (lane == 0) parent[sk] = sk;                          // (the union-find forest's start)
  const uint32_t a = segoff_g[sk], b = segoff_g[sk + 1];
  const uint64_t* src = hashes + off[sk] + spart[(size_t)sk * (G + 1) + g];
  for (uint32_t e = lane; e < b - a; e += 64u) { keys[a + e] = src[e]; node[a + e] = sk; }
  (void)n;
}
// world == 1: the one slice is the collection itself, in its own order (the sort reads it in place and numbers it); what is
// left to make is node[t] = the sketch of element t.  A lane takes 8 consecutive elements: one search, then a walk.
// (and what else the pre-pass needs before its first real kernel: the union-find forest's start, parent[i] = i)
__global__ __launch_bounds__(256) void k_whole_nodes(const uint64_t* __restrict__ off, uint32_t nsk, uint32_t n, uint32_t* __restrict__ node,
                                                     uint32_t* __restrict__ parent) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nsk; i += gridDim.x * blockDim.x) parent[i] = i;
  const uint32_t t0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8u;
  if (t0 >= n) return;
  uint32_t lo = 0, hi = nsk;   // last s with off[s] <= t0
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= t0) lo = mid; else hi = mid;
  }
  uint64_t next = off[lo + 1];
  const uint32_t t1 = min(t0 + 8u, n);
  uint32_t v[8];
#pragma unroll
  for (uint32_t j = 0; j < 8; j++) {
    const uint32_t t = t0 + j;
    while (t < t1 && (uint64_t)t >= next) { lo++; next = off[lo + 1]; }     // (empty sketches are stepped over)
    v[j] = lo;
  }
  if (t1 - t0 == 8) {
    *reinterpret_cast<uint4*>(node + t0) = make_uint4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<uint4*>(node + t0 + 4) = make_uint4(v[4], v[5], v[6], v[7]);
  } else {
    for (uint32_t j = 0; j < t1 - t0; j++) node[t0 + j] = v[j];
  }
}

// where the segments (sketches) of a slice start inside its key array: the collection's own offsets (one owner) or the
// slice's table (k_slice_scan)
struct BkSeg { const uint64_t* off64; const uint32_t* off32; };
// the sketch of pooled element o: a table -- or, when one owner's sketches all have the same length (bottom-num sketches:
// the usual case), a multiplication: the table's entries are scattered reads, and the kernels that go through the sorted
// positions are bound by the scattered lines they touch
struct NodeMap { const uint32_t* node; uint32_t len, magic; };        // len == 0: use the table; magic = floor(2^32 / len)
__device__ __forceinline__ uint32_t node_of(const NodeMap& m, uint32_t o) {
  if (m.len == 0) return m.node[o];
  uint32_t q = __umulhi(o, m.magic);                     // floor(o / len) or one less
  if (o - q * m.len >= m.len) q++;
  return q;
}
__device__ __forceinline__ uint32_t bk_seg(const BkSeg& g, uint32_t s) { return g.off64 ? (uint32_t)g.off64[s] : g.off32[s]; }

// ---- the pooled sort in four passes instead of eight -------------------------------------------------------------------
// The order of the pooled hashes is all the dictionary needs, and 32 bits decide it for all but a few of them: N keys
// uniform over a span leave ~N^2 / 2^33 pairs that tie in their 32 most significant VARYING bits (2 M keys: ~500 pairs;
// 20 M: ~50 000).  So the LSD sort runs over those 32 bits only -- bits [s0, s0 + 32), s0 = (bits below the first one in
// which the smallest and the largest key differ) - 32, found on the device by k_key_span and read by the sort's kernels
// from there -- and k_tie_fix puts the ties in order: the passes are stable, so keys that tie sit next to each other in
// their original order; the lane at the head of a group of at most 4 sorts it in registers, runs of EQUAL keys need nothing,
// and a longer group that is out of order raises rs->overflow -- the slice is then sorted again with all eight passes
// (never seen on hashed keys; keys crafted to differ only in their low bits pay for it).
// A segment (sketch) is sorted, so its two ends are its extremes, and the highest bit in which ANY two keys of the slice
// differ is the highest bit of OR(end ^ reference) over all segment ends.  A lane per sketch; the last workgroup to finish
// turns the OR into the shift.  (One workgroup looping over 10 000 sketches took 47 us: two dependent loads per turn, most of
// them TLB misses in a 160 MB array.)
__global__ __launch_bounds__(256) void k_key_span(const uint64_t* __restrict__ keys, BkSeg seg, uint32_t nsk, uint32_t n, RangeState* rs) {
  const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long d = 0ull;
  if (sidx < nsk) {
    const uint32_t a = bk_seg(seg, sidx), b = bk_seg(seg, sidx + 1);
    if (b > a) { const unsigned long long ref = keys[n - 1]; d = (keys[a] ^ ref) | (keys[b - 1] ^ ref); }
  }
  for (int off = 32; off; off >>= 1) d |= (unsigned long long)__shfl_xor(d, off);
  __shared__ unsigned long long red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    d = red[0] | red[1] | red[2] | red[3];
    if (d) atomicOr(&rs->span_or, d);
    __threadfence();
    if (atomicAdd(&rs->span_done, 1u) == gridDim.x - 1) {
      const unsigned long long all = __hip_atomic_load(&rs->span_or, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t hb = all ? 64u - (uint32_t)__builtin_clzll(all) : 0u;    // bits below the common prefix
      rs->sort_shift = hb > 32u ? hb - 32u : 0u;
    }
  }
}
// first / one-past-last sorted position whose key has the bits above `sh` of `p`
__device__ __forceinline__ uint32_t tie_lower(const uint64_t* keys, uint32_t n, uint32_t sh, uint64_t p) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((keys[mid] >> sh) < p) lo = mid + 1; else hi = mid; }
  return lo;
}
constexpr uint32_t kTieListMax = 1u << 16;
struct TieList { uint32_t n; uint32_t pad; uint32_t at[kTieListMax]; };
// Phase 1, a lane per sorted position.  Head of a group of at most 4: sorts it in registers.  Inside a longer group and out
// of order with its predecessor (a run of equal hashes -- the same hash in many sketches -- with a different key that ties
// with it in the sorted bits): the group goes on the list of k_tie_sort, once (a bit per group start).
__global__ __launch_bounds__(256) void k_tie_fix(uint64_t* __restrict__ keys, uint32_t* __restrict__ org, uint32_t n, RangeState* rs,
                                                 uint32_t* __restrict__ seen, TieList* __restrict__ list) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const uint32_t sh = rs->sort_shift;
  if (sh == 0) return;                                    // the passes covered every bit
  const uint64_t k0 = keys[e];
  const uint64_t p = k0 >> sh;
  const bool with_prev = e && (keys[e - 1] >> sh) == p;
  if (with_prev) {
    if (keys[e - 1] > k0) {
      uint32_t lo = e - 1, hi = e + 1;
      while (lo > 0 && e - lo < 4 && (keys[lo - 1] >> sh) == p) lo--;
      while (hi < n && hi - lo < 5 && (keys[hi] >> sh) == p) hi++;
      if (hi - lo > 4) {                                  // (a short group is put in order by the lane at its head)
        // the group's first position: galloping back from here (groups are short next to the array), then bisecting
        uint32_t step = 4, hi_b = lo;                        // (keys[lo] has the prefix)
        while (hi_b >= step && (keys[hi_b - step] >> sh) == p) { hi_b -= step; step <<= 1; }
        const uint32_t from = hi_b >= step ? hi_b - step : 0u;
        const uint32_t a = from + tie_lower(keys + from, hi_b - from, sh, p);
        if ((atomicOr(&seen[a >> 5], 1u << (a & 31u)) >> (a & 31u) & 1u) == 0) {
          const uint32_t q = atomicAdd(&list->n, 1u);
          if (q < kTieListMax) list->at[q] = a; else rs->overflow = 1;
        }
      }
    }
    return;
  }
  uint64_t k[4] = {k0, 0, 0, 0};
  uint32_t g = 1;
  while (g < 4 && e + g < n && (keys[e + g] >> sh) == p) { k[g] = keys[e + g]; g++; }
  if (g == 1 || (g == 4 && e + 4 < n && (keys[e + 4] >> sh) == p)) return;     // alone, or a long group (its members look for themselves)
  bool sorted = true;
#pragma unroll
  for (uint32_t i = 1; i < 4; i++) sorted = sorted && (i >= g || k[i - 1] <= k[i]);
  if (sorted) return;
  uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
  for (uint32_t i = 0; i < 4; i++) if (i < g) v[i] = org[e + i];
#pragma unroll
  for (uint32_t i = 1; i < 4; i++)
#pragma unroll
    for (uint32_t j = i; j > 0; j--)
      if (i < g && k[j] < k[j - 1]) {
        const uint64_t tk = k[j]; k[j] = k[j - 1]; k[j - 1] = tk;
        const uint32_t tv = v[j]; v[j] = v[j - 1]; v[j - 1] = tv;
      }
#pragma unroll
  for (uint32_t i = 0; i < 4; i++) if (i < g) { keys[e + i] = k[i]; org[e + i] = v[i]; }
}
// Phase 2: the listed groups, one workgroup each, sorted in LDS (up to 8192 keys; the passes whose byte is the same for
// every key of the group skip themselves).  A longer group gives up: rs->overflow.
__global__ __launch_bounds__(kBsThreads) void k_tie_sort(uint64_t* __restrict__ keys, uint32_t* __restrict__ org, uint32_t n, RangeState* rs,
                                                         const TieList* __restrict__ list) {
  __shared__ BlockSortLds<kBlockSortMax, true, uint32_t> L;
  __shared__ uint32_t ends[2];
  const uint32_t t = threadIdx.x, sh = rs->sort_shift;
  const uint32_t m = min(list->n, kTieListMax);
  for (uint32_t q = blockIdx.x; q < m; q += gridDim.x) {
    __syncthreads();
    const uint32_t a = list->at[q];
    if (t == 0) ends[0] = 0xffffffffu;
    __syncthreads();
    {
      // where the group ends: every lane looks at a few positions behind the start, the first one with another prefix wins
      const uint64_t p = keys[a] >> sh;
      uint32_t first = 0xffffffffu;
      for (uint32_t i = t; i <= (uint32_t)kBlockSortMax; i += kBsThreads) {
        const uint32_t pos = a + i;
        if (pos >= n || (keys[pos] >> sh) != p) { first = i; break; }
      }
      if (first != 0xffffffffu) atomicMin(&ends[0], first);
    }
    __syncthreads();
    const uint32_t g = ends[0];
    if (g > (uint32_t)kBlockSortMax) { if (t == 0) rs->overflow = 1; continue; }
    if (g <= 64) {
      // a short group (the usual one: a run of a few equal hashes and a stranger): one wave, a key per lane, every key's
      // place = the keys before it in (key, arrival) order, counted with 64 broadcasts
      if (t < 64) {
        const bool in = t < g;
        const uint64_t k = in ? keys[a + t] : ~0ull;
        const uint32_t v = in ? org[a + t] : 0u;
        uint32_t place = 0;
        for (uint32_t j = 0; j < g; j++) {
          const uint64_t kj = __shfl(k, (int)j);
          place += (kj < k || (kj == k && j < t)) ? 1u : 0u;
        }
        if (in) { keys[a + place] = k; org[a + place] = v; }
      }
      continue;
    }
    const uint32_t items = (g + kBsThreads - 1) / kBsThreads;
    for (uint32_t i = t; i < items * kBsThreads; i += kBsThreads) {
      L.sk[i] = i < g ? keys[a + i] : ~0ull;
      L.si[i] = i < g ? org[a + i] : 0u;
    }
    block_sort_passes(L, g, items, 0, (int)((sh + 7u) & ~7u));       // (the bits above sh are the same in the whole group)
    for (uint32_t i = t; i < g; i += kBsThreads) { keys[a + i] = L.sk[i]; org[a + i] = L.si[i]; }
  }
}


// ---- range masks: how many hashes two sketches share in every range, without walking them --------------------------------
// A hash held by ONE sketch of the collection can never be a match.  The hashes held by two or more (the runs of length >= 2
// of the pooled sort) get a bit each, distinct among the shared hashes of their (component, range) -- see below -- in
// ceil(K_r / 64) 64-bit words per range r of the tiled kernel.  mask[w][s] = which of them sketch s holds.  Then
//     |A and B in range r|  =  popcount(mask[.][A] & mask[.][B])  over the range's words,
// and with the sketches' crossings of the range boundaries (part) the size of the union up to any boundary follows without
// touching a rank: U_r = part[A][r+1] + part[B][r+1] - matches up to r.  The tiled kernel uses that to find, for every pair
// of a tile, the ONE range in which the union reaches the pair's cut (src/lib.rs:470-499: the walk ends after `num` union
// elements); the pair's lane then walks that range alone, with the counts the masks give for everything before it; pairs
// that never reach a cut (scaled sketches: num = 0) are not walked at all, and count_common is the popcount over all
// ranges.  One family of related genomes has a few dozen shared hashes per range: one word.  A component of many unrelated
// families would need many words per range for sparse masks: beyond 2 R + kMaskWordsExtra words per sketch the masks are
// not built (MaskInfo.ok = 0) and the kernel walks as before.
constexpr uint32_t kSidNone = 0xffffffffu;
// Bits only have to be DISTINCT among the hashes two comparable sketches can share in a range, so they are handed out per
// (component, range) by a counter: bit = how many shared hashes of that component and range came before (in any order).
// Sketches of different components are never walked against each other and reuse the same bits: 50 families of 200 genomes
// need the words of one family.  A frequent hash (set aside: it connects nothing, its holders sit in any component) gets
// a bit of its range's "frequent" words, which every component's words are followed by.
// rlo[k] = the first run of range k (k_hbounds's boundaries), rlo[R] = nruns
__global__ __launch_bounds__(256) void k_range_runs(const uint32_t* __restrict__ starts, const RangeState* __restrict__ rs, uint32_t n,
                                                    uint32_t R, uint32_t* __restrict__ rlo) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > R) return;
  const uint32_t nruns = rs->nruns;
  uint32_t v = 0;
  if (k == R) v = nruns;
  else if (k > 0 && nruns) {
    const uint32_t pos = (uint32_t)(((uint64_t)k * n) / R);
    uint32_t lo = 0, hi = nruns;  // last run with starts[run] <= pos
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (starts[mid] <= pos) lo = mid; else hi = mid;
    }
    v = lo;
  }
  rlo[k] = v;
}
// runbit[run] = the bit of a run held by two sketches or more (bit 31: one of the range's frequent bits), kSidNone for a
// run of one.  cnt[root * R + range] / fcnt[range]: the counters.
__global__ __launch_bounds__(256) void k_shared_bits(const uint32_t* __restrict__ starts, const RangeState* __restrict__ rs, uint32_t n,
                                                     const uint32_t* __restrict__ origin, NodeMap node,
                                                     const uint32_t* __restrict__ roots, const uint8_t* __restrict__ isfreq,
                                                     const uint32_t* __restrict__ rlo, uint32_t R, uint32_t* __restrict__ cnt,
                                                     uint32_t* __restrict__ fcnt, uint32_t* __restrict__ runbit) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nruns = rs->nruns;
  if (r >= nruns) return;
  const uint32_t a = starts[r], len = (r + 1 < nruns ? starts[r + 1] : n) - a;
  if (len < 2) { runbit[r] = kSidNone; return; }
  uint32_t lo = 0, hi = R;        // last range k with rlo[k] <= r
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (rlo[mid] <= r) lo = mid; else hi = mid;
  }
  if (isfreq && isfreq[r]) runbit[r] = atomicAdd(&fcnt[lo], 1u) | 0x80000000u;
  else runbit[r] = atomicAdd(&cnt[(size_t)roots[node_of(node, origin[a])] * R + lo], 1u);
}
// rank and bit of every element, in collection order, in ONE pass over the sorted positions (the runs' bits are read in
// run order; the scatter by origin is the one the ranks need anyway -- with masks it runs here instead of inside the
// run-length write).  ebit: the element's bit (bit 15: a frequent bit), 0xffff = none.
__global__ __launch_bounds__(256) void k_rank_bit_scatter(const uint32_t* __restrict__ runid, const uint32_t* __restrict__ origin,
                                                          const uint32_t* __restrict__ runbit, uint32_t n, uint32_t* __restrict__ rank,
                                                          uint16_t* __restrict__ ebit) {
  const uint32_t i = xcd_chunked_block() * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t run = runid[i], o = origin[i], b = runbit[run];
  rank[o] = run;
  ebit[o] = b == kSidNone ? (uint16_t)0xffffu : (uint16_t)((b & 0x7fffu) | ((b >> 31) << 15));
}
// kmax[r] = the most shared hashes any component has in range r (few counters are not zero: one atomic each)
__global__ __launch_bounds__(256) void k_mask_max(const uint32_t* __restrict__ cnt, uint64_t m, uint32_t R, uint32_t* __restrict__ kmax,
                                                  uint32_t* __restrict__ total) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t v = cnt[i];
  if (v) { atomicMax(&kmax[i % R], v); atomicAdd(total, v); }
}
// per range: the words its busiest component needs (wn), the frequent words behind them, the offsets; and whether the masks
// are worth building.  One workgroup; lane per range for the maximum over the components' counters.
// (lazy_go: the plan walks tiles, the masks do not exist yet, and there are enough pairs that can share a hash for them to
// pay -- building them is two passes over ALL pooled hashes plus the masks themselves, whatever part of the matrix this owner
// computes: ~0.35 ms at 20 M hashes, against ~0.35 ns saved per pair.  Measured on the 50-family collection: worth it for
// one rank of two, a wash for one of four or eight.)
__device__ __forceinline__ bool lazy_go(const PlanState* st, const uint32_t* built, uint32_t n) {
  return !st->skip_tiled && !*built && st->pairs * 32ull >= (unsigned long long)n;
}
__global__ __launch_bounds__(1024) void k_mask_layout(const uint32_t* __restrict__ kmax, const uint32_t* __restrict__ fcnt,
                                                      const uint32_t* __restrict__ total, uint32_t R, uint32_t wmax,
                                                      uint32_t* __restrict__ wn, uint32_t* __restrict__ woff, MaskInfo* __restrict__ info,
                                                      const PlanState* __restrict__ plan = nullptr, const uint32_t* __restrict__ built = nullptr,
                                                      uint32_t n_lazy = 0) {
  __shared__ uint32_t wt[16], st[16];
  if (plan && !lazy_go(plan, built, n_lazy)) return;
  const uint32_t per = (R + 1023) / 1024, r0 = min(threadIdx.x * per, R), r1 = min(r0 + per, R);
  __shared__ uint32_t wide;
  if (threadIdx.x == 0) wide = 0;
  __syncthreads();
  uint32_t mine = 0, shared = 0;
  for (uint32_t r = r0; r < r1; r++) {
    const uint32_t words = (kmax[r] + 63u) >> 6;
    wn[r] = words;
    mine += words + ((fcnt[r] + 63u) >> 6);
    shared += fcnt[r];
    if (kmax[r] >= 0x7fffu || fcnt[r] >= 0x7fffu) wide = 1;         // (an element's bit is kept in 15 bits)
  }
  const uint32_t incl = wave_incl_scan_add(mine);
  uint32_t sh = shared;
  for (int off = 32; off; off >>= 1) sh += __shfl_xor(sh, off);
  if ((threadIdx.x & 63) == 63) wt[threadIdx.x >> 6] = incl;
  if ((threadIdx.x & 63) == 0) st[threadIdx.x >> 6] = sh;
  __syncthreads();
  const uint32_t toowide = wide;
  uint32_t run = incl - mine, tot = 0, stot = *total;
  for (uint32_t ww = 0; ww < 16; ww++) { if (ww < (threadIdx.x >> 6)) run += wt[ww]; tot += wt[ww]; stot += st[ww]; }
  for (uint32_t r = r0; r < r1; r++) { woff[r] = run; run += wn[r] + ((fcnt[r] + 63u) >> 6); }
  if (threadIdx.x == 0) {
    woff[R] = tot;
    info->ok = (tot <= wmax && stot > 0 && !toowide) ? 1u : 0u; info->wtot = tot; info->shared = stot; info->pad = 0;
  }
}
// mask[w][s] for the words of range r, and the crossings once more with the range first: partT[r][s] = part[s][r]
// (the tiled kernel reads both with its 64 columns in consecutive lanes).  A lane per (sketch, range).
__global__ __launch_bounds__(256) void k_build_masks(const uint32_t* __restrict__ part, const uint64_t* __restrict__ off,
                                                     const uint16_t* __restrict__ ebit,
                                                     const uint32_t* __restrict__ wn,
                                                     const uint32_t* __restrict__ woff, const MaskInfo* __restrict__ info, uint32_t nsk,
                                                     uint32_t R, unsigned long long* __restrict__ mask, uint32_t* __restrict__ partT,
                                                     const uint32_t* __restrict__ skip, const uint32_t* __restrict__ built) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)nsk * (R + 1) || *skip || *built || !info->ok) return;
  // (range fastest: a wave reads 64 consecutive crossings and 64 consecutive stretches of ONE sketch's ranks; the few
  // stores -- a word and a crossing per lane -- are the scattered side)
  const uint32_t s = (uint32_t)(g / (R + 1)), r = (uint32_t)(g % (R + 1));
  const uint32_t a = part[(size_t)s * (R + 1) + r];
  partT[(size_t)r * nsk + s] = a;
  if (r == R) return;
  const uint32_t b = part[(size_t)s * (R + 1) + r + 1];
  const uint16_t* e = ebit + off[s];
  const uint32_t w0 = woff[r], w1 = woff[r + 1], wf = w0 + wn[r];       // [w0, wf): the component's words; [wf, w1): the frequent ones
  for (uint32_t w = w0; w < w1; w++) {
    unsigned long long m = 0;
    const uint32_t lo = w < wf ? ((w - w0) << 6) : (0x8000u | ((w - wf) << 6));
    for (uint32_t i = a; i < b; i++) {
      const uint32_t v = (uint32_t)e[i] - lo;       // (none = 0xffff, or a bit of the other kind: never below 64)
      if (v < 64u) m |= 1ull << v;
    }
    mask[(size_t)w * nsk + s] = m;
  }
}
constexpr uint32_t kMaskWordsExtra = 64;       // words beyond two per range the masks may take
constexpr uint64_t kCompPairsWithMasks = 16ull << 10;   // the AUTO route's pair limit when the dictionary has (or can build) masks
// masks are kept for collections of at most 8 Mi (sketch, range) pairs whose mask table stays below 4 GB (32-bit byte offsets)
static bool masks_fit(uint64_t n, uint64_t R) {
  return n * R <= (8ull << 20) && n * (2 * R + kMaskWordsExtra + 3) * 8 < (1ull << 32);
}

// What an owner publishes about its slice.  The share is [SliceHeader][roots: nsk u32][hbound: Rg u64][ranks: nmax u32].
struct SliceHeader {
  uint32_t n_elems, nruns, nfreq;
  uint32_t id_space;             // the slice's local ranks are below this (= nruns: they are dense)
  uint64_t freq_hash[64];
};
struct DictState;
__global__ __launch_bounds__(64) void k_slice_header(const RangeState* __restrict__ rs, const uint64_t* __restrict__ uniq, uint32_t n,
                                                     SliceHeader* __restrict__ h, DictState* single_owner, uint32_t sparse_ids);
// What everybody derives from the gathered headers: the rank offset of every slice, the frequent hashes (ascending)
struct DictState {
  uint32_t nruns;              // distinct hashes of the collection
  uint32_t nfreq;              // frequent hashes set aside, over all slices (<= kMaxFreq)
  uint32_t part_built;         // the range partition table exists (k_partition, k_plan_geometry)
  uint32_t overflow;           // one owner: the four-pass sort gave up (RangeState::overflow) -- the dictionary is void and is built again
  uint32_t rbase[64];          // dense rank of the first hash of slice g
  uint64_t freq_hash[64];
};
// (ds: a single owner's header IS the collection's state -- one launch less)
__global__ __launch_bounds__(64) void k_slice_header(const RangeState* __restrict__ rs, const uint64_t* __restrict__ uniq, uint32_t n,
                                                     SliceHeader* __restrict__ h, DictState* ds, uint32_t sparse_ids) {
  const uint32_t k = threadIdx.x;
  const uint64_t f = k < rs->nfreq ? uniq[rs->freq_run[k]] : 0ull;
  if (k == 0) { h->n_elems = n; h->nruns = rs->nruns; h->nfreq = rs->nfreq; h->id_space = sparse_ids ? n : rs->nruns; }
  h->freq_hash[k] = f;
  if (ds) {
    if (k == 0) { ds->nruns = rs->nruns; ds->nfreq = rs->nfreq; ds->part_built = 0; ds->overflow = rs->overflow; }
    ds->rbase[k] = 0;
    ds->freq_hash[k] = f;
  }
}
__global__ void k_dict_state(const uint8_t* __restrict__ gathered, uint64_t share_bytes, uint32_t G, DictState* __restrict__ ds) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t base = 0, nf = 0;
  for (uint32_t g = 0; g < G; g++) {
    const SliceHeader* h = reinterpret_cast<const SliceHeader*>(gathered + (size_t)g * share_bytes);
    ds->rbase[g] = base;
    base += h->id_space;
    for (uint32_t k = 0; k < h->nfreq && nf < 64; k++) ds->freq_hash[nf++] = h->freq_hash[k];
  }
  ds->nruns = base;
  ds->nfreq = nf;
  ds->part_built = 0;
  ds->overflow = 0;
}
// ---- the same bits from an ASSEMBLED dictionary (built by `world` owners, each sorting one slice of hash space): there are no
// runs to look at -- every owner has the ranks of all elements, the roots and the crossings, and builds the masks of all
// sketches from those: how often every rank occurs (a histogram: 20 M atomic adds on ~10 M words), then ONE element of
// every rank that occurs twice or more claims the rank's bit (atomicCAS), takes it from the counter of its (component,
// range) -- the range by a search of its sketch's crossings -- and every element reads its rank's bit.
constexpr uint16_t kEbitShared = 0xfffeu, kEbitHead = 0xfffdu;   // (an element's state before its bit is known; 0xffff: no bit)
// Masks of a sliced dictionary, built from the assembled ranks.  The owners of the slices have told, with every rank, whether
// its hash is held more than once and which element is the first of its run (k_rle_write's rank flags, unpacked into ebit by
// k_reassemble): the FIRST element of every shared hash hands out the bit (one atomic per shared hash on the counter of its
// (component, range); frequent hashes number themselves per range), every other element reads it.  Two passes over the
// pooled hashes.  (First version: a histogram of the ranks and a CAS per element to elect the one that claims -- 40 M atomics
// at random addresses, 2.2 ms at 20 M hashes.)
__global__ __launch_bounds__(256) void k_claim_bits(const uint32_t* __restrict__ rank, const uint64_t* __restrict__ hashes,
                                                    const uint64_t* __restrict__ off, const uint32_t* __restrict__ part,
                                                    const uint32_t* __restrict__ roots, const DictState* __restrict__ ds, uint32_t nsk,
                                                    uint32_t R, uint32_t n, const uint16_t* __restrict__ ebit, uint32_t* __restrict__ bitof,
                                                    uint32_t* __restrict__ cnt, uint32_t* __restrict__ fcnt, uint32_t split,
                                                    const PlanState* __restrict__ st, const uint32_t* __restrict__ built) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n || !lazy_go(st, built, n)) return;
  if (ebit[t] != kEbitHead) return;
  // my sketch (last s with off[s] <= t), my place in it, the range that place falls into (last r with part[s][r] <= place)
  uint32_t lo = 0, hi = nsk;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= t) lo = mid; else hi = mid; }
  const uint32_t sk = lo, place = t - (uint32_t)off[sk];
  const uint32_t* p = part + (size_t)sk * (R + 1);
  lo = 0; hi = R;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (p[mid] <= place) lo = mid; else hi = mid; }
  const uint32_t r = lo;
  bool freq = false;
  if (split) {
    const uint64_t h = hashes[t];
    const uint32_t nf = ds->nfreq;
    for (uint32_t k = 0; k < nf; k++) freq = freq || ds->freq_hash[k] == h;
  }
  bitof[rank[t]] = freq ? (atomicAdd(&fcnt[r], 1u) | 0x80000000u) : atomicAdd(&cnt[(size_t)roots[sk] * R + r], 1u);
}
__global__ __launch_bounds__(256) void k_elem_bits(const uint32_t* __restrict__ rank, uint32_t n, const uint32_t* __restrict__ bitof,
                                                   uint16_t* __restrict__ ebit, const PlanState* __restrict__ st,
                                                   const uint32_t* __restrict__ built) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n || !lazy_go(st, built, n)) return;
  if (ebit[t] == (uint16_t)0xffffu) return;
  const uint32_t b = bitof[rank[t]];
  ebit[t] = (uint16_t)((b & 0x7fffu) | ((b >> 31) << 15));
}
// (the two single-launch steps of the layout, for the lazily built masks: nothing when the plan skips the tiles or they exist)
__global__ __launch_bounds__(256) void k_mask_max_lazy(const uint32_t* __restrict__ cnt, uint64_t m, uint32_t R, uint32_t* __restrict__ kmax,
                                                       uint32_t* __restrict__ total, const PlanState* __restrict__ st,
                                                       const uint32_t* __restrict__ built, uint32_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m || !lazy_go(st, built, n)) return;
  const uint32_t v = cnt[i];
  if (v) { atomicMax(&kmax[i % R], v); atomicAdd(total, v); }
}
// the slices' range boundaries one after the other
__global__ __launch_bounds__(256) void k_gather_bounds(const uint8_t* __restrict__ gathered, uint64_t share_bytes, uint64_t hbound_at,
                                                       uint32_t Rg, uint32_t R, uint64_t* __restrict__ hbound) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const uint32_t g = i / Rg, k = i - g * Rg;
  hbound[i] = reinterpret_cast<const uint64_t*>(gathered + (size_t)g * share_bytes + hbound_at)[k];
}
// rank of every element of the collection, in collection order, from the slices' local ranks: a workgroup per sketch
// (its slice boundaries and the slices' segment starts sit in LDS; no search per element)
__global__ __launch_bounds__(256) void k_reassemble(const uint64_t* __restrict__ off, uint32_t nsk,
                                                    const uint32_t* __restrict__ spart, uint32_t G, const uint32_t* __restrict__ segoff,
                                                    const uint8_t* __restrict__ gathered, uint64_t share_bytes, uint64_t ranks_at,
                                                    const DictState* __restrict__ ds, uint32_t* __restrict__ rank, uint32_t flags,
                                                    uint16_t* __restrict__ eflag) {
  __shared__ uint32_t sp[65], so[64], rb[64];
  for (uint32_t s = blockIdx.x; s < nsk; s += gridDim.x) {
    __syncthreads();
    if (threadIdx.x <= G) sp[threadIdx.x] = spart[(size_t)s * (G + 1) + threadIdx.x];
    if (threadIdx.x < G) { so[threadIdx.x] = segoff[(size_t)threadIdx.x * (nsk + 1) + s]; rb[threadIdx.x] = ds->rbase[threadIdx.x]; }
    __syncthreads();
    const uint64_t base = off[s];
    const uint32_t len = (uint32_t)(off[s + 1] - base);
    for (uint32_t p = threadIdx.x; p < len; p += 256) {
      uint32_t g = 0;
      while (g + 1 < G && sp[g + 1] <= p) g++;     // last g with sp[g] <= p
      const uint32_t* seg = reinterpret_cast<const uint32_t*>(gathered + (size_t)g * share_bytes + ranks_at);
      const uint32_t w = seg[so[g] + (p - sp[g])];
      rank[base + p] = rb[g] + (flags ? (w & 0x3fffffffu) : w);
      // (what k_claim_bits / k_elem_bits start from: not shared / shared / shared and the first of its run)
      if (eflag) eflag[base + p] = (w >> 31) ? (((w >> 30) & 1u) ? kEbitHead : kEbitShared) : (uint16_t)0xffffu;
    }
  }
}

// ---- components of the "shares a hash" graph ------------------------------------------------
// Two sketches in different connected components have no hash in common: common = 0 and size =
// min(n, |A| + |B|) without looking at them.  The pre-pass below finds the components with a
// lock-free union-find over the runs of the sorted pooled hashes; rows and columns are then
// visited component by component and only tiles that hold same-component pairs are launched
// (the rest of the matrix is filled by k_fill_disjoint).  All-vs-all over unrelated genomes is
// mostly such pairs; a collection that is one big component costs what it did before.
__device__ __forceinline__ uint32_t uf_load(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // past the per-CU cache
}
__device__ __forceinline__ uint32_t uf_find(uint32_t* parent, uint32_t x) {
  uint32_t p = uf_load(parent + x);
  while (p != x) {
    const uint32_t gp = uf_load(parent + p);
    if (gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // halving: gp is an ancestor
    x = p; p = gp;
  }
  return x;
}
__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t x, uint32_t y) {
  while (true) {
    x = uf_find(parent, x); y = uf_find(parent, y);
    if (x == y) return;
    if (x > y) { const uint32_t t = x; x = y; y = t; }
    if (atomicCAS(parent + y, y, x) == y) return;   // the larger root goes under the smaller: parent[v] <= v always
  }
}
__global__ __launch_bounds__(256) void k_uf_init(uint32_t* parent, uint32_t m) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) parent[i] = i;
}
// sorted position i continues the run of i-1 (same hash): their sketches are connected
// Related sketches meet as neighbours in thousands of runs, and every one of those unions ends at
// the same few parent words.  Device-scope loads on this part are served past the per-XCD L2s, so
// they are rationed: a first launch unites a small sample of the neighbour pairs, which already
// connects nearly everything; the later launches (fresh caches) look at more pairs with ORDINARY
// cached loads first -- a stale parent word still names an ancestor, so equal roots in a cached
// view prove the two are connected -- and only the few that are not go to the atomic path.
// Position i-1's sketch and parent are what the lane to the left has just loaded for ITS position (Shift == 0): a lane
// gathers one sketch id and one parent word instead of two of each -- the kernel is bound by the number of scattered lines
// its loads touch (one line per lane and cycle per CU), not by their bytes.
template <int Shift, bool Filter>
__global__ __launch_bounds__(256) void k_uf_runs(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ origin,
                                                 NodeMap node, uint64_t n, uint32_t* parent,
                                                 const uint32_t* __restrict__ runid, const uint8_t* __restrict__ isfreq) {
  const uint64_t i = ((uint64_t)xcd_chunked_block() * blockDim.x + threadIdx.x) << Shift;
  const bool in = i < n;                                  // (whole waves stay for the shuffles)
  const int lane = threadIdx.x & 63;
  const uint64_t k0 = in ? keys[i] : 0ull;
  bool eq = in && i != 0 && keys[i - 1] == k0;
  if (eq && isfreq && isfreq[runid[i]]) eq = false;       // a frequent hash connects nothing (see "frequent hashes" below)
  uint32_t a, b;
  if (Shift == 0) {
    const uint64_t want = __ballot(eq);                   // my own sketch is wanted by me or by the lane to my right
    const bool load = eq || ((want >> 1) >> lane) & 1ull;
    a = load ? node_of(node, origin[i]) : 0u;
    b = (uint32_t)__shfl_up((int)a, 1);
    if (eq && lane == 0) b = node_of(node, origin[i - 1]);
    eq = eq && a != b;
    if (Filter) {
      const uint64_t want2 = __ballot(eq);
      const bool load2 = eq || ((want2 >> 1) >> lane) & 1ull;
      uint32_t pa = load2 ? parent[a] : a;
      uint32_t pb = (uint32_t)__shfl_up((int)pa, 1);
      if (eq) {
        if (lane == 0) pb = parent[b];
        for (int hop = 0; hop < 64 && pa != a; hop++) { a = pa; pa = parent[a]; }
        for (int hop = 0; hop < 64 && pb != b; hop++) { b = pb; pb = parent[b]; }
        if (a == b) eq = false;
      }
    }
  } else {
    if (!eq) return;
    a = node_of(node, origin[i]); b = node_of(node, origin[i - 1]);
    if (a == b) return;
    if (Filter) {
      for (int hop = 0; hop < 64; hop++) { const uint32_t p = parent[a]; if (p == a) break; a = p; }
      for (int hop = 0; hop < 64; hop++) { const uint32_t p = parent[b]; if (p == b) break; b = p; }
      if (a == b) return;
    }
  }
  if (eq) uf_union(parent, a, b);
}
// The first sample of the neighbour pairs for a collection of at most kUfLdsNodes sketches: ONE workgroup, the forest in
// LDS.  The unions of related sketches all meet at the same few parent words; in LDS that contention costs nanoseconds,
// through the L2 (agent-scope atomics) microseconds each (1000 sketches of one family: 93 us for 7 800 sampled pairs).
constexpr uint32_t kUfLdsNodes = 16384;
// Several workgroups: each takes every gridDim.x-th sampled pair, builds a forest of its own in LDS and leaves every
// sketch's root in its row of `out` ([gridDim.x][nsk]); k_uf_merge then unites the forests (nsk unions per workgroup, most
// of them between sketches the earlier forests have connected already).
// (find with path halving, like uf_find: without it a forest that has taken many unions is walked hop by hop -- sixteen
// positions per lane took 15 us per union)
__device__ __forceinline__ uint32_t lds_uf_find(uint32_t* lpar, uint32_t x) {
  uint32_t p = __hip_atomic_load(&lpar[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (p != x) {
    const uint32_t gp = __hip_atomic_load(&lpar[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (gp != p) __hip_atomic_store(&lpar[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // gp is an ancestor of x
    x = p; p = gp;
  }
  return x;
}
template <int Shift>
__global__ __launch_bounds__(1024) void k_uf_runs_lds(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ origin,
                                                      NodeMap node, uint64_t n, uint32_t nsk, uint32_t* __restrict__ out,
                                                      const uint32_t* __restrict__ runid, const uint8_t* __restrict__ isfreq) {
  extern __shared__ uint32_t lpar[];
  for (uint32_t i = threadIdx.x; i < nsk; i += 1024) lpar[i] = i;
  __syncthreads();
  // Eight sampled positions per lane and turn, every load of a step issued for all eight before the first is used: a
  // position is a chain of three dependent loads (keys, origin, sketch), and one workgroup has to cover their latency itself
  // (a lane's turns -- two at ~2 K positions per forest -- then cost as much as one).
  const uint64_t stride = (uint64_t)gridDim.x * 1024;
  for (uint64_t t0 = (uint64_t)blockIdx.x * 1024 + threadIdx.x + 1; (t0 << Shift) < n; t0 += stride * 8) {
    uint64_t idx[8], k0[8], k1[8];
    bool eq[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      idx[j] = (t0 + (uint64_t)j * stride) << Shift;
      eq[j] = idx[j] < n;
      if (!eq[j]) idx[j] = 1;                             // (t0 >= 1 and (t0 << Shift) < n: position 1 exists)
      k0[j] = keys[idx[j]]; k1[j] = keys[idx[j] - 1];
    }
    uint32_t rid[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { eq[j] = eq[j] && k0[j] == k1[j]; rid[j] = (isfreq && eq[j]) ? runid[idx[j]] : 0u; }
    if (isfreq) {
#pragma unroll
      for (int j = 0; j < 8; j++) if (eq[j] && isfreq[rid[j]]) eq[j] = false;
    }
    uint32_t oa[8], ob[8], xs[8], ys[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { oa[j] = eq[j] ? origin[idx[j]] : 0u; ob[j] = eq[j] ? origin[idx[j] - 1] : 0u; }
#pragma unroll
    for (int j = 0; j < 8; j++) { xs[j] = eq[j] ? node_of(node, oa[j]) : 0u; ys[j] = eq[j] ? node_of(node, ob[j]) : 0u; }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (!eq[j]) continue;
      uint32_t x = xs[j], y = ys[j];
      while (true) {
        // (relaxed workgroup-scope loads: other lanes are changing the forest)
        x = lds_uf_find(lpar, x); y = lds_uf_find(lpar, y);
        if (x == y) break;
        if (x > y) { const uint32_t sw = x; x = y; y = sw; }
        if (atomicCAS(&lpar[y], y, x) == y) break;      // the larger root goes under the smaller
      }
    }
  }
  __syncthreads();
  uint32_t* dst = out + (size_t)blockIdx.x * nsk;
  for (uint32_t i = threadIdx.x; i < nsk; i += 1024) {   // flattened: every sketch points at its root
    uint32_t x = i;
    while (true) { const uint32_t p = lpar[x]; if (p == x) break; x = p; }
    dst[i] = x;
  }
}
// the slices' forests (root of every sketch within slice g) united into one
__global__ __launch_bounds__(256) void k_uf_merge(const uint8_t* __restrict__ gathered, uint64_t share_bytes, uint64_t roots_at,
                                                  uint32_t G, uint32_t nsk, uint32_t* parent) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)G * nsk) return;
  const uint32_t g = (uint32_t)(t / nsk), i = (uint32_t)(t % nsk);
  const uint32_t r = reinterpret_cast<const uint32_t*>(gathered + (size_t)g * share_bytes + roots_at)[i];
  if (r == i) return;
  // the forest before this one names the same root for i: its lane does this very union (related sketches have the same
  // root -- their smallest index -- in nearly every forest, and all those unions would queue at that root's word: 37
  // forests of one family of 10 000 took 200 us here)
  if (g && reinterpret_cast<const uint32_t*>(gathered + (size_t)(g - 1) * share_bytes + roots_at)[i] == r) return;
  // an ordinary (cached) look first: a stale parent word still names an ancestor, so equal roots in this view prove the two
  // are connected already -- by a forest whose unions ran earlier in this launch
  uint32_t a = i, b = r;
  for (int hop = 0; hop < 64; hop++) { const uint32_t p = parent[a]; if (p == a) break; a = p; }
  for (int hop = 0; hop < 64; hop++) { const uint32_t p = parent[b]; if (p == b) break; b = p; }
  if (a == b) return;
  uf_union(parent, a, b);
}
// The same in LDS, two levels deep (at most kUfLdsNodes sketches): workgroup b unites rows [8 b, 8 b + 8) in a forest of its
// own -- a lane takes sketch i through its eight rows one after the other, the eight roots requested together -- and leaves
// every sketch's root in row b of `out`; a second launch of ONE workgroup does the same with those rows and leaves the roots,
// flattened, in parent[] (and root[]).  The unions of related sketches meet at a few words: nanoseconds apart in LDS,
// a queue at the L2 as device-scope atomics (37 forests of 10 000 sketches: 125-135 us with k_uf_merge; everything in ONE
// workgroup: 286 us -- 370 K unions through one CU's LDS chains).
__global__ __launch_bounds__(1024) void k_uf_merge_lds(const uint8_t* __restrict__ rows, uint64_t row_bytes, uint64_t roots_at, uint32_t G,
                                                       uint32_t nsk, uint32_t* __restrict__ out, uint32_t* __restrict__ root) {
  extern __shared__ uint32_t lpar[];
  for (uint32_t i = threadIdx.x; i < nsk; i += 1024) lpar[i] = i;
  __syncthreads();
  const uint32_t g0 = blockIdx.x * 8u, g1 = min(g0 + 8u, G);
  for (uint32_t i = threadIdx.x; i < nsk; i += 1024) {
    uint32_t r[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++)
      r[k] = g0 + k < g1 ? reinterpret_cast<const uint32_t*>(rows + (size_t)(g0 + k) * row_bytes + roots_at)[i] : i;
    uint32_t prev = i;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t v = r[k];
      if (v == i || v == prev) continue;                   // (nothing to unite, or the row before said the same)
      prev = v;
      uint32_t x = i, y = v;
      while (true) {
        x = lds_uf_find(lpar, x); y = lds_uf_find(lpar, y);
        if (x == y) break;
        if (x > y) { const uint32_t t = x; x = y; y = t; }
        if (atomicCAS(&lpar[y], y, x) == y) break;          // the larger root goes under the smaller
      }
    }
  }
  __syncthreads();
  uint32_t* dst = out + (size_t)blockIdx.x * nsk;
  for (uint32_t i = threadIdx.x; i < nsk; i += 1024) {
    uint32_t x = i;
    while (true) { const uint32_t p = lpar[x]; if (p == x) break; x = p; }
    dst[i] = x;
    if (root) root[i] = x;
  }
}
__global__ __launch_bounds__(256) void k_uf_roots(uint32_t* parent, uint32_t m, uint32_t* __restrict__ root) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) root[i] = uf_find(parent, i);
}
// ---- frequent hashes --------------------------------------------------------------------------------
// One hash held by every sketch (a contaminant, an adapter k-mer) makes one component out of unrelated
// genomes, and every pair would have to be walked.  Hashes whose run in the sorted pool is longer than a
// share of the sketches are therefore set aside (at most kMaxFreq of them, else none): they do not unite
// anything in the union-find, so the components are the clusters of everything else.  Two sketches in
// different clusters can then share ONLY frequent hashes, and their pair follows from two small per-sketch
// records -- a bit mask of the frequent hashes the sketch holds and the position of each inside it: for the
// k-th shared one (ascending), the union holds posA + posB - k smaller elements, which decides whether it
// lies inside the first n of the union (reference src/lib.rs:470-499); |A u B| = |A| + |B| - shared.
// Exact for any threshold: the threshold only moves work between the walk and this rule.
__global__ __launch_bounds__(256) void k_freq_mark(const uint32_t* __restrict__ starts, uint32_t n, uint32_t threshold,
                                                   RangeState* st) {
  const uint32_t nruns = st->nruns;
  const int lane = threadIdx.x & 63;
  for (uint32_t r0 = blockIdx.x * blockDim.x; r0 < nruns; r0 += gridDim.x * blockDim.x) {
    const uint32_t r = r0 + threadIdx.x;
    const bool hit = r < nruns && ((r + 1 < nruns ? starts[r + 1] : n) - starts[r]) > threshold;
    // one atomic per wave: in a dense family thousands of runs are long, and they would all queue at one counter word
    const uint64_t m = __ballot(hit);
    if (m == 0) continue;
    // more than kMaxFreq already: nothing will be set aside, the exact count is of no interest (a plain L2 read, unlike
    // the read-modify-writes, which are served one at a time: 4 000 long runs of a dense family cost 20 us there)
    if (__hip_atomic_load(&st->nfreq_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > kMaxFreq) continue;
    uint32_t base = 0;
    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&st->nfreq_seen, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(m));
    if (hit) {
      const uint32_t k = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (k < kMaxFreq) st->freq_run[k] = r;
    }
  }
}
// one wave: sort the (at most 64) run ids, decide, mark
// (cap: this slice's share of the kMaxFreq hashes that can be set aside in all)
__global__ __launch_bounds__(64) void k_freq_finalize(RangeState* st, uint8_t* __restrict__ isfreq, uint32_t cap) {
  const uint32_t seen = st->nfreq_seen, lane = threadIdx.x;
  if (seen == 0 || seen > cap) { if (lane == 0) st->nfreq = 0; return; }
  const uint32_t mine = lane < seen ? st->freq_run[lane] : 0xffffffffu;
  uint32_t rank = 0;
  for (uint32_t k = 0; k < seen; k++) rank += (uint32_t)__shfl((int)mine, (int)k) < mine ? 1u : 0u;   // run ids are distinct
  if (lane < seen) { st->freq_run[rank] = mine; isfreq[mine] = (uint8_t)(rank + 1); }
  if (lane == 0) st->nfreq = seen;
}
// mask / position records of every sketch: which of the frequent hashes it holds, and where (a binary search per hash)
__global__ __launch_bounds__(256) void k_freq_records(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ off, uint32_t nsk,
                                                      const DictState* __restrict__ ds, unsigned long long* __restrict__ mask,
                                                      uint32_t* __restrict__ pos) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsk) return;
  const uint32_t nfreq = ds->nfreq;
  const uint64_t* v = hashes + off[s];
  const uint32_t len = (uint32_t)(off[s + 1] - off[s]);
  unsigned long long m = 0;
  uint32_t lo = 0;                     // the frequent hashes ascend: each search starts where the last one ended
  for (uint32_t b = 0; b < nfreq; b++) {
    const uint64_t f = ds->freq_hash[b];
    uint32_t hi = len;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (v[mid] < f) lo = mid + 1; else hi = mid;
    }
    if (lo < len && v[lo] == f) {
      m |= 1ull << b;
      pos[(size_t)b * nsk + s] = lo;                     // ([frequent hash][sketch]: the fill reads a hash's positions by column)
    }
  }
  mask[s] = m;
}

// every pair as if it shared nothing but frequent hashes (none, usually); the compare kernels then
// overwrite the pairs they walk
// (a column per lane, eight rows per workgroup row: no division to find the pair, the column's length and record read once,
// no floating-point division for a pair that shares nothing -- one thread per pair with pid / ncols, pid % ncols and a
// division by the size was bound by its arithmetic, not by the 8 bytes it writes: 0.30 ms for 10 000 x 10 000)
constexpr uint32_t kFillRows = 8;
__global__ __launch_bounds__(256) void k_fill_disjoint(const uint64_t* __restrict__ roff, uint32_t nrows,
                                                       const uint64_t* __restrict__ coff, uint32_t ncols, uint32_t num,
                                                       const uint32_t* __restrict__ row_nums, CompareOut out,
                                                       const unsigned long long* __restrict__ rmask, const uint32_t* __restrict__ rpos,
                                                       const unsigned long long* __restrict__ cmask, const uint32_t* __restrict__ cpos,
                                                       uint32_t pos_stride) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncols) return;
  const uint64_t lb = coff[j + 1] - coff[j];
  const unsigned long long cm = rmask ? cmask[j] : 0ull;
  const uint32_t* pb = rmask ? cpos + j : nullptr;          // (position of frequent hash b: pb[b * pos_stride])
  for (uint32_t i0 = blockIdx.y * kFillRows; i0 < nrows; i0 += gridDim.y * kFillRows) {
    const uint32_t i1 = min(i0 + kFillRows, nrows);
    for (uint32_t i = i0; i < i1; i++) {
      const uint64_t pid = (uint64_t)i * ncols + j;
      const uint64_t la = roff[i + 1] - roff[i];
      const uint64_t n = row_nums ? row_nums[i] : num;
      uint64_t cc = 0, common = 0;
      if (rmask) {
        unsigned long long m = rmask[i] & cm;
        const uint32_t* pa = rpos + i;
        while (m) {
          const int b = __ffsll((long long)m) - 1;
          m &= m - 1;
          const uint64_t u = (uint64_t)pa[(size_t)b * pos_stride] + pb[(size_t)b * pos_stride] - cc;   // union elements smaller than this shared hash
          if (n == 0 || u < n) common++;
          cc++;
        }
      }
      const uint64_t tot = la + lb - cc;
      const uint64_t size = (n != 0 && tot > n) ? n : tot;
      if (out.common) out.common[pid] = common;
      if (out.size) out.size[pid] = size;
      if (out.jaccard) out.jaccard[pid] = common ? (double)common / (double)size : 0.0;
      if (out.count_common) out.count_common[pid] = cc;
      if (out.containment) out.containment[pid] = (cc == 0 && la != 0) ? 0.0 : (double)cc / (double)la;   // (0 / 0 stays what it was)
    }
  }
}

// ---- device-side plan ----------------------------------------------------------------------------
// Everything between the union-find and the compare kernels is decided on the device, so that a
// block compare has ONE host synchronisation, at its end:
//   k_plan_keys      (component << 32 | sketch) keys, sorted: slot order = sketches of a component adjacent
//   k_plan_ranges    per row slot the column slots of its component (and vice versa); pairs = sum
//   k_plan_route     few sharing pairs -> per-component pair kernel, else the tiled kernel
//   k_comp_count / scan / k_comp_fill     the pair kernel's work list
//   k_tiles_count16, k_plan_geometry      how many 16-row tiles hold sharing pairs -> rows per tile
//   k_flag_tiles     the tile list of that geometry (ordered inside 256-tile chunks)
// The compare kernels are launched unconditionally with persistent grids and return at once when the
// plan did not pick them; the tiled pre-pass (ranks, ranges, partition table) is skipped the same way.
__global__ __launch_bounds__(256) void k_plan_keys(const uint32_t* __restrict__ root, uint32_t n, uint64_t* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = ((uint64_t)root[i] << 32) | i;
}
// first slot of `keys` whose component is >= comp
__device__ __forceinline__ uint32_t comp_lower_bound(const uint64_t* __restrict__ keys, uint32_t n, uint64_t comp) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((keys[mid] >> 32) < comp) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// other_lo/hi[i] = slots of `other` that hold the component of slot i of `mine`
__global__ __launch_bounds__(256) void k_plan_ranges(const uint64_t* __restrict__ mine, uint32_t n_mine,
                                                     const uint64_t* __restrict__ other, uint32_t n_other,
                                                     uint32_t* __restrict__ other_lo, uint32_t* __restrict__ other_hi,
                                                     PlanState* st) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t mine_pairs = 0;
  if (i < n_mine) {
    const uint64_t comp = mine[i] >> 32;
    const uint32_t lo = comp_lower_bound(other, n_other, comp), hi = comp_lower_bound(other, n_other, comp + 1);
    other_lo[i] = lo; other_hi[i] = hi;
    mine_pairs = hi - lo;
  }
  if (st) {
    const uint64_t wsum = wave_sum64(mine_pairs);
    if ((threadIdx.x & 63) == 0 && wsum) atomicAdd(&st->pairs, (unsigned long long)wsum);
  }
}
__global__ void k_plan_route(PlanState* st, uint32_t forced_route, uint32_t visit_all, unsigned long long comp_pairs_limit) {
  uint32_t route = forced_route;
  if (route != kRouteComponents && route != kRouteTiled)
    route = (st->pairs <= comp_pairs_limit && !visit_all) ? (uint32_t)kRouteComponents : (uint32_t)kRouteTiled;
  st->route = route;
  st->skip_tiled = route != kRouteTiled;
  st->skip_comp = route != kRouteComponents;
}
// work items of the per-component kernel: (column, <= 32 rows of its component)
constexpr uint32_t kRowsPerItem = 32;
__device__ __forceinline__ uint32_t comp_items_of(uint32_t c, const uint32_t* row_lo, const uint32_t* row_hi, uint32_t symmetric,
                                                  uint32_t* first_row) {
  const uint32_t rs = symmetric ? max(row_lo[c], c) : row_lo[c];   // same order on both axes: upper triangle only
  *first_row = rs;
  return row_hi[c] > rs ? (row_hi[c] - rs + kRowsPerItem - 1) / kRowsPerItem : 0u;
}
__global__ __launch_bounds__(256) void k_comp_count(uint32_t ncols, const uint32_t* __restrict__ row_lo,
                                                    const uint32_t* __restrict__ row_hi, uint32_t symmetric,
                                                    uint32_t* __restrict__ cnt, const PlanState* __restrict__ st) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  uint32_t rs;
  cnt[c] = st->skip_comp ? 0u : comp_items_of(c, row_lo, row_hi, symmetric, &rs);
}
__global__ __launch_bounds__(256) void k_comp_fill(const uint64_t* __restrict__ ckey, uint32_t ncols,
                                                   const uint32_t* __restrict__ row_lo, const uint32_t* __restrict__ row_hi,
                                                   uint32_t symmetric, const uint32_t* __restrict__ off, CompWork* __restrict__ work,
                                                   uint32_t work_cap, const PlanState* __restrict__ st) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols || st->skip_comp) return;
  uint32_t rs;
  const uint32_t n = comp_items_of(c, row_lo, row_hi, symmetric, &rs);
  const uint32_t col = (uint32_t)ckey[c];
  for (uint32_t k = 0; k < n; k++)
    if (off[c] + k < work_cap) work[off[c] + k] = CompWork{col, rs + k * kRowsPerItem, min(row_hi[c], rs + (k + 1) * kRowsPerItem)};
}
// does the tr x 64 tile (ti, tj) of the slot orders hold a pair of one component (that this launch is responsible for)?
struct TileTest {
  uint32_t nrows, ncols;
  const uint32_t* col_lo; const uint32_t* col_hi;   // per row slot: the column slots of its component
  const uint64_t* rkey; const uint64_t* ckey;       // slot -> (component << 32 | local index)
  PairScope sc;
  uint32_t all_on;
};
__device__ __forceinline__ bool tile_shares(uint32_t ti, uint32_t tj, uint32_t tr, const TileTest& t) {
  // own_mode 1: a tile wholly below the diagonal of slot space -- its pairs are written as mirrors of the tile above
  if (t.sc.own_mode == 1 && (uint64_t)tj * kTB + kTB - 1 < (uint64_t)ti * tr) return false;
  if (t.all_on) return true;
  const uint32_t cbeg = tj * kTB, cend = min(cbeg + (uint32_t)kTB, t.ncols);
  const uint32_t r1 = min((ti + 1) * tr, t.nrows);
  for (uint32_t i = ti * tr; i < r1; i++) {
    const uint32_t c0 = max(t.col_lo[i], cbeg), c1 = min(t.col_hi[i], cend);
    if (c0 >= c1) continue;
    if (t.sc.own_mode != 2) return true;
    // own_mode 2: columns of one component ascend with the slot, so [c0, c1) spans the indices [ja, jb]; row i owns
    // the circular interval [i, i + N/2].  Interval overlap is a superset test (a tile flagged in vain only costs time).
    const uint32_t gi = t.sc.row_base + (uint32_t)t.rkey[i];
    const uint32_t ja = (uint32_t)t.ckey[c0], jb = (uint32_t)t.ckey[c1 - 1];
    const uint64_t top = (uint64_t)gi + t.sc.ntotal / 2;
    if (jb >= gi && ja <= top) return true;
    if (top >= t.sc.ntotal && ja <= top - t.sc.ntotal) return true;
  }
  return false;
}
__global__ __launch_bounds__(256) void k_tiles_count16(TileTest t, PlanState* st) {
  if (st->skip_tiled) return;
  const uint32_t tiles_r = (t.nrows + 15) / 16, tiles_c = (t.ncols + kTB - 1) / kTB;
  const uint64_t all = (uint64_t)tiles_r * tiles_c;
  uint32_t mine = 0;
  for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < all; x += (uint64_t)gridDim.x * blockDim.x)
    mine += tile_shares((uint32_t)(x / tiles_c), (uint32_t)(x % tiles_c), 16, t) ? 1u : 0u;
  const uint32_t wsum = (uint32_t)wave_sum64(mine);
  if ((threadIdx.x & 63) == 0 && wsum) atomicAdd(&st->count16, wsum);
}
// The height of the tiles, from the number of 16-row tiles that hold sharing pairs (tools/sweep_pf*.sh,
// profiles/r03_tile_shape.txt; one family, every pair walked).  One tile is one latency chain, and a wave walks its rows
// one after the other:
//   * fewer than ~0.4 rounds of the chip's workgroup slots: 8 rows (8 waves x 1 row) keep more of the chip busy
//     (1000 x 1000: 0.68 ms against 0.79 with 16 rows);
//   * up to ~2400: 16 rows (8 waves x 2 rows; 1600 x 1600 1.05 against 1.12 / 1.24 with 8 / 32 rows);
//   * more: 32 rows (8 waves x 4 rows) -- four rows per wave between barriers and one column stage for 32 rows amortise the
//     per-stretch work best (10 000 x 10 000: 25.0 ms against 29.5 with 16 rows).
// All by the pipelined kernel; the plain one (k_compare_tiled, 16 rows by 4 waves x 4 rows: 26.4 ms at 10 000 x 10 000, and
// behind at every smaller size) is kept for the experiments build.  rpw x 4 = rows per tile.
struct TileShape { uint32_t rpw, pf; };
__device__ __host__ inline TileShape tile_shape_for(uint64_t count16, uint32_t fill_tiles) {
  if (10 * count16 < fill_tiles) return {2u, 1u};
  if (17 * count16 < 5ull * fill_tiles) return {4u, 1u};
  return {8u, 1u};
}
__global__ void k_plan_geometry(PlanState* st, uint32_t forced_rpw, uint32_t forced_pf, uint32_t fill_tiles, uint32_t* part_built) {
  if (!st->skip_tiled) *part_built = 1;   // k_partition ran just before this launch (same stream)
  TileShape sh = {forced_rpw, forced_pf};
  if (!forced_rpw) sh = tile_shape_for(st->count16, fill_tiles);
  st->rpw = sh.rpw; st->pf = sh.pf;
}
__global__ __launch_bounds__(256) void k_flag_tiles(TileTest t, uint32_t wpb, uint32_t* __restrict__ tiles, uint32_t tiles_cap,
                                                    PlanState* st) {
  __shared__ uint32_t wcnt[4], base_s;
  if (st->skip_tiled) return;
  const uint32_t tr = st->rpw * wpb;
  const uint32_t tiles_r = (t.nrows + tr - 1) / tr, tiles_c = (t.ncols + kTB - 1) / kTB;
  const uint64_t all = (uint64_t)tiles_r * tiles_c;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // chunks of 256 consecutive tiles; inside a chunk the list keeps tile order
  for (uint64_t c0 = (uint64_t)blockIdx.x * 256; c0 < all; c0 += (uint64_t)gridDim.x * 256) {
    const uint64_t x = c0 + threadIdx.x;
    uint32_t ti = 0, tj = 0;
    bool on = false;
    if (x < all) {
      ti = (uint32_t)(x / tiles_c); tj = (uint32_t)(x % tiles_c);
      on = tile_shares(ti, tj, tr, t);
    }
    const uint64_t m = __ballot(on);
    if (lane == 0) wcnt[w] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      base_s = tot ? atomicAdd(&st->ntiles, tot) : 0u;
    }
    __syncthreads();
    if (on) {
      uint32_t pos = base_s + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      for (int k = 0; k < w; k++) pos += wcnt[k];
      if (pos < tiles_cap) { tiles[2 * pos] = ti; tiles[2 * pos + 1] = tj; }
    }
    __syncthreads();
  }
}

// temporaries of building a dictionary and of planning one block compare: process-wide, grow-only, used under the device mutex
struct TiledScratch {
  DeviceBuffer keys0, keys1, org0, org1, uniq, starts, node, parent, tiles, work, plan, pk0, pk1, pk2, pk3, rng, cnt, runid, isfreq,
      sample0, sample1, rstate, cat, wroots, ties, mv0, mv1;
};
TiledScratch& tiled_scratch() {
  static TiledScratch* t = new TiledScratch();
  return *t;
}

}  // namespace

// tuning (which kernel serves a block; never a result) and the record of the last block
// (process-wide; block compares run under the device mutex, these accessors take a lock of their own)
static std::mutex g_tune_mu;
static CompareTuning g_tuning;
static CompareStats g_stats;
void compare_set_tuning(const CompareTuning& t) { std::lock_guard<std::mutex> l(g_tune_mu); g_tuning = t; }
CompareTuning compare_get_tuning() { std::lock_guard<std::mutex> l(g_tune_mu); return g_tuning; }
CompareStats compare_last_stats() { std::lock_guard<std::mutex> l(g_tune_mu); return g_stats; }
static void set_stats(const CompareStats& st) { std::lock_guard<std::mutex> l(g_tune_mu); g_stats = st; }

// Geometry experiments (tools/): compiled in only with -DSMH_EXPERIMENTS, read once.
struct TiledExperiments {
  uint32_t per_range = 24;            // pooled elements per sketch per range
  uint32_t capA = 1024, capB = 48;    // LDS dwords of the row pool; column elements per range
  int rpw = 0, wpb = 4, minw = 8;     // rpw 0 = chosen by the plan
  bool pf = false;                    // with rpw: the pipelined kernel
  bool xcd = true;
  bool no_masks = false;              // the walk from the first range on (A/B of the range masks)
};
static const TiledExperiments& tiled_experiments() {
  static const TiledExperiments ex = [] {
    TiledExperiments e;
#ifdef SMH_EXPERIMENTS
    if (const char* v = std::getenv("SOURMASH_AMD_CMP_PER_RANGE")) e.per_range = (uint32_t)std::max(4, std::atoi(v));
    if (const char* v = std::getenv("SOURMASH_AMD_CMP_LDS")) {
      int ca = 0, cb = 0;
      if (sscanf(v, "%d,%d", &ca, &cb) == 2 && ca >= 256 && cb >= 8) { e.capA = (uint32_t)ca; e.capB = (uint32_t)cb; }
    }
    if (const char* v = std::getenv("SOURMASH_AMD_CMP_GEO")) sscanf(v, "%d,%d,%d", &e.rpw, &e.wpb, &e.minw);
    if (const char* v = std::getenv("SOURMASH_AMD_CMP_PF")) e.pf = std::atoi(v) != 0;
    if (std::getenv("SOURMASH_AMD_CMP_NO_XCD")) e.xcd = false;
    if (std::getenv("SOURMASH_AMD_CMP_NO_MASKS")) e.no_masks = true;
#endif
    return e;
  }();
  return ex;
}

static void release_implicit_dict();
void release_compare_scratch() {
  TiledScratch& T = tiled_scratch();
  for (DeviceBuffer* b : {&T.keys0, &T.keys1, &T.org0, &T.org1, &T.uniq, &T.starts, &T.node, &T.parent, &T.tiles, &T.work, &T.plan,
                          &T.pk0, &T.pk1, &T.pk2, &T.pk3, &T.rng, &T.cnt, &T.runid, &T.isfreq, &T.sample0, &T.sample1, &T.rstate,
                          &T.cat, &T.wroots, &T.ties, &T.mv0, &T.mv1})
    b->release();
  release_implicit_dict();
}

// byte passes of a radix sort that can differ among keys (component << 32 | index) with
// component < M and index < n
static uint32_t plan_key_passes(uint32_t M, uint32_t n) {
  uint32_t mask = 0;
  for (int b = 0; b < 4; b++) {
    if (b == 0 || (n - 1) >> (8 * b)) mask |= 1u << b;
    if (b == 0 || (M - 1) >> (8 * b)) mask |= 1u << (4 + b);
  }
  return mask;
}

// ---- the dictionary of one collection ------------------------------------------------------------
// Everything a block compare needs to know about the collection as a whole, built once (by one owner, or by `world`
// cooperating owners with one all-gather between collection_begin and collection_finish) and then used by any number of
// collection_compare calls over sub-blocks of it:
//   rank[total]        dense order-preserving u32 rank of every hash (the tiled kernel compares ranks)
//   root[n]            component of every sketch in the "shares a (non-frequent) hash" graph
//   fmask / fpos       which of the frequent hashes a sketch holds, and where
//   part[n][R+1]       where every sketch crosses the R range boundaries of the tiled kernel
struct CollectionDict {
  const uint64_t* hashes = nullptr;   // not owned: collection element t is hashes[t]
  uint32_t n = 0, world = 1, rank = 0, max_len = 0;
  uint64_t total = 0;
  uint32_t Rg = 1, R = 1;             // ranges of the tiled kernel per slice / in all
  uint32_t n_mine = 0, n_max = 0;     // elements of this owner's slice / of the largest slice
  uint64_t roots_at = 0, hbound_at = 0, ranks_at = 0, share_bytes = 0;
  bool finished = false, split = false;
  bool force_radix = false;           // the pooled sort with all eight passes (the four-pass sort + tie fix gave up on this collection, or the tuning asks)
  // where compare reads them: buffers of their own, or -- one owner -- straight inside the share
  const uint32_t* rank_ptr = nullptr;
  const uint32_t* root_ptr = nullptr;
  const uint64_t* hbound_ptr = nullptr;
  DeviceBuffer off, splitters, spart, segoff, share, dstate, rankv, root, fmask, fpos, hbound, part;
  // range masks (one owner): shared number of every element, shared hashes below every range boundary, word offsets,
  // what k_mask_layout decided, the masks and the crossings range-major (built with the partition table)
  DeviceBuffer sid, sb, woff, minfo, masks, partT;
  uint32_t mask_words_max = 0;
  bool has_masks = false;             // one owner: built with the dictionary
  bool lazy_tried = false;            // several owners: built by the first block compare that may walk tiles (k_claim_bits ...)
  uint32_t uniform_len = 0;           // the sketches' common length (0: they differ)
  bool lazy_ready = false;            // several owners: the elements' states are in sid (k_reassemble): masks can be built on demand
  bool share_flags = false;           // several owners: the shares' ranks carry "shared" / "first of its run" in bits 31 / 30
  std::vector<uint64_t> rel_off;
};

static inline uint64_t align8(uint64_t x) { return (x + 7) & ~7ull; }

// unites G rows of per-sketch roots ([g] at rows + g * row_bytes + roots_at) into parent[] (flattened) and, if given, root[]
static void uf_merge_rows_lds(const uint8_t* rows, uint64_t row_bytes, uint64_t roots_at, uint32_t G, uint32_t n, TiledScratch& T,
                              uint32_t* parent, uint32_t* root, hipStream_t s) {
  if (G <= 8) {
    hipLaunchKernelGGL(k_uf_merge_lds, dim3(1), dim3(1024), (size_t)n * 4, s, rows, row_bytes, roots_at, G, n, parent, root);
    return;
  }
  const uint32_t L1 = (G + 7) / 8;                          // (G <= 64: at most 8 rows for the second level)
  T.mv0.ensure((size_t)L1 * n * 4);
  hipLaunchKernelGGL(k_uf_merge_lds, dim3(L1), dim3(1024), (size_t)n * 4, s, rows, row_bytes, roots_at, G, n, T.mv0.as<uint32_t>(),
                     (uint32_t*)nullptr);
  hipLaunchKernelGGL(k_uf_merge_lds, dim3(1), dim3(1024), (size_t)n * 4, s, T.mv0.as<uint8_t>(), (uint64_t)n * 4, (uint64_t)0, L1, n, parent,
                     root);
}
static void collection_begin_into(CollectionDict& D, const uint64_t* hashes_dev, const uint64_t* offsets_dev, const uint64_t* offsets_host,
                                  uint32_t n, uint32_t world, uint32_t rank, Device& dev, hipStream_t s, bool full_sort = false) {
  // (the four-pass sort is tried afresh for every collection; a rebuild, or the tuning, asks for all eight passes)
  D.force_radix = full_sort || compare_get_tuning().dictionary == 1;
  if (world == 0 || world > 64 || rank >= world) throw_internal("collection: world must be 1..64 and rank < world");
  if (n == 0) throw_internal("collection: no sketches");
  D.finished = false;
  D.max_len = 0;
  TiledScratch& T = tiled_scratch();
  const TiledExperiments& ex = tiled_experiments();
  const uint64_t base = offsets_host[0];
  D.hashes = hashes_dev + base; D.n = n; D.world = world; D.rank = rank;
  D.total = offsets_host[n] - base;
  if (D.total >= (1ull << 31) - 2) throw_internal("compare block: more than 2^31 hashes");   // (ranks stay below the sentinels)
  for (uint32_t i = 0; i < n; i++) D.max_len = std::max<uint32_t>(D.max_len, (uint32_t)(offsets_host[i + 1] - offsets_host[i]));
  D.uniform_len = D.max_len;
  for (uint32_t i = 0; i < n; i++)
    if (offsets_host[i + 1] - offsets_host[i] != D.max_len) { D.uniform_len = 0; break; }
  // offsets relative to the first element (a copy the dictionary owns: the caller's array may be reused)
  D.off.ensure((size_t)(n + 1) * 8);
  if (offsets_host != D.rel_off.data()) {
    D.rel_off.resize((size_t)n + 1);      // (a member: the copy may read it after this function has returned; a rebuild reads it)
    for (uint32_t i = 0; i <= n; i++) D.rel_off[i] = offsets_host[i] - base;
  }
  if (offsets_dev && base == 0) HIP_CHECK(hipMemcpyAsync(D.off.ptr, offsets_dev, (size_t)(n + 1) * 8, hipMemcpyDeviceToDevice, s));
  else HIP_CHECK(hipMemcpyAsync(D.off.ptr, D.rel_off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
  const uint64_t* off = D.off.as<uint64_t>();
  const uint32_t G = world;
  // ranges of the tiled kernel: about 24 pooled elements per sketch per range, granularity from the LONGEST sketch
  // (its segments must fit the LDS stage); tiles of shorter sketches walk several ranges per step
  uint32_t R0 = (uint32_t)(((uint64_t)D.max_len + ex.per_range - 1) / ex.per_range);
  R0 = std::min<uint32_t>(std::max<uint32_t>(R0, 1), 8192);
  D.Rg = (R0 + G - 1) / G; D.R = D.Rg * G;

  // ---- slices of hash space: splitters from a sorted sample, where every sketch crosses them, slice sizes
  D.splitters.ensure((size_t)G * 8);
  const uint32_t S = G == 1 ? 0u : (uint32_t)std::min<uint64_t>(D.total, 4096);
  if (S) {
    T.sample0.ensure((size_t)S * 8); T.sample1.ensure((size_t)S * 8);
    hipLaunchKernelGGL(k_sample_keys, dim3((S + 255) / 256), dim3(256), 0, s, D.hashes, D.total, S, T.sample0.as<uint64_t>());
    const int cur = radix_sort_u64_keys(T.sample0.as<uint64_t>(), T.sample1.as<uint64_t>(), S, dev.scratch, s, 0xffu);
    hipLaunchKernelGGL(k_pick_splitters, dim3(1), dim3(64), 0, s, cur ? T.sample1.as<uint64_t>() : T.sample0.as<uint64_t>(), S, G,
                       D.splitters.as<uint64_t>());
  } else {
    HIP_CHECK(hipMemsetAsync(D.splitters.ptr, 0, (size_t)G * 8, s));
  }
  if (G == 1) {
    D.n_mine = D.n_max = (uint32_t)D.total;      // one slice: the collection in its own order, no slice tables
  } else {
    D.spart.ensure((size_t)n * (G + 1) * 4);
    D.segoff.ensure((size_t)G * (n + 1) * 4);
    hipLaunchKernelGGL(k_slice_parts, dim3((unsigned)(((uint64_t)n * (G + 1) + 255) / 256)), dim3(256), 0, s, D.hashes, off, n,
                       D.splitters.as<uint64_t>(), G, D.spart.as<uint32_t>());
    T.rstate.ensure(sizeof(RangeState) + 64 * 4);
    uint32_t* d_sizes = reinterpret_cast<uint32_t*>(T.rstate.as<uint8_t>() + sizeof(RangeState));
    hipLaunchKernelGGL(k_slice_scan, dim3(G), dim3(1024), 0, s, D.spart.as<uint32_t>(), n, G, D.segoff.as<uint32_t>(), d_sizes);
    HIP_CHECK(hipGetLastError());
    // the one read-back of building a shared dictionary: the slice sizes (every owner computes the same table)
    std::vector<uint32_t> sizes(G);
    HIP_CHECK(hipMemcpyAsync(sizes.data(), d_sizes, (size_t)G * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    D.n_mine = sizes[rank];
    D.n_max = *std::max_element(sizes.begin(), sizes.end());
  }
  const uint32_t nm = D.n_mine;
  D.roots_at = align8(sizeof(SliceHeader));
  D.hbound_at = align8(D.roots_at + (uint64_t)n * 4);
  D.ranks_at = align8(D.hbound_at + (uint64_t)D.Rg * 8);
  D.share_bytes = align8(D.ranks_at + (uint64_t)D.n_max * 4);
  D.share.ensure(D.share_bytes + 16);      // (+16: the tiled kernel reads the ranks four at a time from any position)
  uint8_t* share = D.share.as<uint8_t>();
  T.rstate.ensure(sizeof(RangeState) + 64 * 4);
  RangeState* rs = T.rstate.as<RangeState>();
  HIP_CHECK(hipMemsetAsync(rs, 0, sizeof(RangeState), s));

  // ---- my slice: gather, sort (hash, place), runs of equal hashes = local dense ranks + document frequencies
  const size_t ne = std::max<uint32_t>(nm, 1);
  T.keys0.ensure(ne * 8); T.keys1.ensure(ne * 8); T.org0.ensure(ne * 4); T.org1.ensure(ne * 4); T.node.ensure(ne * 4);
  T.uniq.ensure(ne * 8); T.starts.ensure((ne + 1) * 4); T.runid.ensure(ne * 4); T.parent.ensure((size_t)n * 4);
  uint64_t* sk = T.keys0.as<uint64_t>();
  uint32_t* so = T.org0.as<uint32_t>();
  D.split = compare_get_tuning().split_frequent != 0 && n >= 32;
  const uint32_t threshold = std::max<uint32_t>(16u, n / 4);
  // one owner whose sketches all have the same length (> 1): the sketch of an element is a multiplication, not a table
  NodeMap nmap{T.node.as<uint32_t>(), 0u, 0u};
  if (G == 1 && D.uniform_len > 1) { nmap.len = D.uniform_len; nmap.magic = (uint32_t)((1ull << 32) / D.uniform_len); }
  if (nm) {
    // hashes are uniform over their span: the 32 most significant bits that vary decide the order of all but a few of them
    // (four passes, then k_tie_fix); all eight passes only when that has failed once for this collection
    int cur;
    const uint32_t mask = D.force_radix ? 0xffu : 0x0fu;
    const uint32_t* shift_dev = D.force_radix ? nullptr : &rs->sort_shift;
    if (G == 1) {
      if (nmap.len) hipLaunchKernelGGL(k_uf_init, dim3((n + 255) / 256), dim3(256), 0, s, T.parent.as<uint32_t>(), n);   // (no table to make)
      else hipLaunchKernelGGL(k_whole_nodes, dim3((nm / 8 + 256) / 256), dim3(256), 0, s, off, n, nm, T.node.as<uint32_t>(), T.parent.as<uint32_t>());
      if (shift_dev) hipLaunchKernelGGL(k_key_span, dim3((n + 255) / 256), dim3(256), 0, s, D.hashes, BkSeg{off, nullptr}, n, nm, rs);
      cur = radix_sort_u64_place(D.hashes, T.keys0.as<uint64_t>(), T.keys1.as<uint64_t>(), T.org0.as<uint32_t>(), T.org1.as<uint32_t>(), nm,
                                 dev.scratch, s, mask, shift_dev);
    } else {
      // (a wavefront per sketch while no sketch is longer than 64 turns of it; else a lane per 8 slice positions)
      if (D.max_len <= 4096u)
        hipLaunchKernelGGL(k_slice_gather, dim3((n + 3) / 4), dim3(256), 0, s, D.hashes, off, n, D.spart.as<uint32_t>(), G, rank,
                           D.segoff.as<uint32_t>() + (size_t)rank * (n + 1), nm, T.keys0.as<uint64_t>(), T.node.as<uint32_t>(), T.parent.as<uint32_t>());
      else
        hipLaunchKernelGGL(k_slice_gather_pos, dim3((nm / 8 + 256) / 256), dim3(256), 0, s, D.hashes, off, n, D.spart.as<uint32_t>(), G, rank,
                           D.segoff.as<uint32_t>() + (size_t)rank * (n + 1), nm, T.keys0.as<uint64_t>(), T.node.as<uint32_t>(), T.parent.as<uint32_t>());
      if (shift_dev)
        hipLaunchKernelGGL(k_key_span, dim3((n + 255) / 256), dim3(256), 0, s, T.keys0.as<uint64_t>(),
                           BkSeg{nullptr, D.segoff.as<uint32_t>() + (size_t)rank * (n + 1)}, n, nm, rs);
      // (the first pass reads keys0 and writes keys1: keys0 is input and work buffer at once)
      cur = radix_sort_u64_place(T.keys0.as<uint64_t>(), T.keys0.as<uint64_t>(), T.keys1.as<uint64_t>(), T.org0.as<uint32_t>(),
                                 T.org1.as<uint32_t>(), nm, dev.scratch, s, mask, shift_dev);
    }
    if (cur) { sk = T.keys1.as<uint64_t>(); so = T.org1.as<uint32_t>(); }
    if (shift_dev) {
      const size_t seen_bytes = ((size_t)nm / 32 + 2) * 4;
      T.ties.ensure(seen_bytes + sizeof(TieList));
      HIP_CHECK(hipMemsetAsync(T.ties.ptr, 0, seen_bytes + 8, s));      // the bitmap and the list's counter
      uint32_t* seen = T.ties.as<uint32_t>();
      TieList* list = reinterpret_cast<TieList*>(T.ties.as<uint8_t>() + seen_bytes);
      hipLaunchKernelGGL(k_tie_fix, dim3((nm + 255) / 256), dim3(256), 0, s, sk, so, nm, rs, seen, list);
      hipLaunchKernelGGL(k_tie_sort, dim3((unsigned)dev.cu_count()), dim3(kBsThreads), 0, s, sk, so, nm, rs, list);
    }
  }
  // (with the ranks: rank[origin[i]] = run of sorted position i goes out with the runs -- unless the range masks are built:
  // then the ranks leave together with the elements' bits, further down)
  D.has_masks = G == 1 && nm > 0 && compare_get_tuning().no_range_masks == 0 && masks_fit(n, D.R);
  // (several owners: every rank leaves with two flags -- its hash is held more than once; it is the first of its run -- so that
  // whoever assembles the ranks can hand out the range masks' bits without counting; same decision on every owner)
  D.share_flags = G > 1 && D.total < (1ull << 30);
  run_length_encode_u64_async(sk, nm, T.uniq.as<uint64_t>(), T.starts.as<uint32_t>(), dev.scratch, s, D.has_masks ? nullptr : so,
                              D.has_masks ? nullptr : reinterpret_cast<uint32_t*>(share + D.ranks_at), &rs->nruns, nullptr,
                              T.runid.as<uint32_t>(), D.share_flags);
  // ---- frequent hashes: held by more than a quarter of the sketches (at least 16) -- see k_freq_mark
  const uint8_t* isfreq = nullptr;
  if (D.split && nm) {
    T.isfreq.ensure(ne);
    HIP_CHECK(hipMemsetAsync(T.isfreq.ptr, 0, ne, s));
    hipLaunchKernelGGL(k_freq_mark, dim3((unsigned)std::min<uint64_t>((nm + 255) / 256, 2048)), dim3(256), 0, s, T.starts.as<uint32_t>(),
                       nm, threshold, rs);
    hipLaunchKernelGGL(k_freq_finalize, dim3(1), dim3(64), 0, s, rs, T.isfreq.as<uint8_t>(), std::max<uint32_t>(1u, kMaxFreq / G));
    isfreq = T.isfreq.as<uint8_t>();
  }
  // ---- components of the "shares a hash" graph within my slice (lock-free union-find over the runs of equal hashes)
  uint32_t* roots = reinterpret_cast<uint32_t*>(share + D.roots_at);
  if (!nm) hipLaunchKernelGGL(k_uf_init, dim3((n + 255) / 256), dim3(256), 0, s, T.parent.as<uint32_t>(), n);   // (else: k_whole_nodes / k_slice_gather did)
  if (nm) {
    // A sample of the neighbour pairs first.  A small pool (at most kUfLdsNodes sketches, 4 M hashes): 1/64 of the pairs, in LDS forests of a few
    // workgroups (k_uf_runs_lds; ~8 K pairs each) that k_uf_merge unites -- the contended unions happen in LDS, the global
    // parent array sees one union per sketch and forest.  Larger pools: 1/256 straight to the atomic path, then 1/16 through
    // the cached filter.  Finally everything through the cached filter, which sends on only the pairs not connected yet.
    if (n <= kUfLdsNodes) {
      // how much of the pairs the forests see: enough to connect n sketches (about 6 n sampled positions; a random graph is
      // connected from ~n ln n / 2 edges on), not more -- what they leave unconnected costs the full pass its atomic path
      // (one rank's slice of the dense 10 000-sketch collection at 1/64: 207 us there), what they see costs LDS time
      int shift = 8;
      while (shift > 4 && ((uint64_t)nm >> shift) < 6ull * n) shift--;
      // (~2 K sampled positions per forest, two per lane: a forest is built by ONE CU, whose loads touch a line per lane and
      // cycle -- 16 K positions per forest took 120-240 us)
      const uint32_t W = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, ((uint64_t)nm >> shift) / 2048));
      T.wroots.ensure((size_t)W * n * 4);
#define SMH_UF(S_)                                                                                                            \
  if (shift == S_)                                                                                                            \
    hipLaunchKernelGGL((k_uf_runs_lds<S_>), dim3(W), dim3(1024), (size_t)n * 4, s, sk, so, nmap, (uint64_t)nm, n, \
                       T.wroots.as<uint32_t>(), T.runid.as<uint32_t>(), isfreq);
      SMH_UF(4) SMH_UF(5) SMH_UF(6) SMH_UF(7) SMH_UF(8)
#undef SMH_UF
      // (parent[] is still the identity here; the merge leaves every sketch straight under its root: the cached filter of the
      // full pass below gives up after 64 hops, and whatever it cannot prove connected takes the atomic path)
      uf_merge_rows_lds(T.wroots.as<uint8_t>(), (uint64_t)n * 4, 0, W, n, T, T.parent.as<uint32_t>(), nullptr, s);
      // a large slice of several owners' dictionary: 1/16 of the pairs through the cached filter before all of them (half of a
      // dense 10 000-sketch pool: slice 1.47 -> 1.30 ms; one owner's whole pool is no faster for it: 3.45 -> 3.39 ms without)
      if (nm > (1u << 22) && G > 1) {
        hipLaunchKernelGGL((k_uf_runs<4, true>), dim3((unsigned)((nm / 16 + 256) / 256)), dim3(256), 0, s, sk, so, nmap,
                           (uint64_t)nm, T.parent.as<uint32_t>(), T.runid.as<uint32_t>(), isfreq);
        hipLaunchKernelGGL(k_uf_roots, dim3((n + 255) / 256), dim3(256), 0, s, T.parent.as<uint32_t>(), n, T.parent.as<uint32_t>());
      }
    } else {
      hipLaunchKernelGGL((k_uf_runs<8, false>), dim3((unsigned)((nm / 256 + 256) / 256)), dim3(256), 0, s, sk, so, nmap,
                         (uint64_t)nm, T.parent.as<uint32_t>(), T.runid.as<uint32_t>(), isfreq);
      hipLaunchKernelGGL(k_uf_roots, dim3((n + 255) / 256), dim3(256), 0, s, T.parent.as<uint32_t>(), n, T.parent.as<uint32_t>());
      hipLaunchKernelGGL((k_uf_runs<4, true>), dim3((unsigned)((nm / 16 + 256) / 256)), dim3(256), 0, s, sk, so, nmap,
                         (uint64_t)nm, T.parent.as<uint32_t>(), T.runid.as<uint32_t>(), isfreq);
      hipLaunchKernelGGL(k_uf_roots, dim3((n + 255) / 256), dim3(256), 0, s, T.parent.as<uint32_t>(), n, T.parent.as<uint32_t>());
    }
    hipLaunchKernelGGL((k_uf_runs<0, true>), dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, s, sk, so, nmap,
                       (uint64_t)nm, T.parent.as<uint32_t>(), T.runid.as<uint32_t>(), isfreq);
  }
  hipLaunchKernelGGL(k_uf_roots, dim3((n + 255) / 256), dim3(256), 0, s, T.parent.as<uint32_t>(), n, roots);
  // ---- what the slice publishes: header, roots (written above), range boundaries, local ranks in slice order
  D.dstate.ensure(sizeof(DictState));
  hipLaunchKernelGGL(k_slice_header, dim3(1), dim3(64), 0, s, rs, T.uniq.as<uint64_t>(), nm, reinterpret_cast<SliceHeader*>(share),
                     G == 1 ? D.dstate.as<DictState>() : (DictState*)nullptr, 0u);
  hipLaunchKernelGGL(k_hbounds, dim3((D.Rg + 255) / 256), dim3(256), 0, s, T.starts.as<uint32_t>(), T.uniq.as<uint64_t>(), rs, nm, D.Rg,
                     D.splitters.as<uint64_t>() + rank, reinterpret_cast<uint64_t*>(share + D.hbound_at));   // (slice's lower end: by pointer)
  // ---- range masks (one owner): a bit for every shared hash (distinct within its component and range), every element's bit,
  // the words of every range (see "range masks")
  if (D.has_masks) {
    const uint32_t R = D.R;
    T.cnt.ensure(((size_t)n * R + 3 * R + 8) * 4);    // counters per (component, range); frequent counters, maxima, total; first runs
    uint32_t* cnt = T.cnt.as<uint32_t>();
    uint32_t* fcnt = cnt + (size_t)n * R;
    uint32_t* kmax = fcnt + R;
    uint32_t* total = kmax + R;
    uint32_t* rlo = total + 4;
    D.sid.ensure(ne * 2);                                           // the bit of every element (u16)
    T.pk0.ensure(ne * 4);                                           // the bit of every run (the plan's key buffer: free until the first compare)
    uint32_t* runbit = T.pk0.as<uint32_t>();
    D.sb.ensure((size_t)(R + 1) * 4); D.woff.ensure((size_t)(R + 2) * 4); D.minfo.ensure(sizeof(MaskInfo));
    D.mask_words_max = 2 * R + kMaskWordsExtra;            // (a range's bits stay below 2^15 whatever it is: k_mask_layout checks)
    HIP_CHECK(hipMemsetAsync(cnt, 0, ((size_t)n * R + 2 * R + 4) * 4, s));
    hipLaunchKernelGGL(k_range_runs, dim3((R + 256) / 256), dim3(256), 0, s, T.starts.as<uint32_t>(), rs, nm, R, rlo);
    hipLaunchKernelGGL(k_shared_bits, dim3((nm + 255) / 256), dim3(256), 0, s, T.starts.as<uint32_t>(), rs, nm, so, nmap,
                       roots, isfreq, rlo, R, cnt, fcnt, runbit);
    hipLaunchKernelGGL(k_rank_bit_scatter, dim3((nm + 255) / 256), dim3(256), 0, s, T.runid.as<uint32_t>(), so, runbit, nm,
                       reinterpret_cast<uint32_t*>(share + D.ranks_at), D.sid.as<uint16_t>());
    hipLaunchKernelGGL(k_mask_max, dim3((unsigned)(((uint64_t)n * R + 255) / 256)), dim3(256), 0, s, cnt, (uint64_t)n * R, R, kmax, total);
    hipLaunchKernelGGL(k_mask_layout, dim3(1), dim3(1024), 0, s, kmax, fcnt, total, R, D.mask_words_max, D.sb.as<uint32_t>(),
                       D.woff.as<uint32_t>(), D.minfo.as<MaskInfo>());
  }
  HIP_CHECK(hipGetLastError());
  if (!D.force_radix && G > 1) {
    // a share goes to the other owners next (the caller waits for it anyway): it must not be a void one.  (One owner: the
    // flag travels in the dictionary's state and is looked at where the block compare synchronises, collection_compare.)
    uint32_t ovf = 0;
    HIP_CHECK(hipMemcpyAsync(&ovf, &rs->overflow, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (ovf) {
      dev.count("dictionary_rebuilt");
      collection_begin_into(D, D.hashes, nullptr, D.rel_off.data(), n, world, rank, dev, s, true);
    }
  }
}

CollectionDict* collection_begin(const uint64_t* hashes_dev, const uint64_t* offsets_dev, const uint64_t* offsets_host, uint32_t n,
                                 uint32_t world, uint32_t rank, Device& dev, hipStream_t s) {
  std::unique_ptr<CollectionDict> D(new CollectionDict());
  collection_begin_into(*D, hashes_dev, offsets_dev, offsets_host, n, world, rank, dev, s);
  return D.release();
}

uint64_t collection_share_bytes(const CollectionDict* D) { return D->share_bytes; }
const void* collection_share(const CollectionDict* D) { return D->share.ptr; }
uint32_t collection_len(const CollectionDict* D) { return D->n; }

// gathered: world x share_bytes, owner-major (what an all-gather of the shares returns); world == 1: may be null
void collection_finish(CollectionDict* Dp, const void* gathered_dev, Device& dev, hipStream_t s) {
  CollectionDict& D = *Dp;
  TiledScratch& T = tiled_scratch();
  const uint32_t G = D.world, n = D.n;
  const uint8_t* gathered = reinterpret_cast<const uint8_t*>(gathered_dev);
  if (G == 1) gathered = D.share.as<uint8_t>();   // a single owner's share is the whole dictionary
  if (!gathered) throw_internal("collection_finish: the gathered shares are missing");
  const uint64_t* off = D.off.as<uint64_t>();
  D.dstate.ensure(sizeof(DictState));
  DictState* ds = D.dstate.as<DictState>();
  if (G > 1) hipLaunchKernelGGL(k_dict_state, dim3(1), dim3(1), 0, s, gathered, D.share_bytes, G, ds);   // (one owner: k_slice_header did it)
  if (G == 1) {
    // one owner: its share already IS the dictionary (slice order == collection order, one forest, one boundary list)
    D.rank_ptr = reinterpret_cast<const uint32_t*>(gathered + D.ranks_at);
    D.root_ptr = reinterpret_cast<const uint32_t*>(gathered + D.roots_at);
    D.hbound_ptr = reinterpret_cast<const uint64_t*>(gathered + D.hbound_at);
  } else {
    // ranks of every element, in collection order
    D.rankv.ensure(std::max<uint64_t>(D.total, 1) * 4 + 16);   // (+16: read four at a time from any position)
    // (with the flags of the shares, and if range masks may be wanted: the elements' states for k_claim_bits)
    D.lazy_ready = D.share_flags && D.total > 0 && compare_get_tuning().no_range_masks == 0 && masks_fit(n, D.R);
    if (D.lazy_ready) D.sid.ensure((size_t)D.total * 2);
    if (D.total)
      hipLaunchKernelGGL(k_reassemble, dim3(std::min<uint32_t>(n, 65536)), dim3(256), 0, s, off, n, D.spart.as<uint32_t>(), G,
                         D.segoff.as<uint32_t>(), gathered, D.share_bytes, D.ranks_at, ds, D.rankv.as<uint32_t>(), D.share_flags ? 1u : 0u,
                         D.lazy_ready ? D.sid.as<uint16_t>() : (uint16_t*)nullptr);
    // components: the slices' forests united
    T.parent.ensure((size_t)n * 4);
    D.root.ensure((size_t)n * 4);
    if (n <= kUfLdsNodes && G <= 64) {
      uf_merge_rows_lds(gathered, D.share_bytes, D.roots_at, G, n, T, T.parent.as<uint32_t>(), D.root.as<uint32_t>(), s);
    } else {
      hipLaunchKernelGGL(k_uf_init, dim3((n + 255) / 256), dim3(256), 0, s, T.parent.as<uint32_t>(), n);
      hipLaunchKernelGGL(k_uf_merge, dim3((unsigned)(((uint64_t)G * n + 255) / 256)), dim3(256), 0, s, gathered, D.share_bytes, D.roots_at, G, n,
                         T.parent.as<uint32_t>());
      hipLaunchKernelGGL(k_uf_roots, dim3((n + 255) / 256), dim3(256), 0, s, T.parent.as<uint32_t>(), n, D.root.as<uint32_t>());
    }
    // range boundaries of the tiled kernel, slice after slice
    D.hbound.ensure((size_t)D.R * 8);
    hipLaunchKernelGGL(k_gather_bounds, dim3((D.R + 255) / 256), dim3(256), 0, s, gathered, D.share_bytes, D.hbound_at, D.Rg, D.R,
                       D.hbound.as<uint64_t>());          // (one launch; a copy per slice was 5 us each)
    D.rank_ptr = D.rankv.as<uint32_t>(); D.root_ptr = D.root.as<uint32_t>(); D.hbound_ptr = D.hbound.as<uint64_t>();
  }
  // frequent hashes: the per-sketch records (which of them it holds, and where)
  if (D.split) {
    D.fmask.ensure((size_t)n * 8); D.fpos.ensure((size_t)n * kMaxFreq * 4);
    hipLaunchKernelGGL(k_freq_records, dim3((n + 255) / 256), dim3(256), 0, s, D.hashes, off, n, ds,
                       D.fmask.as<unsigned long long>(), D.fpos.as<uint32_t>());
  }
  if (G > 1) {
    D.minfo.ensure(sizeof(MaskInfo));
    HIP_CHECK(hipMemsetAsync(D.minfo.ptr, 0, sizeof(MaskInfo), s));      // (no masks until a block compare has built them)
    D.lazy_tried = false;
  }
  D.part.ensure((size_t)n * (D.R + 1) * 4);    // filled by the first block compare that may take the tiled route
  HIP_CHECK(hipGetLastError());
  D.finished = true;
  (void)dev;
}

void collection_free(CollectionDict* D) {
  if (!D) return;
  (void)hipDeviceSynchronize();    // ONE wait for whatever may still be using the buffers, then they all go back to the pool
  for (DeviceBuffer* b : {&D->off, &D->splitters, &D->spart, &D->segoff, &D->share, &D->dstate, &D->rankv, &D->root, &D->fmask, &D->fpos,
                          &D->hbound, &D->part, &D->sid, &D->sb, &D->woff, &D->minfo, &D->masks, &D->partT})
    b->release_after_sync();
  delete D;
}
static CollectionDict& implicit_dict();
static void release_implicit_dict() {
  CollectionDict& D = implicit_dict();
  for (DeviceBuffer* b : {&D.off, &D.splitters, &D.spart, &D.segoff, &D.share, &D.dstate, &D.rankv, &D.root, &D.fmask, &D.fpos,
                          &D.hbound, &D.part, &D.sid, &D.sb, &D.woff, &D.minfo, &D.masks, &D.partT})
    b->release();
  D.finished = false;
}

// one owner, and the four-pass sort of the pooled hashes gave up: the dictionary again, with all eight passes (rare: the
// caller's call takes twice as long)
static void collection_rebuild(CollectionDict& D, Device& dev, hipStream_t s) {
  collection_begin_into(D, D.hashes, nullptr, D.rel_off.data(), D.n, D.world, D.rank, dev, s, true);
  collection_finish(&D, nullptr, dev, s);
  dev.count("dictionary_rebuilt");
}

// rows [row_lo, row_hi) x columns [col_lo, col_hi) of the collection's all-vs-all matrix; outputs row-major
// (row_hi - row_lo) x (col_hi - col_lo).  own_mode: see PairScope (1 needs rows == columns == everything, 2 needs
// columns == everything; both need one num).
void collection_compare(CollectionDict* Dp, uint32_t row_lo, uint32_t row_hi, uint32_t col_lo, uint32_t col_hi, uint32_t num,
                        const uint32_t* row_nums, uint32_t own_mode, const CompareOut& out, Device& dev, hipStream_t s) {
  CollectionDict& D = *Dp;
  if (!D.finished) throw_internal("collection_compare before collection_finish");
  if (row_lo > row_hi || row_hi > D.n || col_lo > col_hi || col_hi > D.n) throw_internal("collection_compare: block outside the collection");
  const uint32_t nrows = row_hi - row_lo, ncols = col_hi - col_lo;
  if (nrows == 0 || ncols == 0) return;
  TiledScratch& T = tiled_scratch();
  const CompareTuning tune = compare_get_tuning();
  const TiledExperiments& ex = tiled_experiments();
  const bool all_cols = col_lo == 0 && col_hi == D.n;
  if (row_nums || !tune.use_symmetry) own_mode = 0;
  if (own_mode == 1 && !(all_cols && row_lo == 0 && row_hi == D.n)) own_mode = 0;
  if (own_mode == 2 && !all_cols) own_mode = 0;
  const bool same = own_mode == 1;      // one slot order serves both axes
  PairScope sc;
  sc.own_mode = own_mode; sc.ntotal = D.n; sc.row_base = row_lo; sc.col_base = col_lo;
  sc.mir_lo = own_mode ? row_lo : 0; sc.mir_hi = own_mode ? row_hi : 0;
  const bool want_cc = out.count_common || out.containment;
  const uint64_t* off = D.off.as<uint64_t>();
  SketchSet rows, cols;
  rows.hashes = D.hashes; rows.offsets = off + row_lo; rows.n = nrows;
  cols.hashes = D.hashes; cols.offsets = off + col_lo; cols.n = ncols;
  // the AUTO route's pair limit, clamped ONCE so that the work list sized from it always holds what the device-side
  // choice (k_plan_route, same value) can produce: a tuning value never changes a result or raises
  const uint64_t kWorkCapMax = 1ull << 26;
  uint64_t comp_limit = std::min<uint64_t>(tune.comp_pairs_limit, kWorkCapMax > ncols ? kWorkCapMax - ncols : 0);
  // (with range masks a tile is a prologue and one short walk, ~0.1 ms for a round of tiles however few: the pair kernel,
  // ~7 ns per pair of num = 2000 sketches, only wins below ~16 Ki sharing pairs -- 2 000 sketches in 50 families, 80 000 pairs:
  // 1.10 -> 0.81 ms; without masks a tile's chain of staged stretches lasts ~0.4 ms and the tuning's limit stands)
  // (one owner's dictionary only: a sliced one builds its masks when the block has MANY sharing pairs -- lazy_go)
  if (compare_get_tuning().no_range_masks == 0 && D.has_masks) comp_limit = std::min<uint64_t>(comp_limit, kCompPairsWithMasks);
  // ---- every pair as if it shared nothing but frequent hashes; the compare kernels overwrite the pairs they walk
  // (with every tile launched nothing would be left: skipped).  It needs nothing the plan makes and is bound by its writes
  // (800 MB at 10 000 x 10 000), so it runs on the library's second stream beside the plan's small launches and is
  // waited for before the first compare kernel.
  const uint64_t np = (uint64_t)nrows * ncols;
  const hipEvent_t ev_fork = dev.fork_event(), ev_join = dev.join_event();     // (the device's own pair; under its mutex)
  bool fill_pending = false;
  struct FillGuard {            // an error on the way out must not leave the fill writing into the caller's buffers
    bool& pending; hipStream_t s2;
    ~FillGuard() { if (pending) (void)hipStreamSynchronize(s2); }
  } fill_guard{fill_pending, dev.copy_stream()};
  if (!(tune.visit_all_tiles && tune.route == kRouteTiled)) {
    // (a small block's fill is a few microseconds: not worth two events)
    const bool beside = np >= (4ull << 20);
    hipStream_t s2 = beside ? dev.copy_stream() : s;
    const unsigned long long* fm = D.split ? D.fmask.as<unsigned long long>() : nullptr;
    const uint32_t* fp = D.split ? D.fpos.as<uint32_t>() : nullptr;
    if (beside) {
      HIP_CHECK(hipEventRecord(ev_fork, s));        // what came before on s (the dictionary, the caller's buffers) is done
      HIP_CHECK(hipStreamWaitEvent(s2, ev_fork, 0));
    }
    dev.prof_begin(s2);
    hipLaunchKernelGGL(k_fill_disjoint, dim3((ncols + 255) / 256, std::min<uint32_t>((nrows + kFillRows - 1) / kFillRows, 65535u)), dim3(256), 0, s2,
                       rows.offsets, nrows, cols.offsets, ncols, num,
                       row_nums, out, fm ? fm + row_lo : nullptr, fp ? fp + row_lo : nullptr,
                       fm ? fm + col_lo : nullptr, fp ? fp + col_lo : nullptr, D.n);
    HIP_CHECK(hipGetLastError());
    dev.prof_end("compare_fill", s2);
    if (beside) {
      HIP_CHECK(hipEventRecord(ev_join, s2));
      fill_pending = true;
    }
  }

  T.plan.ensure(sizeof(PlanState));
  PlanState* st = T.plan.as<PlanState>();
  HIP_CHECK(hipMemsetAsync(st, 0, sizeof(PlanState), s));

  // ---- a large block of one owner's dictionary: the partition table and the masks (both the dictionary's, built once) are
  // made NOW, beside the fill -- they need nothing of the plan, and behind it they were 0.16 ms on the way to the first tile
  // (st->skip_tiled is still 0 here: the gate they share with the plan's route is open; part_built says "already there")
  const bool early_tables = fill_pending && tune.route != kRouteComponents && D.world == 1 && D.finished;
  if (early_tables) {
    DictState* ds = D.dstate.as<DictState>();
    const uint32_t R = D.R;
    hipLaunchKernelGGL(k_partition, dim3((unsigned)(((uint64_t)D.n * (R + 1) + 255) / 256)), dim3(256), 0, s, D.hashes, off, D.n,
                       D.hbound_ptr, R, D.part.as<uint32_t>(), &st->skip_tiled, &ds->part_built);
    if (D.has_masks && tune.no_range_masks == 0) {
      D.masks.ensure((size_t)D.n * (D.mask_words_max + 3) * 8);
      D.partT.ensure((size_t)D.n * (R + 1) * 4);
      hipLaunchKernelGGL(k_build_masks, dim3((unsigned)(((uint64_t)D.n * (R + 1) + 255) / 256)), dim3(256), 0, s, D.part.as<uint32_t>(), off,
                         D.sid.as<uint16_t>(), D.sb.as<uint32_t>(), D.woff.as<uint32_t>(), D.minfo.as<MaskInfo>(), D.n, R,
                         D.masks.as<unsigned long long>(), D.partT.as<uint32_t>(), &st->skip_tiled, &ds->part_built);
    }
  }

  // ---- slot orders: sketches sorted by component (stable: the index is the low half of the key)
  const uint32_t* root_r = D.root_ptr + row_lo;
  const uint32_t* root_c = D.root_ptr + col_lo;
  T.pk0.ensure((size_t)nrows * 8); T.pk1.ensure((size_t)nrows * 8);
  hipLaunchKernelGGL(k_plan_keys, dim3((nrows + 255) / 256), dim3(256), 0, s, root_r, nrows, T.pk0.as<uint64_t>());
  const uint64_t* rkey = radix_sort_u64_keys(T.pk0.as<uint64_t>(), T.pk1.as<uint64_t>(), nrows, dev.scratch, s,
                                             plan_key_passes(D.n, nrows)) ? T.pk1.as<uint64_t>() : T.pk0.as<uint64_t>();
  const uint64_t* ckey = rkey;
  if (!same) {
    T.pk2.ensure((size_t)ncols * 8); T.pk3.ensure((size_t)ncols * 8);
    hipLaunchKernelGGL(k_plan_keys, dim3((ncols + 255) / 256), dim3(256), 0, s, root_c, ncols, T.pk2.as<uint64_t>());
    ckey = radix_sort_u64_keys(T.pk2.as<uint64_t>(), T.pk3.as<uint64_t>(), ncols, dev.scratch, s, plan_key_passes(D.n, ncols))
               ? T.pk3.as<uint64_t>() : T.pk2.as<uint64_t>();
  }
  // ---- per slot, the other side's slots of the same component; pairs that can share a hash
  T.rng.ensure(((size_t)nrows + ncols) * 2 * 4);
  uint32_t* col_lo_s = T.rng.as<uint32_t>();              // by row slot
  uint32_t* col_hi_s = col_lo_s + nrows;
  uint32_t* row_lo_s = same ? col_lo_s : col_hi_s + nrows; // by column slot
  uint32_t* row_hi_s = same ? col_hi_s : row_lo_s + ncols;
  hipLaunchKernelGGL(k_plan_ranges, dim3((nrows + 255) / 256), dim3(256), 0, s, rkey, nrows, ckey, ncols, col_lo_s, col_hi_s, st);
  if (!same)
    hipLaunchKernelGGL(k_plan_ranges, dim3((ncols + 255) / 256), dim3(256), 0, s, ckey, ncols, rkey, nrows, row_lo_s, row_hi_s,
                       (PlanState*)nullptr);
  hipLaunchKernelGGL(k_plan_route, dim3(1), dim3(1), 0, s, st, tune.route, tune.visit_all_tiles, (unsigned long long)comp_limit);
  HIP_CHECK(hipGetLastError());

  // ---- per-component pair kernel: one workgroup per (column, <= 32 rows of its component).  Its work list is made here; the
  // kernel is launched with the tiled ones, after the fill has been waited for (the compare kernels write over what it wrote)
  uint32_t work_cap = 0;
  bool comp_planned = false;
  auto join_fill = [&] {
    if (fill_pending) HIP_CHECK(hipStreamWaitEvent(s, ev_join, 0));
    fill_pending = false;
  };
  auto launch_comp = [&] {
    if (!comp_planned) return;
    comp_planned = false;
    const uint32_t col_max = D.max_len;
    const bool q_lds = col_max <= 8192;
    const size_t lds = q_lds ? (size_t)(col_max ? col_max : 1) * 8 : 16;
    const unsigned grid = (unsigned)std::min<uint64_t>(work_cap, (uint64_t)dev.cu_count() * 8);
    dev.prof_begin(s);
#define SMH_CC(L_, C_) hipLaunchKernelGGL((k_compare_comp<L_, C_>), dim3(grid), dim3(256), lds, s, rows, cols, \
                                          reinterpret_cast<const CompWork*>(T.work.ptr), &st->nwork, work_cap, rkey, num, row_nums, \
                                          sc, out)
    if (q_lds) { if (want_cc) SMH_CC(true, true); else SMH_CC(true, false); }
    else { if (want_cc) SMH_CC(false, true); else SMH_CC(false, false); }
#undef SMH_CC
    HIP_CHECK(hipGetLastError());
    dev.prof_end("compare_comp", s);
  };
  if (tune.route != kRouteTiled) {
    // the route is taken when pairs <= comp_limit: every item holds a pair, every column adds at most one
    // partly filled item.  (A forced route on a huge block is capped; the overflow is reported.)
    uint64_t cap = (tune.route == kRouteComponents ? ((uint64_t)ncols * ((nrows + kRowsPerItem - 1) / kRowsPerItem)) : comp_limit) + ncols;
    if (cap > kWorkCapMax) cap = kWorkCapMax;
    work_cap = (uint32_t)cap;
    T.cnt.ensure((size_t)ncols * 4);
    T.work.ensure((size_t)work_cap * sizeof(CompWork));
    hipLaunchKernelGGL(k_comp_count, dim3((ncols + 255) / 256), dim3(256), 0, s, ncols, row_lo_s, row_hi_s, same ? 1u : 0u,
                       T.cnt.as<uint32_t>(), st);
    exclusive_scan_u32_dev(T.cnt.as<uint32_t>(), ncols, &st->nwork, dev.scratch, s);
    hipLaunchKernelGGL(k_comp_fill, dim3((ncols + 255) / 256), dim3(256), 0, s, ckey, ncols, row_lo_s, row_hi_s, same ? 1u : 0u,
                       T.cnt.as<uint32_t>(), reinterpret_cast<CompWork*>(T.work.ptr), work_cap, st);
    comp_planned = true;
  }
  if (tune.route == kRouteComponents) { join_fill(); launch_comp(); }       // (no tiled part follows)

  // ---- tiled kernel: tile list, launch
  const int wpb = ex.wpb, minw = ex.minw;
  uint32_t tiles_cap = 0;
  if (tune.route != kRouteComponents) {
    const uint32_t R = D.R;
    TileTest tt;
    tt.nrows = nrows; tt.ncols = ncols; tt.col_lo = col_lo_s; tt.col_hi = col_hi_s; tt.rkey = rkey; tt.ckey = ckey; tt.sc = sc;
    tt.all_on = 0;
    // rows per tile: decided on the device from the number of 16-row tiles that hold sharing pairs
    // (fewer than ~4 rounds over the chip: 8-row, then 4-row tiles keep all wave slots busy;
    // profiles/r01_compare_small_geometry.txt).  With every tile launched the count is known here.
    const uint32_t fill_tiles = (uint32_t)dev.cu_count() * 32;
    const uint32_t tiles_c = (ncols + kTB - 1) / kTB;
    // (experiments build: SOURMASH_AMD_CMP_GEO forces the height, SOURMASH_AMD_CMP_PF = 1 the pipelined kernel)
    uint32_t forced_rpw = ex.rpw > 0 ? (uint32_t)ex.rpw : 0u, forced_pf = ex.pf ? 1u : 0u;
    if (!forced_rpw && tune.visit_all_tiles) {
      const uint64_t all16 = (uint64_t)((nrows + 15) / 16) * tiles_c * (same ? 1 : 2) / 2;
      const TileShape sh = tile_shape_for(all16, fill_tiles);
      forced_rpw = sh.rpw; forced_pf = sh.pf;
    }
    if (!forced_rpw)
      hipLaunchKernelGGL(k_tiles_count16, dim3((unsigned)std::min<uint64_t>(((uint64_t)((nrows + 15) / 16) * tiles_c + 255) / 256, 4096)),
                         dim3(256), 0, s, tt, st);
    DictState* ds = D.dstate.as<DictState>();
    if (!early_tables)
      hipLaunchKernelGGL(k_partition, dim3((unsigned)(((uint64_t)D.n * (R + 1) + 255) / 256)), dim3(256), 0, s, D.hashes, off, D.n,
                         D.hbound_ptr, R, D.part.as<uint32_t>(), &st->skip_tiled, &ds->part_built);
    // a sliced dictionary (world > 1) has its masks built here, from the assembled ranks, roots and crossings
    const bool lazy_masks = D.world > 1 && D.lazy_ready && tune.no_range_masks == 0 && !ex.no_masks;
    const bool use_masks = (D.has_masks || lazy_masks) && tune.no_range_masks == 0 && !ex.no_masks;
    if (lazy_masks && !D.lazy_tried) {
      // (once per dictionary: a later block compare finds the masks, or -- the plan of this one skipped the tiles -- walks)
      D.lazy_tried = true;
      const size_t ids = (size_t)D.total + 1;                    // (ranks are below the number of pooled hashes)
      T.mv1.ensure(ids * 4);
      T.cnt.ensure(((size_t)D.n * R + 3 * R + 8) * 4);
      uint32_t* cnt = T.cnt.as<uint32_t>();
      uint32_t* fcnt = cnt + (size_t)D.n * R;
      uint32_t* kmax = fcnt + R;
      uint32_t* total = kmax + R;
      D.sb.ensure((size_t)(R + 1) * 4); D.woff.ensure((size_t)(R + 2) * 4); D.minfo.ensure(sizeof(MaskInfo));
      D.mask_words_max = 2 * R + kMaskWordsExtra;
      HIP_CHECK(hipMemsetAsync(cnt, 0, ((size_t)D.n * R + 2 * R + 4) * 4, s));
      const uint32_t nt = (uint32_t)D.total;
      const unsigned gb = (unsigned)((D.total + 255) / 256);
      hipLaunchKernelGGL(k_claim_bits, dim3(gb), dim3(256), 0, s, D.rank_ptr, D.hashes, off, D.part.as<uint32_t>(), D.root_ptr, ds, D.n, R,
                         nt, D.sid.as<uint16_t>(), T.mv1.as<uint32_t>(), cnt, fcnt, D.split ? 1u : 0u, st, &ds->part_built);
      hipLaunchKernelGGL(k_elem_bits, dim3(gb), dim3(256), 0, s, D.rank_ptr, nt, T.mv1.as<uint32_t>(), D.sid.as<uint16_t>(), st,
                         &ds->part_built);
      hipLaunchKernelGGL(k_mask_max_lazy, dim3((unsigned)(((uint64_t)D.n * R + 255) / 256)), dim3(256), 0, s, cnt, (uint64_t)D.n * R, R, kmax,
                         total, st, &ds->part_built, nt);
      hipLaunchKernelGGL(k_mask_layout, dim3(1), dim3(1024), 0, s, kmax, fcnt, total, R, D.mask_words_max, D.sb.as<uint32_t>(),
                         D.woff.as<uint32_t>(), D.minfo.as<MaskInfo>(), st, &ds->part_built, nt);
    }
    if (use_masks && !(early_tables && D.has_masks)) {
      D.masks.ensure((size_t)D.n * (D.mask_words_max + 3) * 8);      // (+3 words: the kernel reads three words per range whatever it has)
      D.partT.ensure((size_t)D.n * (R + 1) * 4);
      hipLaunchKernelGGL(k_build_masks, dim3((unsigned)(((uint64_t)D.n * (R + 1) + 255) / 256)), dim3(256), 0, s, D.part.as<uint32_t>(), off,
                         D.sid.as<uint16_t>(), D.sb.as<uint32_t>(), D.woff.as<uint32_t>(), D.minfo.as<MaskInfo>(), D.n, R,
                         D.masks.as<unsigned long long>(), D.partT.as<uint32_t>(), &st->skip_tiled, &ds->part_built);
    }
    hipLaunchKernelGGL(k_plan_geometry, dim3(1), dim3(1), 0, s, st, forced_rpw, forced_pf, fill_tiles, &ds->part_built);
    // the list: at 16 rows per tile at most every tile; shorter tiles are only chosen when fewer than
    // fill_tiles 16-row tiles are flagged (each splits into at most 4)
    const uint32_t rows_min = (forced_rpw ? forced_rpw : 1u) * (uint32_t)wpb;
    uint64_t cap = forced_rpw ? (uint64_t)((nrows + rows_min - 1) / rows_min) * tiles_c
                              : std::max<uint64_t>((uint64_t)((nrows + 4 * wpb - 1) / (4 * wpb)) * tiles_c, 4ull * fill_tiles);
    if (cap >= (1ull << 30)) throw_internal("compare block: too many tiles");
    tiles_cap = (uint32_t)cap;
    T.tiles.ensure((size_t)tiles_cap * 8 + 8);
    const uint64_t flag_tiles = (uint64_t)((nrows + rows_min - 1) / rows_min) * tiles_c;   // the finest geometry the plan may pick
    tt.all_on = tune.visit_all_tiles ? 1u : 0u;
    hipLaunchKernelGGL(k_flag_tiles, dim3((unsigned)std::min<uint64_t>((flag_tiles + 255) / 256, 8192)), dim3(256), 0, s, tt, (uint32_t)wpb,
                       T.tiles.as<uint32_t>(), tiles_cap, st);
    HIP_CHECK(hipGetLastError());
    TiledArgs a;
    a.rrank = D.rank_ptr; a.roff = rows.offsets; a.rpart = D.part.as<uint32_t>() + (size_t)row_lo * (R + 1); a.nrows = nrows;
    a.crank = D.rank_ptr; a.coff = cols.offsets; a.cpart = D.part.as<uint32_t>() + (size_t)col_lo * (R + 1); a.ncols = ncols;
    a.R = R; a.num = num; a.row_nums = row_nums;
    a.tiles = T.tiles.as<uint32_t>(); a.tiles_cap = tiles_cap; a.rkey = rkey; a.ckey = ckey; a.st = st;
    a.use_xcd = ex.xcd ? 1u : 0u;
    a.scope = sc;
    a.masks = use_masks ? D.masks.as<unsigned long long>() : nullptr; a.partT = D.partT.as<uint32_t>(); a.woff = D.woff.as<uint32_t>();
    a.wn = D.sb.as<uint32_t>(); a.minfo = D.minfo.as<MaskInfo>(); a.nsk = D.n;
    // LDS budget per workgroup ~18 KB so that 8 workgroups of 4 waves fit a CU: the merge loop is a
    // dependent LDS-read -> compare -> advance chain, and occupancy is what hides its latency
    // (profiles/r01_compare_geometry.txt: 575 -> 1000 M pairs/s from 3 to 8 waves per SIMD)
    a.capA = ex.capA;         // 16 rows x (~24 elements + sentinel) with 2.5x head-room
    a.capBt = ex.capB * kTB;  // columns up to 47 elements in one range
    a.ovf_steps = &st->ovf_steps;
    a.out = out;
    join_fill();
    launch_comp();
    [[maybe_unused]] const size_t lds = (size_t)(520 + a.capA + a.capBt) * 4;      // (the plain kernel: experiments build)
    [[maybe_unused]] const unsigned grid = (unsigned)dev.cu_count() * 8;          // a multiple of 8: one stretch of the list per XCD
    dev.prof_begin(s);
    bool launched = false;
#define SMH_CT(R_, W_, M_)                                                                                           \
  if ((forced_rpw == 0 || forced_rpw == R_) && wpb == W_ && minw == M_) {                                          \
    launched = true;                                                                                                 \
    if (want_cc) hipLaunchKernelGGL((k_compare_tiled<true, R_, W_, M_>), dim3(grid), dim3(64 * W_), lds, s, a);    \
    else hipLaunchKernelGGL((k_compare_tiled<false, R_, W_, M_>), dim3(grid), dim3(64 * W_), lds, s, a);           \
  }
    // the instantiations the plan chooses among (each returns at once unless it is the one): 8-, 16- and 32-row tiles by the
    // pipelined kernel (8 waves x 1 / 2 / 4 rows, 39 KB of LDS: 4 workgroups per CU)
    const size_t lds_pf = (size_t)(kPfHeader + 2 * (a.capA + a.capBt)) * 4;
    const unsigned grid_pf = (unsigned)dev.cu_count() * 4;
    // (with masks, the 8- and 16-row tiles -- launched when there are few tiles, where a tile's own time counts -- run six
    // waves per SIMD with 80 vector registers instead of eight with 64: fewer scalar values parked in vector lanes, a tile
    // 10-15 % shorter; the 32-row tiles of a large block need the eight: 2.8 against 3.2 ms at 10 000 x 10 000)
    const bool six_ok = use_masks;
    const unsigned grid_pf6 = (unsigned)dev.cu_count() * 3 / 8 * 8;      // (a multiple of 8: one stretch of the list per XCD)
#define SMH_PF(R_, W_)                                                                                                      \
  if (wpb == 4 && minw == 8 && (forced_rpw == 0 || (forced_rpw * 4u == R_ * W_ && forced_pf))) {                           \
    launched = true;                                                                                                        \
    if (six_ok && R_ < 4) {                                                                                                 \
      if (want_cc) hipLaunchKernelGGL((k_compare_tiled_pf<true, R_, W_, 6>), dim3(grid_pf6), dim3(64 * W_), lds_pf, s, a);  \
      else hipLaunchKernelGGL((k_compare_tiled_pf<false, R_, W_, 6>), dim3(grid_pf6), dim3(64 * W_), lds_pf, s, a);         \
    } else {                                                                                                                \
      if (want_cc) hipLaunchKernelGGL((k_compare_tiled_pf<true, R_, W_, 8>), dim3(grid_pf), dim3(64 * W_), lds_pf, s, a);   \
      else hipLaunchKernelGGL((k_compare_tiled_pf<false, R_, W_, 8>), dim3(grid_pf), dim3(64 * W_), lds_pf, s, a);          \
    }                                                                                                                       \
  }
    bool pf32_big = false;
#ifdef SMH_EXPERIMENTS
    pf32_big = std::getenv("SOURMASH_AMD_CMP_PF32_BIG") != nullptr;
#endif
    if (!pf32_big) SMH_PF(4, 8)
    SMH_PF(2, 8) SMH_PF(1, 8)
#ifdef SMH_EXPERIMENTS
    if (forced_rpw && forced_pf) { SMH_PF(1, 4) }
    if (forced_rpw && !forced_pf) { SMH_CT(4, 4, 8) SMH_CT(2, 4, 8) SMH_CT(1, 4, 8) }
    if (forced_rpw == 8 && forced_pf && wpb == 4 && minw == 8 && pf32_big) {
      launched = true;
      // 32-row tiles with a row stage of twice the size: 3 workgroups per CU (28.0 ms at 10 000 x 10 000 against 25.0)
      TiledArgs a32 = a;
      a32.capA = 2 * a.capA;
      const size_t lds32 = (size_t)(kPfHeader + 2 * (a32.capA + a32.capBt)) * 4;
      if (want_cc) hipLaunchKernelGGL((k_compare_tiled_pf<true, 4, 8, 6>), dim3((unsigned)dev.cu_count() * 3), dim3(512), lds32, s, a32);
      else hipLaunchKernelGGL((k_compare_tiled_pf<false, 4, 8, 6>), dim3((unsigned)dev.cu_count() * 3), dim3(512), lds32, s, a32);
    }
    SMH_CT(4, 4, 1) SMH_CT(8, 4, 1) SMH_CT(16, 4, 1) SMH_CT(4, 8, 1) SMH_CT(8, 8, 1) SMH_CT(2, 8, 1)
    SMH_CT(1, 8, 8) SMH_CT(4, 8, 8) SMH_CT(2, 8, 8) SMH_CT(4, 8, 6) SMH_CT(4, 4, 6) SMH_CT(2, 16, 8) SMH_CT(4, 16, 8)
#endif
#undef SMH_PF
#undef SMH_CT
    if (!launched) throw_internal("compare geometry not instantiated");
    HIP_CHECK(hipGetLastError());
    dev.prof_end("compare_tiled", s);
  }

  // ---- the one synchronisation of the call: what the plan decided, for the record
  PlanState h;
  uint32_t hd[4] = {0, 0, 0, 0};   // DictState: nruns, nfreq, part_built, overflow
  HIP_CHECK(hipMemcpyAsync(&h, st, sizeof(PlanState), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipMemcpyAsync(hd, D.dstate.ptr, 16, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (hd[3] && D.world == 1 && !D.force_radix) {
    // the dictionary was void (keys that tie in the sorted bits were left out of order: equal hashes may carry different
    // ranks): what the kernels compared was rubbish, though nothing of it was used as an address.  Again, with the full sort.
    collection_rebuild(D, dev, s);
    return collection_compare(Dp, row_lo, row_hi, col_lo, col_hi, num, row_nums, own_mode, out, dev, s);
  }
  CompareStats rec;
  rec.route = h.route;
  if (h.route == kRouteComponents) {
    if (h.nwork > work_cap) throw_internal("compare block: the per-component route was forced on a block with too many sharing pairs");
    rec.tiles_visited = h.pairs; rec.tiles_total = np; rec.pairs_per_tile = 1;
  } else {
    if (h.ntiles > tiles_cap) throw_internal("compare block: tile list overflow");
    const uint32_t tr = h.rpw * (uint32_t)wpb;
    rec.rows_per_tile = tr;
    rec.tiles_visited = h.ntiles;
    rec.tiles_total = (uint64_t)((nrows + tr - 1) / tr) * ((ncols + kTB - 1) / kTB);
    rec.pairs_per_tile = (uint64_t)tr * kTB;
    rec.lds_overflow_steps = h.ovf_steps;
    rec.pipelined = h.pf;
    rec.span_halvings = h.halvings;
    rec.prefetched_after_halving = h.pf_after_halving;
    if (h.bad_tables) throw_internal("compare block: a range table of the tiled kernel was inconsistent (segment end before its start)");
  }
  rec.frequent_hashes = D.split ? hd[1] : 0;
  set_stats(rec);
}

// rows x cols through the dictionary path: the same CSR on both axes is one collection compared with itself (upper
// triangle + mirrors when there is one num); two different sets are concatenated into one collection and the block
// rows x cols of its matrix is computed.
static CollectionDict& implicit_dict() {
  static CollectionDict* d = new CollectionDict();
  return *d;
}
static void launch_tiled(const SketchSet& rows, const SketchSet& cols, uint64_t nr_elems, uint64_t nc_elems,
                         uint32_t num, const uint32_t* row_nums, const CompareOut& out, Device& dev,
                         hipStream_t s, bool same_sets) {
  TiledScratch& T = tiled_scratch();
  // same_sets: the caller vouches that rows and columns are one CSR (same hashes, same offsets)
  const bool same = same_sets && rows.hashes == cols.hashes && rows.n == cols.n && nr_elems == nc_elems;
  std::vector<uint64_t> tmp_r, tmp_c, cat_off;
  auto host_offsets = [&](const SketchSet& set, std::vector<uint64_t>& tmp) -> const uint64_t* {
    if (set.h_offsets) return set.h_offsets;      // the caller had them at hand: no read-back
    tmp.resize((size_t)set.n + 1);
    HIP_CHECK(hipMemcpyAsync(tmp.data(), set.offsets, tmp.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return tmp.data();
  };
  const uint64_t* h_off = nullptr;
  const uint64_t* hashes = nullptr;
  const uint64_t* d_off = nullptr;
  uint32_t n = 0;
  if (same) {
    h_off = host_offsets(cols, tmp_c);
    hashes = cols.hashes; d_off = cols.offsets; n = cols.n;
  } else {
    // one collection: the rows' sketches, then the columns'
    const uint64_t* ro = host_offsets(rows, tmp_r);
    const uint64_t* co = host_offsets(cols, tmp_c);
    n = rows.n + cols.n;
    cat_off.resize((size_t)n + 1);
    for (uint32_t i = 0; i <= rows.n; i++) cat_off[i] = ro[i] - ro[0];
    for (uint32_t j = 0; j <= cols.n; j++) cat_off[rows.n + j] = nr_elems + (co[j] - co[0]);
    h_off = cat_off.data();
    T.cat.ensure(std::max<uint64_t>(nr_elems + nc_elems, 1) * 8);
    if (nr_elems) HIP_CHECK(hipMemcpyAsync(T.cat.ptr, rows.hashes + ro[0], nr_elems * 8, hipMemcpyDeviceToDevice, s));
    if (nc_elems) HIP_CHECK(hipMemcpyAsync(T.cat.as<uint64_t>() + nr_elems, cols.hashes + co[0], nc_elems * 8, hipMemcpyDeviceToDevice, s));
    hashes = T.cat.as<uint64_t>();
  }
  // the dictionary object of these one-off calls is kept (its buffers only grow): no allocation per call
  CollectionDict& D = implicit_dict();
  collection_begin_into(D, hashes, d_off, h_off, n, 1, 0, dev, s);
  collection_finish(&D, nullptr, dev, s);
  if (same) collection_compare(&D, 0, n, 0, n, num, row_nums, 1, out, dev, s);
  else collection_compare(&D, 0, rows.n, rows.n, n, num, row_nums, 0, out, dev, s);
}

// ---- the mirrored-block exchange of a sharded matrix (own_mode 2), device side ---------------------
// pack: what rank r sends to the rank that holds rows [col_lo, col_hi): its own block's columns col_lo.. transposed, so that
// the receiver gets (its rows) x (r's rows) row-major.  32 x 32 tiles through LDS: both sides coalesced.
namespace {
__global__ __launch_bounds__(256) void k_mirror_pack(const uint64_t* __restrict__ out, uint32_t n_local, uint32_t n_total,
                                                     uint32_t col_lo, uint32_t ncols, uint64_t* __restrict__ packed) {
  __shared__ uint64_t tile[32][33];
  const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
  const uint32_t j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  for (uint32_t k = ty; k < 32; k += 8) {
    const uint32_t i = i0 + k, j = j0 + tx;
    if (i < n_local && j < ncols) tile[k][tx] = out[(size_t)i * n_total + col_lo + j];
  }
  __syncthreads();
  for (uint32_t k = ty; k < 32; k += 8) {
    const uint32_t j = j0 + k, i = i0 + tx;
    if (i < n_local && j < ncols) packed[(size_t)j * n_local + i] = tile[tx][k];
  }
}
// apply: from a received block (this rank's rows x the sender's rows [peer_lo, peer_hi)) keep the entries the SENDER's
// rows own -- pair (j, i) with j the sender's row -- the rest of the block is this rank's own work.
__global__ __launch_bounds__(256) void k_mirror_apply(uint64_t* __restrict__ out, uint32_t row_lo, uint32_t n_local, uint32_t n_total,
                                                      uint32_t peer_lo, uint32_t npeer, const uint64_t* __restrict__ recv) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)n_local * npeer) return;
  const uint32_t i = (uint32_t)(t / npeer), j = (uint32_t)(t % npeer);
  if (owns_pair(peer_lo + j, row_lo + i, n_total)) out[(size_t)i * n_total + peer_lo + j] = recv[t];
}
}  // namespace
void launch_mirror_pack(const void* out, uint32_t n_local, uint32_t n_total, uint32_t col_lo, uint32_t col_hi, void* packed, hipStream_t s) {
  const uint32_t nc = col_hi - col_lo;
  if (nc == 0 || n_local == 0) return;
  hipLaunchKernelGGL(k_mirror_pack, dim3((nc + 31) / 32, (n_local + 31) / 32), dim3(256), 0, s, (const uint64_t*)out, n_local, n_total,
                     col_lo, nc, (uint64_t*)packed);
  HIP_CHECK(hipGetLastError());
}
void launch_mirror_apply(void* out, uint32_t row_lo, uint32_t n_local, uint32_t n_total, uint32_t peer_lo, uint32_t peer_hi, const void* recv,
                         hipStream_t s) {
  const uint32_t np = peer_hi - peer_lo;
  if (np == 0 || n_local == 0) return;
  hipLaunchKernelGGL(k_mirror_apply, dim3((unsigned)(((uint64_t)n_local * np + 255) / 256)), dim3(256), 0, s, (uint64_t*)out, row_lo, n_local,
                     n_total, peer_lo, np, (const uint64_t*)recv);
  HIP_CHECK(hipGetLastError());
}

void launch_compare_pair(const uint64_t* A, uint32_t la, const uint64_t* B, uint32_t lb, uint64_t n, PairOut* out_dev,
                         Device& dev, hipStream_t s) {
  const uint64_t total = (uint64_t)la + lb;
  dev.prof_begin(s);
  if (n == 0 && total > 32768) {
    HIP_CHECK(hipMemsetAsync(out_dev, 0, sizeof(PairOut), s));
    // about 64 merged elements per lane, at most 16 workgroups per CU
    uint64_t blocks = (total / 64 + 255) / 256;
    const uint64_t cap = (uint64_t)dev.cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_pair_grid, dim3((unsigned)blocks), dim3(256), 0, s, A, la, B, lb, out_dev);
  } else if (total <= 8192) {
    hipLaunchKernelGGL(k_pair_block<256>, dim3(1), dim3(256), 0, s, A, la, B, lb, n, out_dev);
  } else {
    hipLaunchKernelGGL(k_pair_block<1024>, dim3(1), dim3(1024), 0, s, A, la, B, lb, n, out_dev);
  }
  HIP_CHECK(hipGetLastError());
  dev.prof_end("compare_pair", s);
}

void launch_compare_block(const SketchSet& rows, const SketchSet& cols, uint32_t num,
                          const uint32_t* row_nums, const CompareOut& out, Device& dev,
                          hipStream_t s, uint32_t max_row_len, uint32_t max_col_len, uint64_t nr_elems,
                          uint64_t nc_elems, bool same_sets) {
  const uint64_t npairs = (uint64_t)rows.n * cols.n;
  if (npairs == 0) return;
  const uint32_t route = compare_get_tuning().route;
  // big blocks: dictionary-encode once, then the tiled kernel; small ones: one wavefront per pair
  const bool block_ok = nr_elems + nc_elems > 0;
  if (block_ok && (route == kRouteAuto ? (npairs >= 4096 && rows.n >= 8 && cols.n >= 16)
                                       : (route == kRouteComponents || route == kRouteTiled))) {
    launch_tiled(rows, cols, nr_elems, nc_elems, num, row_nums, out, dev, s, same_sets);
    return;
  }
  // a few against many: the few side sits in LDS, the many side streams
  {
    const bool rows_many = rows.n >= cols.n;
    const SketchSet& many = rows_many ? rows : cols;
    const SketchSet& few = rows_many ? cols : rows;
    const uint32_t few_max = rows_many ? max_col_len : max_row_len;
    if (route == kRouteAuto ? many.n >= 64 : route == kRouteFew) {
      const bool q_lds = few_max <= 8192;
      const size_t lds = q_lds ? (size_t)(few_max ? few_max : 1) * 8 : 16;
      const bool want_cc = out.count_common || out.containment;
      uint32_t gx = (many.n + 3) / 4;
      const uint32_t cap = (uint32_t)dev.cu_count() * 8;
      if (gx > cap) gx = cap;
      dev.prof_begin(s);
#define SMH_CF(L_, C_) hipLaunchKernelGGL((k_compare_few<L_, C_>), dim3(gx, few.n), dim3(256), lds, s, many, few, \
                                          rows_many ? 1u : 0u, num, row_nums, out)
      if (q_lds) { if (want_cc) SMH_CF(true, true); else SMH_CF(true, false); }
      else { if (want_cc) SMH_CF(false, true); else SMH_CF(false, false); }
#undef SMH_CF
      HIP_CHECK(hipGetLastError());
      dev.prof_end("compare_few", s);
      CompareStats rec;
      rec.route = kRouteFew;
      set_stats(rec);
      return;
    }
  }
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
  CompareStats rec;
  rec.route = kRouteWave;
  set_stats(rec);
}

}  // namespace smh
