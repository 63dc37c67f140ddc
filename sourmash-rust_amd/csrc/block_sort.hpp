// block_sort.hpp -- radix passes of up to 8192 u64 keys inside ONE workgroup, keys and payload in LDS (device code shared by
// sort.hip: k_block_sort / k_small_fold, and compare_kernels.hip: the bucketed dictionary of a small pool).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smh {
namespace {

__device__ __forceinline__ uint64_t lanemask_lt() {
  uint32_t lane = __lane_id();
  return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// Up to 8192 keys: the radix passes inside ONE workgroup, keys and their places in LDS, one launch instead of twenty-one.
// Same ranking as k_radix_scatter (lanes holding the same digit found with eight ballots, per-wave digit counters); a
// pass reads every key into registers and -- after the barriers -- writes it back to the same array at its new place.
// One bacterial genome per call (5 Mbp -> 5 059 candidates at scaled=1000) spent 150 of its 245 us in those launches.
constexpr int kBlockSortMax = 8192;                // with the keys' places (payload): 64 + 16 KB of LDS
constexpr int kBsThreads = 1024, kBsWaves = kBsThreads / 64;   // (16 waves hide the LDS latency of a pass better than 8: -4..8 %)
// PT: the payload carried with every key (its original place as u16, or any u32), void-like when !WithIdx
template <int CAP, bool WithIdx, typename PT = uint16_t>
struct BlockSortLds {
  uint64_t sk[CAP];
  PT si[WithIdx ? CAP : 1];
  uint16_t wcount[kBsWaves][256];   // (16 bits are enough: at most CAP keys)
  uint16_t lbase[kBsWaves][256];
  uint32_t wtot[kBsWaves];
  uint32_t skip;
};
// the passes over bits [shift_lo, shift_hi) of L.sk[0 .. items * 1024) (and L.si, the keys' payload, when WithIdx), least
// significant byte first; ends on a barrier.  [0, 64): a full sort; a sub-range: by those bits only (stable).
template <int CAP, bool WithIdx, typename PT>
__device__ __forceinline__ void block_sort_passes(BlockSortLds<CAP, WithIdx, PT>& L, uint32_t n, uint32_t items, int shift_lo = 0,
                                                  int shift_hi = 64) {
  constexpr int kBsItems = CAP / kBsThreads;
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const uint32_t covered = items * kBsThreads;
  const uint64_t lt = lanemask_lt();
  const uint32_t wbase = (uint32_t)w * items * 64;
  for (int shift = shift_lo; shift < shift_hi; shift += 8) {
    for (int i = t; i < kBsWaves * 256; i += kBsThreads) (&L.wcount[0][0])[i] = 0;
    static_assert(CAP <= 65535 + 1, "16-bit counters");
    if (t == 0) L.skip = 0;
    __syncthreads();
    uint64_t key[kBsItems];
    uint32_t meta[kBsItems];   // digit << 16 | rank among the wave's keys with that digit
    PT idx[kBsItems];
#pragma unroll
    for (int i = 0; i < kBsItems; i++) {
      key[i] = 0; meta[i] = 0; idx[i] = 0;
      if ((uint32_t)i < items) {
        const uint32_t pos = wbase + (uint32_t)i * 64 + lane;
        const uint64_t k = L.sk[pos];
        const uint32_t d = (uint32_t)(k >> shift) & 255u;
        key[i] = k;
        if (WithIdx) idx[i] = L.si[pos];
        uint64_t m = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; b++) {
          const uint64_t bal = __ballot((d >> b) & 1);
          m &= ((d >> b) & 1) ? bal : ~bal;
        }
        const uint32_t prior = L.wcount[w][d];
        const uint32_t below = (uint32_t)__popcll(m & lt);
        meta[i] = (d << 16) | (prior + below);
        if (below == 0) L.wcount[w][d] = (uint16_t)(prior + (uint32_t)__popcll(m));
      }
    }
    __syncthreads();
    uint32_t tot = 0, incl = 0;
    if (t < 256) {   // thread d: the keys with digit d
#pragma unroll
      for (int ww = 0; ww < kBsWaves; ww++) tot += L.wcount[ww][t];
      // every real key has this digit (the pads -- all ones, always at the end -- count under digit 255): identity
      if (tot - (t == 255 ? covered - n : 0u) == n) L.skip = 1;
      incl = tot;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      if (lane == 63) L.wtot[w] = incl;
    }
    __syncthreads();
    if (t < 256) {   // where digit d starts, and inside it where each wave's keys go
      uint32_t run = incl - tot;
      for (int ww = 0; ww < w; ww++) run += L.wtot[ww];
#pragma unroll
      for (int ww = 0; ww < kBsWaves; ww++) { L.lbase[ww][t] = (uint16_t)run; run += L.wcount[ww][t]; }
    }
    __syncthreads();
    if (!L.skip) {
#pragma unroll
      for (int i = 0; i < kBsItems; i++)
        if ((uint32_t)i < items) {
          const uint32_t np = L.lbase[w][meta[i] >> 16] + (meta[i] & 0xFFFFu);
          L.sk[np] = key[i];
          if (WithIdx) L.si[np] = idx[i];
        }
    }
    __syncthreads();
  }
}
}  // namespace
}  // namespace smh
