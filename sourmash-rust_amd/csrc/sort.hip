// sort.hip -- hand-written gfx950 primitives behind the sketch "fold" step:
//   * LSD radix sort of u64 keys (optional u64 payload), 8-bit digits, wave64 ballot ranking
//   * run-length encoding of a sorted key array (unique keys + run starts)
//   * segmented reductions over runs (sum of weights, min position, capped count)
//   * single-workgroup exclusive scan used by both
//
// Where this sits in the reference: it is the device-side replacement for the sorted
// Vec<u64> + binary_search + Vec::insert maintenance of KmerMinHash::add_hash
// (reference src/lib.rs:192-245) when hashes arrive in bulk: sort + unique + count gives
// the same `mins` / `abunds` as inserting one by one (DESIGN.md "fold").
//
// Wave = 64 lanes everywhere; masks are 64-bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "block_sort.hpp"
#include "device.hpp"
#include "kernels.hpp"

namespace smh {

namespace {

constexpr int kSortThreads = 256;   // one thread per digit value in the count / scatter kernels
constexpr int kSortItems = 8;                        // keys per thread per tile (16 KB LDS stage: 6 workgroups per CU)
constexpr int kSortTile = kSortThreads * kSortItems;  // 2048 keys per workgroup
constexpr int kWaves = kSortThreads / 64;

// ---------------------------------------------------------------------------------
// all eight digit histograms in one read of the keys (decides which passes can be skipped)
__global__ __launch_bounds__(256) void k_hist_all(const uint64_t* __restrict__ keys, size_t n,
                                                  unsigned long long* __restrict__ ghist) {
  __shared__ uint32_t h[8 * 256];
  for (int i = threadIdx.x; i < 8 * 256; i += blockDim.x) h[i] = 0;
  __syncthreads();
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t k = keys[i];
#pragma unroll
    for (int p = 0; p < 8; p++) atomicAdd(&h[p * 256 + ((k >> (8 * p)) & 255)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 256; i += blockDim.x)
    if (h[i]) atomicAdd(&ghist[i], (unsigned long long)h[i]);
}

// per-workgroup histogram of one digit; layout digit-major so one scan orders everything
// (shift_add, nullable: a device word added to the shift -- the caller's passes start at a bit only the device knows)
__global__ __launch_bounds__(kSortThreads) void k_radix_count(const uint64_t* __restrict__ keys,
                                                              size_t n, int shift,
                                                              uint32_t* __restrict__ blockhist,
                                                              uint32_t nblocks, const uint32_t* __restrict__ shift_add) {
  __shared__ uint32_t h[256];
  if (shift_add) shift += (int)*shift_add;
  if (threadIdx.x < 256) h[threadIdx.x] = 0;
  __syncthreads();
  size_t base = (size_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int i = 0; i < kSortItems; i++) {
    size_t idx = base + (size_t)i * kSortThreads + threadIdx.x;
    if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256) blockhist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of a u32 array, three small launches: (1) every workgroup scans its chunk of
// kScanChunk entries in place and records the chunk total, (2) one workgroup scans the totals,
// (3) every workgroup adds its chunk offset.  Arrays here are 256 * tiles entries (<= a few
// million), so (2) is a single workgroup looping over at most a few hundred totals.
constexpr int kScanThreads = 1024;
constexpr int kScanPer = 8;
constexpr int kScanChunk = kScanThreads * kScanPer;

__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* wtot /*[16]*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wtot[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int i = 0; i < kScanThreads / 64; i++) {
    uint32_t t = wtot[i];
    if (i < w) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_chunks(uint32_t* __restrict__ data, size_t m,
                                                              uint32_t* __restrict__ chunk_tot) {
  __shared__ uint32_t wtot[kScanThreads / 64];
  const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanPer;
  uint32_t v[kScanPer], s = 0;
#pragma unroll
  for (int i = 0; i < kScanPer; i++) { v[i] = base + i < m ? data[base + i] : 0; s += v[i]; }
  uint32_t tot;
  uint32_t run = block_excl_scan_1024(s, wtot, &tot);
#pragma unroll
  for (int i = 0; i < kScanPer; i++) {
    if (base + i < m) data[base + i] = run;
    run += v[i];
  }
  if (threadIdx.x == 0) chunk_tot[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_totals(uint32_t* __restrict__ chunk_tot, uint32_t nchunks,
                                                              uint32_t* __restrict__ total_out) {
  __shared__ uint32_t wtot[kScanThreads / 64];
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nchunks; b0 += kScanThreads) {
    uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < nchunks ? chunk_tot[i] : 0, tot;
    uint32_t ex = block_excl_scan_1024(v, wtot, &tot);
    if (i < nchunks) chunk_tot[i] = carry + ex;
    carry += tot;
  }
  if (total_out && threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_add(uint32_t* __restrict__ data, size_t m,
                                                           const uint32_t* __restrict__ chunk_off) {
  const uint32_t add = chunk_off[blockIdx.x];
  const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanPer;
#pragma unroll
  for (int i = 0; i < kScanPer; i++)
    if (base + i < m) data[base + i] += add;
}

// Offsets of a pass in ONE small launch (tiles <= kRowScanMax): workgroup d scans row d of the digit-major
// per-tile counts in place and leaves the digit's total; the scatter kernel adds the exclusive prefix of
// the 256 totals itself.  (The generic three-launch scan of 256 * tiles counters is pure latency for the
// few thousand tiles of a 2 M-key sort: 15 of the 41 us of a pass.)
constexpr uint32_t kRowScanMax = 16384;
__global__ __launch_bounds__(256) void k_radix_rowscan(uint32_t* __restrict__ blockhist, uint32_t nblocks,
                                                       uint32_t* __restrict__ digit_tot) {
  // 1024 counters per turn, four consecutive ones per lane, the next turn's four requested before this turn's scan (a turn
  // of 256 counters behind its own load: 39 round trips for the 9 766 tiles of 20 M keys, 26 us per pass)
  __shared__ uint32_t wt[2][4];
  uint32_t* row = blockhist + (size_t)blockIdx.x * nblocks;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t carry = 0;
  uint32_t nx[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { const uint32_t i = threadIdx.x * 4u + (uint32_t)k; nx[k] = i < nblocks ? row[i] : 0u; }
  for (uint32_t b0 = 0, turn = 0; b0 < nblocks; b0 += 1024, turn++) {
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = nx[k];
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t i = b0 + 1024u + threadIdx.x * 4u + (uint32_t)k; nx[k] = i < nblocks ? row[i] : 0u; }
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wt[turn & 1u][w] = incl;
    __syncthreads();                                       // (two sets of wave totals: one barrier per turn)
    uint32_t before = 0, tot = 0;
    for (int k = 0; k < 4; k++) { const uint32_t x = wt[turn & 1u][k]; if (k < w) before += x; tot += x; }
    uint32_t run = carry + before + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t i = b0 + threadIdx.x * 4u + (uint32_t)k;
      if (i < nblocks) row[i] = run;
      run += v[k];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) digit_tot[blockIdx.x] = carry;
}

// stable scatter of one digit.  Wave w of the workgroup owns the contiguous sub-tile
// [w*512, (w+1)*512) of the tile and walks it 64 keys per round, so (wave, round, lane)
// order is index order and equal digits keep their relative order.
// The tile is first put in digit order in LDS, then written out by consecutive lanes: a wave's
// store covers a few contiguous digit runs instead of 64 scattered 8-byte pieces (the direct
// scatter ran at 1.6 TB/s of combined traffic, bound by partial-line writes).
// VB: payload bytes per key (0 = keys only, 4 = u32, 8 = u64)
template <int VB>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(
    const uint64_t* __restrict__ kin, uint64_t* __restrict__ kout, const void* __restrict__ vin_,
    void* __restrict__ vout_, size_t n, int shift, const uint32_t* __restrict__ scanned,
    uint32_t nblocks, const uint32_t* __restrict__ digit_tot, const uint32_t* __restrict__ shift_add) {
  __shared__ uint32_t wcount[kWaves][256];
  if (shift_add) shift += (int)*shift_add;
  __shared__ uint32_t wtot2[kWaves];
  __shared__ uint32_t lbase[kWaves][256];   // position in the digit-ordered tile of (wave, digit)'s first key
  __shared__ uint32_t dexcl[256];           // first position of digit d in the digit-ordered tile
  __shared__ uint32_t gbase[256];           // first global position of digit d for this workgroup
  __shared__ uint32_t wtot[kWaves];
  __shared__ __attribute__((aligned(16))) uint64_t stage[kSortTile];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  for (int i = t; i < kWaves * 256; i += kSortThreads) (&wcount[0][0])[i] = 0;
  __syncthreads();

  const size_t tbase = (size_t)blockIdx.x * kSortTile;
  const size_t wbase = tbase + (size_t)w * (kSortItems * 64);
  const uint32_t tile_n = (uint32_t)((n - tbase) < (size_t)kSortTile ? (n - tbase) : (size_t)kSortTile);
  uint64_t key[kSortItems];
  uint32_t meta[kSortItems];  // digit << 16 | rank within the wave's sub-tile
  const uint64_t lt = lanemask_lt();
#pragma unroll
  for (int i = 0; i < kSortItems; i++) {
    size_t idx = wbase + (size_t)i * 64 + lane;
    bool active = idx < n;
    uint64_t k = active ? kin[idx] : 0;
    uint32_t d = (uint32_t)(k >> shift) & 255u;
    key[i] = k;
    // lanes holding the same digit: AND of eight ballots
    uint64_t m = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      uint64_t bal = __ballot(active && ((d >> b) & 1));
      m &= ((d >> b) & 1) ? bal : ~bal;
    }
    uint32_t prior = wcount[w][d];
    uint32_t below = __popcll(m & lt);
    meta[i] = (d << 16) | (prior + below);
    if (active && below == 0) wcount[w][d] = prior + (uint32_t)__popcll(m);
  }
  __syncthreads();
  {
    // thread d resolves digit d (kSortThreads == 256): tile-wide exclusive scan over the digits
    uint32_t tot = 0;
#pragma unroll
    for (int ww = 0; ww < kWaves; ww++) tot += wcount[ww][t];
    uint32_t incl = tot;
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int ww = 0; ww < w; ww++) before += wtot[ww];
    const uint32_t ex = before + incl - tot;
    dexcl[t] = ex;
    // rows scanned one by one (k_radix_rowscan): add the exclusive prefix of the digit totals
    uint32_t dbase = 0;
    if (digit_tot) {
      const uint32_t dt = digit_tot[t];
      uint32_t inc2 = dt;
      for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc2, off);
        if (lane >= off) inc2 += o;
      }
      if (lane == 63) wtot2[w] = inc2;
      __syncthreads();
      for (int ww = 0; ww < w; ww++) dbase += wtot2[ww];
      dbase += inc2 - dt;
    }
    gbase[t] = dbase + scanned[(size_t)t * nblocks + blockIdx.x];
    uint32_t run = ex;
#pragma unroll
    for (int ww = 0; ww < kWaves; ww++) {
      lbase[ww][t] = run;
      run += wcount[ww][t];
    }
  }
  __syncthreads();
  // keys into digit order
#pragma unroll
  for (int i = 0; i < kSortItems; i++) {
    size_t idx = wbase + (size_t)i * 64 + lane;
    if (idx < n) stage[lbase[w][meta[i] >> 16] + (meta[i] & 0xFFFFu)] = key[i];
  }
  __syncthreads();
  uint32_t gpos[kSortItems];
#pragma unroll
  for (int i = 0; i < kSortItems; i++) {
    const uint32_t p = (uint32_t)i * kSortThreads + t;   // consecutive lanes, consecutive positions
    gpos[i] = 0;
    if (p < tile_n) {
      const uint64_t k = stage[p];
      const uint32_t d = (uint32_t)(k >> shift) & 255u;
      gpos[i] = gbase[d] + (p - dexcl[d]);
      kout[gpos[i]] = k;
    }
  }
  if (VB != 0) {
    __syncthreads();   // everyone has read the keys: the stage is reused for the payload
#pragma unroll
    for (int i = 0; i < kSortItems; i++) {
      size_t idx = wbase + (size_t)i * 64 + lane;
      if (idx < n) {
        const uint32_t lp = lbase[w][meta[i] >> 16] + (meta[i] & 0xFFFFu);
        if (VB == 8) stage[lp] = static_cast<const uint64_t*>(vin_)[idx];
        if (VB == 4) reinterpret_cast<uint32_t*>(stage)[lp] = vin_ ? static_cast<const uint32_t*>(vin_)[idx] : (uint32_t)idx;   // (no payload array: the key's place)
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSortItems; i++) {
      const uint32_t p = (uint32_t)i * kSortThreads + t;
      if (p < tile_n) {
        if (VB == 8) static_cast<uint64_t*>(vout_)[gpos[i]] = stage[p];
        if (VB == 4) static_cast<uint32_t*>(vout_)[gpos[i]] = reinterpret_cast<uint32_t*>(stage)[p];
      }
    }
  }
}

// ---------------------------------------------------------------------------------
template <int VB>
__global__ __launch_bounds__(kBsThreads) void k_block_sort(const uint64_t* __restrict__ kin, uint64_t* __restrict__ kout,
                                                           const void* __restrict__ vin_, void* __restrict__ vout_, uint32_t n) {
  __shared__ BlockSortLds<kBlockSortMax, true> L;
  const uint32_t t = threadIdx.x;
  const uint32_t items = (n + kBsThreads - 1) / kBsThreads;          // per lane; the workgroup covers items * 1024 slots
  for (uint32_t i = t; i < items * kBsThreads; i += kBsThreads) {
    L.sk[i] = i < n ? kin[i] : ~0ull;                                // pads sort last (after a real ~0 key: they come later)
    L.si[i] = (uint16_t)i;
  }
  block_sort_passes(L, n, items);
  for (uint32_t i = t; i < n; i += kBsThreads) {
    kout[i] = L.sk[i];
    if (VB == 8) static_cast<uint64_t*>(vout_)[i] = static_cast<const uint64_t*>(vin_)[L.si[i]];
    if (VB == 4) static_cast<uint32_t*>(vout_)[i] = vin_ ? static_cast<const uint32_t*>(vin_)[L.si[i]] : (uint32_t)L.si[i];
  }
}

// The whole fold of a small batch in one launch, without waiting for the candidate count: *count candidates (device) ->
// distinct keys ascending + run starts + {candidates, runs} in result[0..1].  More than kBlockSortMax candidates (or more
// than the buffer holds): result[1] = ~0, nothing else touched -- the caller takes the general path.  This is the call
// shape of one genome per sketch: 5 Mbp leave 5 000 candidates at scaled=1000 (10 000 through the protein arm).
template <int CAP>
__global__ __launch_bounds__(kBsThreads) void k_small_fold(const uint64_t* __restrict__ kin, const unsigned long long* __restrict__ count,
                                                           uint64_t capacity, uint64_t* __restrict__ uniq, uint32_t* __restrict__ starts,
                                                           unsigned long long* __restrict__ result) {
  __shared__ BlockSortLds<CAP, false> L;                         // keys only: 64 or 128 KB + the counters
  const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
  const unsigned long long n64 = *count;
  if (n64 > (unsigned long long)CAP || n64 > capacity) {
    if (t == 0) { result[0] = n64; result[1] = ~0ull; }
    return;
  }
  const uint32_t n = (uint32_t)n64;
  const uint32_t items = (n + kBsThreads - 1) / kBsThreads;
  for (uint32_t i = t; i < items * kBsThreads; i += kBsThreads) L.sk[i] = i < n ? kin[i] : ~0ull;
  if (n) block_sort_passes(L, n, items);
  __syncthreads();
  // run heads: thread t owns slots [t * items, (t + 1) * items)
  const uint32_t lo = t * items;
  uint32_t c = 0;
  for (uint32_t i = lo; i < lo + items && i < n; i++) c += (i == 0 || L.sk[i] != L.sk[i - 1]) ? 1u : 0u;
  uint32_t incl = c;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) L.wtot[w] = incl;
  __syncthreads();
  uint32_t o = incl - c, total = 0;
  for (uint32_t ww = 0; ww < (uint32_t)kBsWaves; ww++) { if (ww < w) o += L.wtot[ww]; total += L.wtot[ww]; }
  for (uint32_t i = lo; i < lo + items && i < n; i++)
    if (i == 0 || L.sk[i] != L.sk[i - 1]) { uniq[o] = L.sk[i]; starts[o] = i; o++; }
  if (t == 0) { result[0] = n64; result[1] = total; }
}

// ---------------------------------------------------------------------------------
// run-length encoding of sorted keys
constexpr int kRleThreads = 256;
constexpr int kRleItems = 8;
constexpr int kRleTile = kRleThreads * kRleItems;

// key2 (optional): a second, more significant key; a run ends where either key changes
__device__ __forceinline__ bool is_head(const uint64_t* keys, const uint64_t* key2, int sh2, size_t i) {
  return i == 0 || keys[i] != keys[i - 1] || (key2 && (key2[i] >> sh2) != (key2[i - 1] >> sh2));
}

__global__ __launch_bounds__(kRleThreads) void k_rle_count(const uint64_t* __restrict__ keys,
                                                           const uint64_t* __restrict__ key2, int sh2,
                                                           size_t n, uint32_t* __restrict__ bc,
                                                           const uint32_t* __restrict__ skip) {
  __shared__ uint32_t wsum[kRleThreads / 64];
  if (skip && *skip) return;   // a device-side plan decided that this result is not needed
  // items of a tile are dealt to the lanes round-robin: every load of a wave is one contiguous 512-byte piece
  const size_t tile0 = (size_t)blockIdx.x * kRleTile;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < kRleItems; i++) {
    const size_t idx = tile0 + (size_t)i * kRleThreads + threadIdx.x;
    if (idx < n && is_head(keys, key2, sh2, idx)) c++;
  }
  for (int off = 32; off; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
    for (int i = 0; i < kRleThreads / 64; i++) s += wsum[i];
    bc[blockIdx.x] = s;
  }
}

// Items dealt round-robin as in k_rle_count (row i of the tile = items i * kRleThreads ...): the number of heads
// before an item is the heads of the rows above it, of the waves before its own in its row, and of the lanes before
// it in its wave (a ballot).  Loads are contiguous per wave and so are the stores of the heads, apart from the gaps
// the repeated keys leave.
// With `origin` / `rank_out` it also scatters the run id of every element back to where the
// element came from (rank_out[origin[i]] = run of sorted position i): the dictionary encoding of
// the compare pre-pass, fused here instead of a second pass over the runs.
__global__ __launch_bounds__(kRleThreads) void k_rle_write(const uint64_t* __restrict__ keys,
                                                           const uint64_t* __restrict__ key2, int sh2,
                                                           uint64_t* __restrict__ uniq2,
                                                           size_t n,
                                                           const uint32_t* __restrict__ bscan,
                                                           uint64_t* __restrict__ uniq,
                                                           uint32_t* __restrict__ starts,
                                                           const uint32_t* __restrict__ origin,
                                                           uint32_t* __restrict__ rank_out,
                                                           const uint32_t* __restrict__ skip,
                                                           uint32_t* __restrict__ runid_out,
                                                           uint32_t self_sum_blocks = 0, uint32_t* __restrict__ nruns_out = nullptr,
                                                           uint32_t rank_flags = 0) {
  constexpr int NW = kRleThreads / 64;
  static_assert(kRleItems * NW <= 64, "k_rle_write: one wave scans the per-row, per-wave head counts");
  __shared__ uint32_t wcnt[kRleItems * NW];
  __shared__ uint32_t before_all[NW];
  if (skip && *skip) return;
  // (with the scatter by origin: every XCD takes one contiguous stretch of the tiles, so that the lines of rank_out -- one
  // per sketch at any time, revisited by the tiles that follow -- are completed inside its L2; workgroups go to the XCDs round-robin)
  uint32_t bx = blockIdx.x;
  if (rank_out) { const uint32_t per = gridDim.x >> 3; if (bx < per * 8u) bx = (bx & 7u) * per + (bx >> 3); }
  // self_sum_blocks != 0: bscan holds the blocks' head COUNTS, not their scan -- a few thousand of them at most: every
  // workgroup adds up the ones before its own (two launches less than scanning them first), the last one also the total
  uint32_t mine_before = 0;
  if (self_sum_blocks) {
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < bx; i += kRleThreads) acc += bscan[i];
    for (int off = 32; off; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) before_all[threadIdx.x >> 6] = acc;
    __syncthreads();
    for (int i = 0; i < NW; i++) mine_before += before_all[i];
    if (nruns_out && bx + 1 == self_sum_blocks && threadIdx.x == 0) *nruns_out = mine_before + bscan[bx];
  }
  const size_t tile0 = (size_t)bx * kRleTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
  uint64_t k[kRleItems];
  uint32_t lp[kRleItems], flags = 0;
#pragma unroll
  for (int i = 0; i < kRleItems; i++) {
    const size_t idx = tile0 + (size_t)i * kRleThreads + threadIdx.x;
    const bool in = idx < n;
    k[i] = in ? keys[idx] : 0;
    const bool head = in && is_head(keys, key2, sh2, idx);
    const uint64_t ball = __ballot(head);
    lp[i] = (uint32_t)__popcll(ball & below);
    if (head) flags |= 1u << i;
    if (lane == 0) wcnt[i * NW + wave] = (uint32_t)__popcll(ball);
  }
  __syncthreads();
  if (threadIdx.x < 64) {                                  // exclusive scan of the kRleItems * NW counts, row-major
    const uint32_t v = threadIdx.x < kRleItems * NW ? wcnt[threadIdx.x] : 0;
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(incl, off);
      if (lane >= off) incl += u;
    }
    if (threadIdx.x < kRleItems * NW) wcnt[threadIdx.x] = incl - v;
  }
  __syncthreads();
  const uint32_t o0 = self_sum_blocks ? mine_before : bscan[bx];
#pragma unroll
  for (int i = 0; i < kRleItems; i++) {
    const size_t idx = tile0 + (size_t)i * kRleThreads + threadIdx.x;
    if (idx >= n) continue;
    const bool head = (flags >> i) & 1u;
    const uint32_t o = o0 + wcnt[i * NW + wave] + lp[i];   // heads before this item
    if (head) {
      uniq[o] = k[i];
      if (key2) uniq2[o] = key2[idx] >> sh2;
      starts[o] = (uint32_t)idx;
    }
    const uint32_t run = o + (head ? 1u : 0u) - 1u;        // the run this item belongs to
    if (rank_out) {
      // rank_flags: bit 31 = the run has more than one element (the hash is shared), bit 30 = this is the run's first element
      // -- what a later owner of the assembled ranks needs to hand out the range masks' bits without counting anything
      uint32_t v = run;
      if (rank_flags) {
        const bool shared = !head || (idx + 1 < n && keys[idx + 1] == k[i]);
        v |= (shared ? 0x80000000u : 0u) | (head ? 0x40000000u : 0u);
      }
      rank_out[origin[idx]] = v;
    }
    if (runid_out) runid_out[idx] = run;                   // the same by sorted position
  }
}

// one thread per run: count, optional weight sum, optional min of the payload
__global__ __launch_bounds__(256) void k_run_reduce(const uint32_t* __restrict__ starts,
                                                    uint32_t nruns, uint32_t total_runs, uint32_t n,
                                                    const uint64_t* __restrict__ weights,
                                                    const uint64_t* __restrict__ pos,
                                                    uint64_t* __restrict__ out_sum,
                                                    uint64_t* __restrict__ out_minpos) {
  uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nruns) return;
  uint32_t lo = starts[u], hi = (u + 1 < total_runs) ? starts[u + 1] : n;
  if (out_sum) {
    uint64_t s = 0;
    if (weights) for (uint32_t i = lo; i < hi; i++) s += weights[i];
    else s = hi - lo;
    out_sum[u] = s;
  }
  if (out_minpos) {
    uint64_t m = ~0ull;
    for (uint32_t i = lo; i < hi; i++) m = pos[i] < m ? pos[i] : m;
    out_minpos[u] = m;
  }
}

}  // namespace

// scan_totals + scan_add in one launch (few chunks): every workgroup sums the totals of the chunks
// before its own; the last one also reports the grand total
__global__ __launch_bounds__(kScanThreads) void k_scan_add_fused(uint32_t* __restrict__ data, size_t m,
                                                                 const uint32_t* __restrict__ chunk_tot, uint32_t nchunks,
                                                                 uint32_t* __restrict__ total_out) {
  __shared__ uint32_t wsum[kScanThreads / 64];
  __shared__ uint32_t off_s;
  uint32_t part = 0;
  for (uint32_t c = threadIdx.x; c < blockIdx.x; c += kScanThreads) part += chunk_tot[c];
  for (int o = 32; o; o >>= 1) part += __shfl_xor(part, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int i = 0; i < kScanThreads / 64; i++) t += wsum[i];
    off_s = t;
    if (total_out && blockIdx.x == nchunks - 1) *total_out = t + chunk_tot[blockIdx.x];
  }
  __syncthreads();
  const uint32_t add = off_s;
  if (blockIdx.x == 0) return;   // nothing to add to the first chunk
  const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanPer;
#pragma unroll
  for (int i = 0; i < kScanPer; i++)
    if (base + i < m) data[base + i] += add;
}

// ---------------------------------------------------------------------------------
// host drivers

// `tmp` must hold ceil(m / kScanChunk) u32 entries
static void exclusive_scan_u32(uint32_t* d, size_t m, uint32_t* total, uint32_t* tmp, hipStream_t s) {
  const uint32_t nchunks = (uint32_t)((m + kScanChunk - 1) / kScanChunk);
  hipLaunchKernelGGL(k_scan_chunks, dim3(nchunks), dim3(kScanThreads), 0, s, d, m, tmp);
  if (nchunks <= 1024) {   // small scans are launch-bound: two launches instead of three
    if (nchunks > 1 || total) hipLaunchKernelGGL(k_scan_add_fused, dim3(nchunks), dim3(kScanThreads), 0, s, d, m, tmp, nchunks, total);
  } else {
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(kScanThreads), 0, s, tmp, nchunks, total);
    hipLaunchKernelGGL(k_scan_add, dim3(nchunks), dim3(kScanThreads), 0, s, d, m, tmp);
  }
  HIP_CHECK(hipGetLastError());
}
static size_t scan_tmp_entries(size_t m) { return (m + kScanChunk - 1) / kScanChunk + 1; }

// pass_mask != 0: run exactly the byte passes whose bit is set (the caller knows which bytes of the
// keys can differ) -- no digit-histogram read-back, so no host synchronisation inside the sort.
// first_in (nullable; needs pass_mask; may be k0 itself): the keys are read from there by the first pass that runs, k0 is only a work buffer, and
// the payload (vbytes 4) starts as every key's index in first_in -- the caller's array stays untouched and no index array is made.
static int radix_sort_impl(uint64_t* k0, uint64_t* k1, void* v0, void* v1, int vbytes, size_t n,
                           DeviceBuffer& scratch, hipStream_t s, int first_pass, int last_pass, uint32_t pass_mask = 0,
                           const uint64_t* first_in = nullptr, const uint32_t* shift_dev = nullptr) {
  if (shift_dev && !pass_mask) throw_internal("radix_sort: a device-side shift needs a pass mask");
  if (first_in && (!pass_mask || vbytes != 4 || !v0)) throw_internal("radix_sort: first_in needs a pass mask and a 32-bit payload");
  if (n < 2) {
    if (first_in && n == 1) {
      if (first_in != k0) HIP_CHECK(hipMemcpyAsync(k0, first_in, 8, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipMemsetAsync(v0, 0, 4, s));
    }
    return 0;
  }
  if (n >= (1ull << 31)) throw_internal("radix_sort_u64: more than 2^31 keys in one call");
  if (n <= (size_t)kBlockSortMax && first_pass == 0 && last_pass == 8) {
    if (first_in) hipLaunchKernelGGL(k_block_sort<4>, dim3(1), dim3(kBsThreads), 0, s, first_in, k1, nullptr, v1, (uint32_t)n);
    else if (v0 && vbytes == 8) hipLaunchKernelGGL(k_block_sort<8>, dim3(1), dim3(kBsThreads), 0, s, k0, k1, v0, v1, (uint32_t)n);
    else if (v0 && vbytes == 4) hipLaunchKernelGGL(k_block_sort<4>, dim3(1), dim3(kBsThreads), 0, s, k0, k1, v0, v1, (uint32_t)n);
    else hipLaunchKernelGGL(k_block_sort<0>, dim3(1), dim3(kBsThreads), 0, s, k0, k1, nullptr, nullptr, (uint32_t)n);
    HIP_CHECK(hipGetLastError());
    return 1;
  }
  const uint32_t nblocks = (uint32_t)((n + kSortTile - 1) / kSortTile);
  const size_t hist_bytes = 8 * 256 * sizeof(unsigned long long);
  const size_t bh_bytes = (size_t)256 * nblocks * sizeof(uint32_t);
  const size_t tmp_bytes = std::max<size_t>(scan_tmp_entries((size_t)256 * nblocks), 256) * sizeof(uint32_t);   // (>= the 256 digit totals)
  scratch.ensure(hist_bytes + bh_bytes + tmp_bytes);
  auto* ghist = (unsigned long long*)scratch.ptr;
  auto* blockhist = (uint32_t*)((char*)scratch.ptr + hist_bytes);
  auto* scan_tmp = (uint32_t*)((char*)scratch.ptr + hist_bytes + bh_bytes);

  unsigned long long hh[8 * 256];
  if (!pass_mask) {
    HIP_CHECK(hipMemsetAsync(ghist, 0, hist_bytes, s));
    int hb = (int)((n + 255) / 256);
    if (hb > 2048) hb = 2048;
    hipLaunchKernelGGL(k_hist_all, dim3(hb), dim3(256), 0, s, k0, n, ghist);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(hh, ghist, hist_bytes, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  }

  int cur = 0;
  uint64_t* kk[2] = {k0, k1};
  void* vv[2] = {v0, v1};
  const uint64_t* src_first = first_in;
  for (int p = first_pass; p < last_pass; p++) {
    bool trivial = false;
    if (pass_mask) trivial = !((pass_mask >> p) & 1u);
    else
      for (int d = 0; d < 256; d++)
        if (hh[p * 256 + d] == n) { trivial = true; break; }
    if (trivial) continue;  // every key has the same digit: the pass is the identity
    if (src_first) {        // the first pass that runs reads the caller's keys and numbers them
      hipLaunchKernelGGL(k_radix_count, dim3(nblocks), dim3(kSortThreads), 0, s, src_first, n, 8 * p, blockhist, nblocks, shift_dev);
      const uint32_t* dt = nullptr;
      if (nblocks <= kRowScanMax) {
        hipLaunchKernelGGL(k_radix_rowscan, dim3(256), dim3(256), 0, s, blockhist, nblocks, scan_tmp);
        dt = scan_tmp;
      } else {
        exclusive_scan_u32(blockhist, (size_t)256 * nblocks, nullptr, scan_tmp, s);
      }
      hipLaunchKernelGGL(k_radix_scatter<4>, dim3(nblocks), dim3(kSortThreads), 0, s, src_first, kk[cur ^ 1], nullptr, vv[cur ^ 1], n, 8 * p,
                         blockhist, nblocks, dt, shift_dev);
      HIP_CHECK(hipGetLastError());
      src_first = nullptr;
      cur ^= 1;
      continue;
    }
    hipLaunchKernelGGL(k_radix_count, dim3(nblocks), dim3(kSortThreads), 0, s, kk[cur], n, 8 * p,
                       blockhist, nblocks, shift_dev);
    const uint32_t* dtot = nullptr;
    if (nblocks <= kRowScanMax) {
      hipLaunchKernelGGL(k_radix_rowscan, dim3(256), dim3(256), 0, s, blockhist, nblocks, scan_tmp);
      dtot = scan_tmp;
    } else {
      exclusive_scan_u32(blockhist, (size_t)256 * nblocks, nullptr, scan_tmp, s);
    }
    if (v0 && vbytes == 8)
      hipLaunchKernelGGL(k_radix_scatter<8>, dim3(nblocks), dim3(kSortThreads), 0, s, kk[cur],
                         kk[cur ^ 1], vv[cur], vv[cur ^ 1], n, 8 * p, blockhist, nblocks, dtot, shift_dev);
    else if (v0 && vbytes == 4)
      hipLaunchKernelGGL(k_radix_scatter<4>, dim3(nblocks), dim3(kSortThreads), 0, s, kk[cur],
                         kk[cur ^ 1], vv[cur], vv[cur ^ 1], n, 8 * p, blockhist, nblocks, dtot, shift_dev);
    else
      hipLaunchKernelGGL(k_radix_scatter<0>, dim3(nblocks), dim3(kSortThreads), 0, s, kk[cur],
                         kk[cur ^ 1], nullptr, nullptr, n, 8 * p, blockhist, nblocks, dtot, shift_dev);
    HIP_CHECK(hipGetLastError());
    cur ^= 1;
  }
  if (src_first) throw_internal("radix_sort: first_in with a pass mask that runs no pass");
  return cur;
}

void small_fold_async(const uint64_t* keys, const unsigned long long* count_dev, uint64_t capacity, uint64_t* uniq, uint32_t* starts,
                      unsigned long long* result_dev, uint32_t expected, hipStream_t s) {
  // the half-size instance keeps 16 keys per lane in registers instead of 32: ~10 us less for the usual few thousand keys
  if (expected * 10 <= (kSmallFoldMax / 2) * 7)
    hipLaunchKernelGGL(k_small_fold<(int)kSmallFoldMax / 2>, dim3(1), dim3(kBsThreads), 0, s, keys, count_dev, capacity, uniq, starts, result_dev);
  else
    hipLaunchKernelGGL(k_small_fold<(int)kSmallFoldMax>, dim3(1), dim3(kBsThreads), 0, s, keys, count_dev, capacity, uniq, starts, result_dev);
  HIP_CHECK(hipGetLastError());
}

int radix_sort_u64(uint64_t* k0, uint64_t* k1, uint64_t* v0, uint64_t* v1, size_t n,
                   DeviceBuffer& scratch, hipStream_t s, int first_pass, int last_pass, uint32_t pass_mask) {
  return radix_sort_impl(k0, k1, v0, v1, 8, n, scratch, s, first_pass, last_pass, pass_mask);
}
int radix_sort_u64_v32(uint64_t* k0, uint64_t* k1, uint32_t* v0, uint32_t* v1, size_t n, DeviceBuffer& scratch,
                       hipStream_t s, uint32_t pass_mask) {
  return radix_sort_impl(k0, k1, v0, v1, 4, n, scratch, s, 0, 8, pass_mask);
}
int radix_sort_u64_place(const uint64_t* keys, uint64_t* k0, uint64_t* k1, uint32_t* v0, uint32_t* v1, size_t n, DeviceBuffer& scratch,
                         hipStream_t s, uint32_t pass_mask, const uint32_t* shift_dev) {
  return radix_sort_impl(k0, k1, v0, v1, 4, n, scratch, s, 0, 8, pass_mask, keys, shift_dev);
}
int radix_sort_u64_keys(uint64_t* k0, uint64_t* k1, size_t n, DeviceBuffer& scratch, hipStream_t s, uint32_t pass_mask) {
  return radix_sort_impl(k0, k1, nullptr, nullptr, 0, n, scratch, s, 0, 8, pass_mask);
}

// the launches of run_length_encode_u64 without the read-back: *nruns_dev receives the number of runs.
// skip (device, nullable): non-zero = do nothing (the caller's device-side plan does not need the result)
void run_length_encode_u64_async(const uint64_t* keys, size_t n, uint64_t* uniq, uint32_t* starts, DeviceBuffer& scratch,
                                 hipStream_t s, const uint32_t* origin, uint32_t* rank_out, uint32_t* nruns_dev,
                                 const uint32_t* skip, uint32_t* runid_out, bool rank_flags) {
  if (n == 0) { HIP_CHECK(hipMemsetAsync(nruns_dev, 0, 4, s)); return; }
  if (rank_flags && n >= (1ull << 30)) throw_internal("run_length_encode_u64: rank flags need fewer than 2^30 keys");
  if (n >= (1ull << 31)) throw_internal("run_length_encode_u64: more than 2^31 keys");
  const uint32_t nblocks = (uint32_t)((n + kRleTile - 1) / kRleTile);
  scratch.ensure((size_t)(nblocks + 1 + scan_tmp_entries(nblocks)) * sizeof(uint32_t));
  auto* bc = (uint32_t*)scratch.ptr;
  hipLaunchKernelGGL(k_rle_count, dim3(nblocks), dim3(kRleThreads), 0, s, keys, (const uint64_t*)nullptr, 0, n, bc, skip);
  if (nblocks <= 4096 && !skip) {
    // (launch-bound sizes: the write kernel adds up the block counts itself and leaves the total -- two launches instead of five)
    hipLaunchKernelGGL(k_rle_write, dim3(nblocks), dim3(kRleThreads), 0, s, keys, (const uint64_t*)nullptr, 0, (uint64_t*)nullptr, n, bc,
                       uniq, starts, origin, rank_out, skip, runid_out, nblocks, nruns_dev, rank_flags ? 1u : 0u);
    HIP_CHECK(hipGetLastError());
    return;
  }
  exclusive_scan_u32(bc, nblocks, bc + nblocks, bc + nblocks + 1, s);
  hipLaunchKernelGGL(k_rle_write, dim3(nblocks), dim3(kRleThreads), 0, s, keys, (const uint64_t*)nullptr, 0, (uint64_t*)nullptr, n, bc,
                     uniq, starts, origin, rank_out, skip, runid_out, 0u, (uint32_t*)nullptr, rank_flags ? 1u : 0u);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(nruns_dev, bc + nblocks, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
}

uint32_t run_length_encode_u64(const uint64_t* keys, size_t n, uint64_t* uniq, uint32_t* starts,
                               DeviceBuffer& scratch, hipStream_t s, const uint32_t* origin, uint32_t* rank_out,
                               const uint64_t* key2, uint64_t* uniq2, int key2_shift) {
  if (n == 0) return 0;
  if (n >= (1ull << 31)) throw_internal("run_length_encode_u64: more than 2^31 keys");
  const uint32_t nblocks = (uint32_t)((n + kRleTile - 1) / kRleTile);
  scratch.ensure((size_t)(nblocks + 1 + scan_tmp_entries(nblocks)) * sizeof(uint32_t));
  auto* bc = (uint32_t*)scratch.ptr;
  hipLaunchKernelGGL(k_rle_count, dim3(nblocks), dim3(kRleThreads), 0, s, keys, key2, key2_shift, n, bc, (const uint32_t*)nullptr);
  exclusive_scan_u32(bc, nblocks, bc + nblocks, bc + nblocks + 1, s);
  hipLaunchKernelGGL(k_rle_write, dim3(nblocks), dim3(kRleThreads), 0, s, keys, key2, key2_shift, uniq2, n, bc, uniq,
                     starts, origin, rank_out, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (uint32_t*)nullptr);
  HIP_CHECK(hipGetLastError());
  uint32_t nruns = 0;
  HIP_CHECK(hipMemcpyAsync(&nruns, bc + nblocks, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  return nruns;
}

// exclusive scan of m counters in place (+ their total), for the compare planner
void exclusive_scan_u32_dev(uint32_t* d, size_t m, uint32_t* total, DeviceBuffer& scratch, hipStream_t s) {
  if (m == 0) { if (total) HIP_CHECK(hipMemsetAsync(total, 0, 4, s)); return; }
  scratch.ensure(scan_tmp_entries(m) * sizeof(uint32_t));
  exclusive_scan_u32(d, m, total, (uint32_t*)scratch.ptr, s);
}

// ---------------------------------------------------------------------------------
// Union of a sketch that lives in HBM (S: ascending distinct hashes, optional u64 counts) with the fold of one more
// batch (D: ascending distinct hashes, optional run starts = counts) -- KmerMinHash::add_hash over the batch
// (reference src/lib.rs:192-245) for a scaled sketch: a new hash is inserted with its count, a present one has its count
// raised.  Both sides are sorted and distinct, so every element's place in the result follows from ranks: D[j] is looked
// up in S (lb_j = #S < D[j]); a present one adds its count to S's in place; the new ones are ranked among themselves by a
// scan (newrank_j) and marked at lb_j, a scan of the marks tells every S[i] how many new hashes precede it.  Two scatters
// write the result -- no sort, every array is read and written once.
__global__ __launch_bounds__(256) void k_union_probe(const uint64_t* __restrict__ S, uint64_t* __restrict__ S_cnt, uint32_t n_s,
                                                     const uint64_t* __restrict__ D, const uint32_t* __restrict__ D_starts,
                                                     const uint64_t* __restrict__ D_cnt, uint32_t n_d,
                                                     uint32_t d_total, uint32_t* __restrict__ lb_out, uint32_t* __restrict__ isnew,
                                                     uint32_t* __restrict__ marks) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_d) return;
  const uint64_t h = D[j];
  uint32_t lo = 0, hi = n_s;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (S[mid] < h) lo = mid + 1; else hi = mid;
  }
  const bool present = lo < n_s && S[lo] == h;
  if (present) {
    if (S_cnt)   // one D element per S element at most: no atomic
      S_cnt[lo] += D_cnt ? D_cnt[j] : (uint64_t)((j + 1 < n_d ? D_starts[j + 1] : d_total) - D_starts[j]);
  } else {
    atomicAdd(&marks[lo], 1u);
  }
  lb_out[j] = lo | (present ? 0u : 0x80000000u);
  isnew[j] = present ? 0u : 1u;
}
__global__ __launch_bounds__(256) void k_union_scatter_old(const uint64_t* __restrict__ S, const uint64_t* __restrict__ S_cnt, uint32_t n_s,
                                                           const uint32_t* __restrict__ before, uint64_t* __restrict__ out,
                                                           uint64_t* __restrict__ out_cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_s) return;
  const uint32_t at = i + before[i + 1];     // new hashes marked at positions <= i precede S[i]
  out[at] = S[i];
  if (out_cnt) out_cnt[at] = S_cnt[i];
}
__global__ __launch_bounds__(256) void k_union_scatter_new(const uint64_t* __restrict__ D, const uint32_t* __restrict__ D_starts,
                                                           const uint64_t* __restrict__ D_cnt, uint32_t n_d, uint32_t d_total, const uint32_t* __restrict__ lb, const uint32_t* __restrict__ newrank,
                                                           uint64_t* __restrict__ out, uint64_t* __restrict__ out_cnt) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_d) return;
  const uint32_t v = lb[j];
  if (!(v & 0x80000000u)) return;
  const uint32_t at = (v & 0x7fffffffu) + newrank[j];
  out[at] = D[j];
  if (out_cnt) out_cnt[at] = D_cnt ? D_cnt[j] : (uint64_t)((j + 1 < n_d ? D_starts[j + 1] : d_total) - D_starts[j]);
}
__global__ __launch_bounds__(256) void k_starts_to_counts(const uint32_t* __restrict__ starts, uint32_t n, uint32_t total,
                                                          uint64_t* __restrict__ counts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) counts[i] = (uint64_t)((i + 1 < n ? starts[i + 1] : total) - starts[i]);
}
void starts_to_counts(const uint32_t* starts, uint32_t n, uint32_t total, uint64_t* counts, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_starts_to_counts, dim3((n + 255) / 256), dim3(256), 0, s, starts, n, total, counts);
  HIP_CHECK(hipGetLastError());
}
void sorted_union_async(const uint64_t* S, uint64_t* S_cnt, uint32_t n_s, const uint64_t* D, const uint32_t* D_starts,
                        const uint64_t* D_cnt, uint32_t n_d, uint32_t d_total, uint64_t* out, uint64_t* out_cnt, uint32_t* n_new_dev, DeviceBuffer& tmp, DeviceBuffer& scratch,
                        hipStream_t s) {
  if (n_d == 0) { HIP_CHECK(hipMemsetAsync(n_new_dev, 0, 4, s)); }
  tmp.ensure(((size_t)2 * n_d + n_s + 1) * 4 + 64);
  uint32_t* lb = tmp.as<uint32_t>();
  uint32_t* newrank = lb + n_d;
  uint32_t* marks = newrank + n_d;
  HIP_CHECK(hipMemsetAsync(marks, 0, ((size_t)n_s + 1) * 4, s));
  if (n_d) {
    hipLaunchKernelGGL(k_union_probe, dim3((n_d + 255) / 256), dim3(256), 0, s, S, S_cnt, n_s, D, D_starts, D_cnt, n_d, d_total, lb, newrank, marks);
    HIP_CHECK(hipGetLastError());
    exclusive_scan_u32_dev(newrank, n_d, n_new_dev, scratch, s);
  }
  exclusive_scan_u32_dev(marks, (size_t)n_s + 1, nullptr, scratch, s);
  if (n_s) hipLaunchKernelGGL(k_union_scatter_old, dim3((n_s + 255) / 256), dim3(256), 0, s, S, S_cnt, n_s, marks, out, out_cnt);
  if (n_d) hipLaunchKernelGGL(k_union_scatter_new, dim3((n_d + 255) / 256), dim3(256), 0, s, D, D_starts, D_cnt, n_d, d_total, lb, newrank, out, out_cnt);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_pos_to_group(uint64_t* __restrict__ pos, uint64_t n,
                                                      const uint64_t* __restrict__ rec_starts, uint32_t nrec,
                                                      const uint32_t* __restrict__ group_of_rec, int keep_bits) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t p = pos[i];
  uint32_t lo = 0, hi = nrec;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (rec_starts[mid] <= p) lo = mid; else hi = mid;
  }
  pos[i] = keep_bits ? (((uint64_t)group_of_rec[lo] << keep_bits) | p) : (uint64_t)group_of_rec[lo];
}
void launch_pos_to_group(uint64_t* pos, uint64_t n, const uint64_t* rec_starts, uint32_t nrec,
                         const uint32_t* group_of_rec, hipStream_t s, int keep_bits) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_pos_to_group, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pos, n, rec_starts, nrec,
                     group_of_rec, keep_bits);
  HIP_CHECK(hipGetLastError());
}

void run_reduce(const uint32_t* starts, uint32_t nruns, uint32_t total_runs, uint32_t n,
                const uint64_t* weights, const uint64_t* pos, uint64_t* out_sum, uint64_t* out_minpos,
                hipStream_t s) {
  if (nruns == 0) return;
  hipLaunchKernelGGL(k_run_reduce, dim3((nruns + 255) / 256), dim3(256), 0, s, starts, nruns, total_runs,
                     n, weights, pos, out_sum, out_minpos);
  HIP_CHECK(hipGetLastError());
}

}  // namespace smh
