// sketch_kernels.hip -- gfx950 kernels for KmerMinHash::add_sequence
// (reference src/lib.rs:252-305) and its helpers:
//
//   k_dna_rolling<K>   DNA arm, ksize <= 128 (2, 4 or 8 32-bit limbs per packed window).  One lane owns a run of R
//                      consecutive k-mer start positions; the tile is read from HBM once with coalesced
//                      16-byte loads and staged in LDS; each lane rolls two 2-bit packed windows (the forward
//                      k-mer, first base least significant, and its complement, first base most significant
//                      = the reverse complement, first base least significant), picks the canonical strand
//                      with one 64-bit compare, takes the first multiply of every murmur word from LDS
//                      product tables indexed by the 2-bit digits (the ASCII bytes are never formed) and
//                      runs the rest of MurmurHash3 x64_128 (first word) on 32-bit halves in registers.
//                      replaces: src/lib.rs:260-267 (+ revcomp 677-689, _checkdna 795-804,
//                      _hash_murmur 33-35) and the `hash <= max_hash` filter of add_hash 198.
//   k_dna_generic      same contract for any ksize, one lane per k-mer, byte-wise.
//   k_first_invalid    first byte outside [ACGTacgt] per record (force=false, lib.rs:268-273); k_first_bad_record
//                      finds the first offending record on the device.
//   k_tile_records     record (and the end of its valid part) of every launch tile's first position, so that no lane
//                      searches all record starts; k_record_stats counts the records of at least ksize bases and the
//                      protein arm's positions for batches of many records (lib.rs:257).
//   k_protein_fused<W> protein arm in one pass: translation + hashing of both strands' windows, no residue
//                      buffer (lib.rs:275-302, 691-793); k_protein_positions rewrites candidate positions.
//   k_translate        six-frame translation into a residue buffer, unknown codons marked (lib.rs:277-301,
//   k_hash_windows     779-793) + every window of `win` residues, skipping dropped codons (lib.rs:289-300):
//                      the two-pass path for batches with non-ASCII bytes, partial ranges, other window lengths.
//   k_hash_segments    murmur64 of whole byte strings (add_word lib.rs:247-250, ffi.rs:15-24).
//   k_synth_dna        benchmark input generator.
//
// All arithmetic is 64-bit integer; results are bit-exact with the reference by construction
// and checked against oracle/ in tests/.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "device.hpp"
#include "kernels.hpp"

namespace smh {
namespace {

// ---------------------------------------------------------------------------------
// MurmurHash3 x64_128 pieces (crate murmurhash3 ~0.0.5 as called at reference src/lib.rs:33-35)
constexpr uint64_t kC1 = 0x87c37b91114253d5ULL;
constexpr uint64_t kC2 = 0x4cf5ad432745937fULL;

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}
__device__ __forceinline__ uint64_t mix_k1(uint64_t k1) { k1 *= kC1; k1 = rotl64(k1, 31); k1 *= kC2; return k1; }
__device__ __forceinline__ uint64_t mix_k2(uint64_t k2) { k2 *= kC2; k2 = rotl64(k2, 33); k2 *= kC1; return k2; }
__device__ __forceinline__ void mm3_block(uint64_t& h1, uint64_t& h2, uint64_t k1, uint64_t k2) {
  h1 ^= mix_k1(k1);
  h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
  h2 ^= mix_k2(k2);
  h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
}
__device__ __forceinline__ uint64_t mm3_finish(uint64_t h1, uint64_t h2, uint64_t len) {
  h1 ^= len; h2 ^= len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  return h1 + h2;  // first word of the digest
}

// byte-at-a-time front end for strings of any length
struct Mm3Stream {
  uint64_t h1, h2, k1, k2;
  uint32_t n;
  __device__ __forceinline__ explicit Mm3Stream(uint64_t seed) : h1(seed), h2(seed), k1(0), k2(0), n(0) {}
  __device__ __forceinline__ void push(uint32_t b) {
    uint32_t r = n & 15u;
    if (r < 8) k1 |= (uint64_t)b << (8 * r);
    else k2 |= (uint64_t)b << (8 * (r - 8));
    n++;
    if ((n & 15u) == 0) { mm3_block(h1, h2, k1, k2); k1 = 0; k2 = 0; }
  }
  __device__ __forceinline__ uint64_t finish() {
    uint32_t rem = n & 15u;
    if (rem > 8) h2 ^= mix_k2(k2);
    if (rem > 0) h1 ^= mix_k1(k1);
    return mm3_finish(h1, h2, n);
  }
};

__device__ __forceinline__ void emit(const CandSink& sink, uint64_t h, uint64_t pos) {
  unsigned long long idx = atomicAdd(sink.count, 1ull);  // hipcc aggregates this per wave
  if (idx < sink.capacity) {
    sink.hash[idx] = h;
    if (sink.pos) sink.pos[idx] = pos;
  }
}

// Survivors are staged in LDS and every workgroup flushes with ONE global atomic + coalesced
// stores: a per-wave global atomic on the single counter word saturates at ~88 M atomics/s
// (MI355X_MICROARCH.md "dequeue") and capped the DNA kernel at 84 G k-mers/s.
struct Stage {
  uint32_t* ctl;   // [0] = count, [2..3] = flush base
  uint64_t* hash;
  uint64_t* pos;   // valid when the sink wants positions
  uint32_t cap;
};
__device__ __forceinline__ void stage_emit(const Stage& st, const CandSink& sink, uint64_t h, uint64_t pos) {
  const uint32_t slot = atomicAdd(&st.ctl[0], 1u);  // LDS atomic
  if (slot < st.cap) { st.hash[slot] = h; if (sink.pos) st.pos[slot] = pos; }
  else emit(sink, h, pos);                          // stage full: straight to the global sink
}
// all threads of the workgroup must call this (it synchronises).  Nothing is written while fewer
// than `at_least` survivors are staged (0 = flush whatever is there).
__device__ __forceinline__ void stage_flush(const Stage& st, const CandSink& sink, int tid, int nthreads,
                                            uint32_t at_least = 0) {
  __syncthreads();
  const uint32_t staged = min(st.ctl[0], st.cap);
  if (staged && staged >= at_least) {
    if (tid == 0) {
      unsigned long long base = atomicAdd(sink.count, (unsigned long long)staged);
      st.ctl[2] = (uint32_t)base; st.ctl[3] = (uint32_t)(base >> 32);
    }
    __syncthreads();
    const uint64_t base = ((uint64_t)st.ctl[3] << 32) | st.ctl[2];
    for (uint32_t e = tid; e < staged; e += nthreads)
      if (base + e < sink.capacity) {
        sink.hash[base + e] = st.hash[e];
        if (sink.pos) sink.pos[base + e] = st.pos[e];
      }
    __syncthreads();
    if (tid == 0) st.ctl[0] = 0;
  }
}

// last record whose start is <= p   (starts has nrec+1 entries, starts[nrec] = total length)
__device__ __forceinline__ uint32_t find_record(const uint64_t* __restrict__ starts, uint32_t nrec,
                                                uint64_t p) {
  uint32_t lo = 0, hi = nrec;  // answer in [lo, hi)
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (starts[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// The tiled kernels need, per lane, the record its run starts in and where that record's valid part ends.  A binary
// search over all record starts is ~log2(nrec) DEPENDENT global loads that every lane of a workgroup waits for at the
// top of every tile (14 at 10 000 records: a tenth of the tile's time), and their misses show up as memory requests.
// One tiny launch looks up the record of every TILE's first position instead, with its end; a lane reads its tile's
// entry (one 16-byte load, the same for the whole workgroup) and is done when the tile lies inside one record; otherwise
// it searches between the entry's two record numbers.
// end of the valid part of record r: vends[r] when given (DNA arm, force=false); a record shorter than min_len adds
// nothing (protein arm, src/lib.rs:257)
__device__ __forceinline__ uint64_t record_valid_end(const uint64_t* __restrict__ starts, const uint64_t* __restrict__ vends,
                                                     uint32_t min_len, uint32_t r) {
  if (vends) return vends[r];
  const uint64_t s0 = starts[r], s1 = starts[r + 1];
  return (min_len && s1 - s0 < min_len) ? s0 : s1;
}
__global__ __launch_bounds__(256) void k_tile_records(const uint64_t* __restrict__ starts, const uint64_t* __restrict__ vends,
                                                      uint32_t nrec, uint32_t min_len, uint64_t base, uint64_t tile,
                                                      uint64_t ntiles, TileRec* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const uint32_t r = find_record(starts, nrec, base + t * tile);
  out[t] = TileRec{r, find_record(starts, nrec, base + (t + 1) * tile), record_valid_end(starts, vends, min_len, r)};
}
// record of position p, known to lie in tile `tix` of the launch, and the end of its valid part
__device__ __forceinline__ uint32_t find_record_in_tile(const SeqBatch& b, uint32_t min_len, uint64_t tix, uint64_t p,
                                                        uint64_t* end) {
  uint32_t lo = 0, hi = b.nrec;                            // answer in [lo, hi)
  if (b.tile_rec) {
    const TileRec t = b.tile_rec[tix];
    if (t.last == t.rec) { *end = t.end; return t.rec; }
    lo = t.rec; hi = t.last + 1;
  }
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (b.starts[mid] <= p) lo = mid; else hi = mid;
  }
  *end = record_valid_end(b.starts, b.vends, min_len, lo);
  return lo;
}

__device__ __forceinline__ uint32_t upper(uint32_t c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }

// ---------------------------------------------------------------------------------
// DNA arm, rolling 2-bit windows, ksize <= 128
// Product tables.  A k-mer reaches murmur as 8-byte words of ASCII letters, and the first thing
// murmur does with a word is multiply it by c1 (k1 words) or c2 (k2 words).  Multiplication
// distributes over the word's two 4-letter halves,  w*c = lo*c + ((hi*c) << 32)  (mod 2^64), and a
// half is one of 256 strings, so the products come from tables indexed by the 2-bit digits the
// kernel already holds: P1[i] = ascii4(i)*c1, P2[i] = ascii4(i)*c2 as u64 (the high half of a word
// needs only the low dword of the entry), plus one small table for the k-mer's last, partial group
// of letters.  This replaces the digit -> ASCII table AND a third of the 64-bit multiplies.
// One copy of the tables per workgroup: replicas (per lane parity ... per 16 lanes) were measured
// and never paid (1, 2: 37.8 ms; 4, 8: 39.7; 16: 50.5 per 10 GB) -- with one copy the entry address
// is (digits << 3) plus an immediate, two full-rate instructions per group.
constexpr int kLutEntries = 256 + 256 + 64 + 1;              // P1, P2, partial group, one zero entry (groups past the k-mer)
constexpr int kLutDwords = kLutEntries * 3;                  // u64 entries (4.5 KiB), then their low dwords once more, packed

// The hash runs on explicit 32-bit halves.  Issue rates on gfx950 (tools/instr_rate.hip,
// profiles/r01_instr_rates.txt): two-operand VALU forms ~100 lanes/clk/CU, everything with three
// operands (v_alignbit, v_add3, v_lshl_add, v_bfe, v_perm), every 32-bit multiply and v_mad_u64_u32
// ~59 -- so the code below is written to need few of the latter: a 64 x 64 -> 64 multiply by a
// constant is one v_mad_u64_u32 + two v_mul_lo_u32 + one v_add3, a 64-bit rotate two v_alignbit,
// `k ^= k >> 33` two plain ops on the halves, and nothing is zero-extended into register pairs.
struct W2 { uint32_t lo, hi; };
__device__ __forceinline__ W2 w2_rotl(W2 x, int r) {   // 0 < r < 64, r != 32
  if (r < 32) return {__builtin_amdgcn_alignbit(x.lo, x.hi, 32 - r), __builtin_amdgcn_alignbit(x.hi, x.lo, 32 - r)};
  return {__builtin_amdgcn_alignbit(x.hi, x.lo, 64 - r), __builtin_amdgcn_alignbit(x.lo, x.hi, 64 - r)};
}
__device__ __forceinline__ W2 w2_mul(W2 x, uint64_t c) {
  const uint32_t cl = (uint32_t)c, ch = (uint32_t)(c >> 32);
  // three v_mad_u64_u32 and one add: t = x.hi*cl; u = x.lo*ch + t (its low dword is the whole cross
  // term); p = x.lo*cl; high dword = p.hi + u.lo.  (The compiler's own choice -- one v_mad_u64_u32,
  // two v_mul_lo_u32 and a v_add3 -- is four half-rate instructions; measured 32.6 -> 32.0 ms per 10 GB.)
  // (Three separate asm statements on purpose.  The scheduler tends to put u right behind t, which
  // costs a wait state (s_nop) per multiply; pinning the order t, p, u inside one statement removes
  // those and is nevertheless 1-2 % slower -- the padding is hidden by the other waves, the lost
  // scheduling freedom is not.  Measured on both kernels, profiles/r02_dna_kernel_steps.txt.)
  uint64_t t, u, p, cy;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(t), "=s"(cy) : "v"(x.hi), "s"(cl));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(u), "=s"(cy) : "v"(x.lo), "s"(ch), "v"(t));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(cy) : "v"(x.lo), "s"(cl));
  // (the add as asm too: left to the compiler it sometimes becomes a 64-bit add of p and {0, u.lo} -- two moves
  // to build the pair and a half-rate v_lshl_add_u64)
  uint32_t hi;
  asm("v_add_u32 %0, %1, %2" : "=v"(hi) : "v"((uint32_t)(p >> 32)), "v"((uint32_t)u));
  return {(uint32_t)p, hi};
}
__device__ __forceinline__ W2 w2_xor(W2 a, W2 b) { return {a.lo ^ b.lo, a.hi ^ b.hi}; }
// 64-bit values the compiler must treat as whole register pairs (an empty asm hides how they were
// put together): otherwise it splits a + b into zero-extended halves, moves and a v_add3
__device__ __forceinline__ uint64_t w2_pair(W2 a) {
  uint64_t v = ((uint64_t)a.hi << 32) | a.lo;
  asm("" : "+v"(v));
  return v;
}
__device__ __forceinline__ W2 w2_split(uint64_t v) { return {(uint32_t)v, (uint32_t)(v >> 32)}; }
__device__ __forceinline__ W2 w2_add(W2 a, W2 b) { return w2_split(w2_pair(a) + w2_pair(b)); }   // one v_lshl_add_u64
// h1 += h2; h2 += h1 (each value becomes a whole pair once)
__device__ __forceinline__ void w2_cross_add(W2& h1, W2& h2) {
  uint64_t a = w2_pair(h1), b = w2_pair(h2);
  a += b; b += a;
  h1 = w2_split(a); h2 = w2_split(b);
}
// x * 5 + c as (x << 2) + x, then + c: two v_lshl_add_u64
__device__ __forceinline__ W2 w2_mul5_add(W2 x, uint64_t c) {
  const uint64_t v = w2_pair(x);
  uint64_t t;
  asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(t) : "v"(v));   // (left to itself the compiler multiplies by 5: two v_mad_u64_u32 and two moves)
  return w2_split(t + c);
}
// fmix64 WITHOUT its last `k ^= k >> 33`.  That step only touches the low dword, and whether the digest
// h = fmix(h1) + fmix(h2) can be <= a threshold is settled by the high dwords up to one carry (open_may_pass):
// the windows that cannot pass -- all but ~1/scaled of them -- skip the last step of both mixes and the 64-bit sum.
__device__ __forceinline__ W2 w2_fmix_open(W2 k) {
  k.lo ^= k.hi >> 1;                                   // k ^= k >> 33
  k = w2_mul(k, 0xff51afd7ed558ccdULL);
  k.lo ^= k.hi >> 1;
  k = w2_mul(k, 0xc4ceb9fe1a85ec53ULL);
  return k;
}
// thr_hi1 = open_thr(thr).  h.hi is a.hi + b.hi or that + 1 (mod 2^32), so h <= thr needs a.hi + b.hi + 1 (mod 2^32)
// <= thr.hi + 1; a superset of the passing windows (exact test: open_finish() <= thr), everything when thr.hi is all ones.
__device__ __forceinline__ uint32_t open_thr(uint64_t thr) {
  const uint32_t hi = (uint32_t)(thr >> 32);
  return hi == 0xffffffffu ? hi : hi + 1u;
}
__device__ __forceinline__ bool open_may_pass(W2 a, W2 b, uint32_t thr_hi1) { return a.hi + b.hi + 1u <= thr_hi1; }
__device__ __forceinline__ uint64_t open_finish(W2 a, W2 b) {
  a.lo ^= a.hi >> 1; b.lo ^= b.hi >> 1;
  return ((((uint64_t)a.hi << 32) | a.lo) + (((uint64_t)b.hi << 32) | b.lo));
}
// murmur64 from premultiplied words: M[w] = word_w * (w even ? c1 : c2)
// The digest is left OPEN: (a, b) with h = open_finish(a, b).
template <int L>
__device__ __forceinline__ void murmur_kmer_pre(const W2 (&M)[2 * L], int K, uint64_t seed, W2 seedv, W2& a, W2& b) {
  W2 h1 = seedv, h2 = seedv;                            // (seedv: the seed's halves in vector registers, see k_dna_rolling)
  const int nblocks = K >> 4, tail = K & 15;
#pragma unroll
  for (int blk = 0; blk < L; blk++) {
    if (blk < nblocks) {
      h1 = w2_xor(h1, w2_mul(w2_rotl(M[2 * blk], 31), kC2));        // rest of mix_k1
      // (rotl(h1, 27) + h2) * 5 + c; in the first block h2 is still the seed: rotl * 5 + (seed * 5 + c)
      if (blk == 0) h1 = w2_mul5_add(w2_rotl(h1, 27), seed * 5 + 0x52dce729u);
      else h1 = w2_mul5_add(w2_add(w2_rotl(h1, 27), h2), 0x52dce729u);
      h2 = w2_xor(h2, w2_mul(w2_rotl(M[2 * blk + 1], 33), kC1));    // rest of mix_k2
      h2 = w2_mul5_add(w2_add(w2_rotl(h2, 31), h1), 0x38495ab5u);
    } else if (blk == nblocks) {
      if (tail > 8) h2 = w2_xor(h2, w2_mul(w2_rotl(M[2 * blk + 1], 33), kC1));
      if (tail > 0) h1 = w2_xor(h1, w2_mul(w2_rotl(M[2 * blk], 31), kC2));
    }
  }
  h1.lo ^= (uint32_t)K; h2.lo ^= (uint32_t)K;          // ^= len (K <= 128)
  w2_cross_add(h1, h2);
  a = w2_fmix_open(h1); b = w2_fmix_open(h2);         // first word of the digest = open_finish(a, b)
}

// KT > 0: ksize fixed at compile time; KT == 0: any ksize the limb count allows, at run time.
// L = 32-bit limbs of a packed window: 2 for ksize <= 32, 4 for ksize <= 64, 8 for ksize <= 128.
// THREADS lanes per workgroup share the product tables; HB = hashes computed together in one straight-line
// block (independent murmur chains the scheduler can interleave).
// PR: thresholds are looked up per record (hp.thr_rec; grouped bottom-num batches) instead of the
// launch-uniform hp.thr
//
// Bookkeeping.  A lane walks its bases four at a time (one dword of the tile).  A group of four is
// CLEAN when all four bases are ACGT, none lies outside the current record's valid part, and all four
// windows that end in it are complete and belong to the lane's run: then there is nothing to decide
// per base.  One unsigned compare tells (g_span), and only groups that are not clean -- the first
// one of a run, the ones around a record boundary or a non-ACGT byte, the last one -- take the
// per-base path below, which produces a 4-bit mask of the windows that may be emitted.
//
// PK: the tile is staged PACKED -- two bits per base (the 2-bit code the rolling needs anyway: the upper-casing, the
// encoding and the validity check are done once, by the lane that stages the bytes, 16 at a time) plus one "dirty" bit per
// source dword that holds a byte other than ACGT.  A lane's run of 128 bases is 32 bytes of LDS instead of 128: a 512-lane
// workgroup needs ~28 KB instead of ~73 KB, so FOUR workgroups share a CU (8 waves per SIMD instead of 4: the kernel waits
// on its own dependent murmur chains in a third of its wave cycles), and the hot path reads one LDS dword per
// 16 bases and decodes nothing.  Dirty dwords end the lane's window of clean groups like a record boundary does; the group
// then re-reads its four bytes from global memory (rare) and takes the per-base path with the same conditions as ever.
template <int KT, int THREADS, int HB, int L, bool PR = false, bool PK = false, int MINW = (PK ? 8 : 4)>
__global__ __launch_bounds__(THREADS, MINW) void k_dna_rolling(SeqBatch b, HashParams hp, CandSink sink,
                                                                     int logR, uint32_t stage_cap) {
  // LDS: static: the product tables (6.8 KiB; a compile-time address, so a table read is one
  // ds_read with the table's base as its immediate offset); dynamic: [staged candidates: count,
  // hashes, positions][sequence tile]
  __shared__ __attribute__((aligned(16))) uint32_t lut[kLutDwords];
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* st_ctl = smem;                                    // [0] = count, [2..3] = flush base
  uint64_t* st_hash = reinterpret_cast<uint64_t*>(st_ctl + 4);
  uint64_t* st_pos = st_hash + stage_cap;
  uint32_t* tile = reinterpret_cast<uint32_t*>(st_pos + (sink.pos ? stage_cap : 0));
  const Stage stage{st_ctl, st_hash, st_pos, stage_cap};

  const int K = KT ? KT : (int)hp.ksize;
  const int tid = threadIdx.x;
  const uint32_t R = 1u << logR;
  const uint64_t TILE = (uint64_t)THREADS << logR;
  // 2K-bit mask, limb by limb
  uint32_t MASK[L];
#pragma unroll
  for (int i = 0; i < L; i++) {
    const int bits = 2 * K - 32 * i;
    MASK[i] = bits >= 32 ? 0xffffffffu : (bits <= 0 ? 0u : ((1u << bits) - 1u));
  }
  const int top_limb = (2 * K - 2) >> 5, top_sh = (2 * K - 2) & 31;
  const bool multi = b.starts != nullptr;

  // product tables (see kLutEntries): entry e at u64 index e.
  // digit d -> "ACGT"[d], first digit in the low byte; the partial group holds the k-mer's last
  // K mod 4 letters (its multiplier follows the parity of the word it belongs to)
  const int g_last = (K - 1) >> 2, nb_last = K - 4 * g_last;      // nb_last == 4: no partial group
  const uint64_t c_last = ((g_last >> 1) & 1) ? kC2 : kC1;
  uint64_t* ptab = reinterpret_cast<uint64_t*>(lut);
  for (int e = tid; e < kLutEntries; e += THREADS) {
    const uint32_t ent = (uint32_t)e;
    const uint32_t idx = ent & 255u;
    const int nb = ent < 512 ? 4 : (ent < 576 ? nb_last : 0);
    uint32_t v = 0;
    for (int j = 0; j < nb; j++) v |= ((0x54474341u >> (8 * ((idx >> (2 * j)) & 3))) & 0xffu) << (8 * j);
    const uint64_t prod = (uint64_t)v * (ent < 256 ? kC1 : (ent < 512 ? kC2 : c_last));
    ptab[e] = prod;
    // The high half of a word needs only the product's low dword.  Read out of the 8-byte entries those look-ups
    // touch the even banks only; the packed copy spreads them over all 64 (SQ_LDS_BANK_CONFLICT 43 -> ... per 64 k-mers).
    lut[2 * kLutEntries + e] = (uint32_t)prod;
  }
  // table base (bytes) of every 4-letter group: a compile-time constant for a compile-time k, fixed for
  // the launch otherwise, so that the per-k-mer code is the same straight line either way
  uint32_t gbase[4 * L];
#pragma unroll
  for (int g = 0; g < 4 * L; g++) {
    const int nb = K - 4 * g;
    uint32_t ent0 = 576u;                                              // past the k-mer: the zero entry
    if (nb >= 4) ent0 = ((g >> 1) & 1) ? 256u : 0u;
    else if (nb > 0) ent0 = 512u;
    gbase[g] = (g & 1) ? 8u * kLutEntries + ent0 * 4u : ent0 * 8u;   // odd groups (high halves): the packed low dwords
  }
  if (tid == 0) st_ctl[0] = 0;

  const uint64_t span = hp.range_hi - hp.range_lo;
  const uint64_t ntiles = (span + TILE - 1) / TILE;
  const uintptr_t gend = ((uintptr_t)(b.seq + b.len) + 15) & ~(uintptr_t)15;
  const uint32_t nsteps = (R + (uint32_t)K - 1 + 3) & ~3u;  // bases walked per lane, multiple of 4
  const uint32_t warm_end = K > 4 ? ((uint32_t)(K - 1) & ~3u) : 0u;   // groups [0, warm_end) end before any window is complete

  for (uint64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    const uint64_t T0 = hp.range_lo + tix * TILE;
    const uint64_t thr = hp.thr;
    const uint32_t thr_hi1 = open_thr(hp.thr);
    // An operand in a scalar register halves the issue rate of a plain two-operand instruction (tools/instr_rate.hip):
    // what the hash xors in per k-mer lives in vector registers.
    W2 seedv{(uint32_t)hp.seed, (uint32_t)(hp.seed >> 32)};
    asm volatile("" : "+v"(seedv.lo), "+v"(seedv.hi));
    uint64_t lthr = hp.thr;   // PR: threshold of the record the lane is in

    // ---- stage [T0, T0 + TILE + K - 1) in LDS: aligned 16-byte loads, one pad dword per R bytes
    const uintptr_t g0 = (uintptr_t)(b.seq + T0);
    const uintptr_t ga = g0 & ~(uintptr_t)15;
    const uint32_t m = (uint32_t)(g0 - ga);
    const uint32_t nchunks = (m + (uint32_t)TILE + 16 * L + 8 + 15 + (PK ? 32u : 0u)) >> 4;  // last lane reads < m+TILE+K+7 (PK: a dword ahead)
    const uint32_t nchunks_cap = (15u + (uint32_t)TILE + 16 * L + 8 + 15 + 32u) >> 4;     // (the same for the largest m: where the dirty bits start)
    __syncthreads();  // tables ready / previous tile fully consumed
    uint32_t* dirty = tile + (nchunks_cap + (nchunks_cap >> (logR - 4)) + 2);     // PK: one bit per source dword, behind the codes
    if (PK) {
      for (uint32_t c = tid; c < (nchunks >> 3) + 3; c += THREADS) dirty[c] = 0;
      if (tid == 0) st_ctl[1] = 0;                                   // "some dword of this tile is dirty"
      __syncthreads();
    }
    for (uint32_t c = tid; c < nchunks; c += THREADS) {
      uintptr_t addr = ga + ((uintptr_t)c << 4);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (addr < gend) v = *reinterpret_cast<const uint4*>(addr);
      if (PK) {
        // 16 bases -> 32 bits of codes (base j of the chunk in bits 2j, 2j+1) + 4 dirty bits
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
        uint32_t codes = 0, bad = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t u4 = w4[q] & 0xDFDFDFDFu;
          const uint32_t c2 = (u4 >> 1) & 0x03030303u;
          const uint32_t code4 = c2 ^ ((c2 >> 1) & 0x01010101u);
          const uint32_t exp4 = __builtin_amdgcn_perm(0u, 0x54474341u, code4);
          bad |= (u4 != exp4) ? (1u << q) : 0u;
          const uint32_t pk = code4 | (code4 >> 6);                   // bytes 0,1 -> bits 0..3; bytes 2,3 -> bits 16..19
          codes |= ((pk & 0xfu) | ((pk >> 12) & 0xf0u)) << (8 * q);
        }
        tile[c + (c >> (logR - 4))] = codes;                         // one pad dword per run: lane-strided reads hit distinct banks
        if (bad) { atomicOr(&dirty[c >> 3], bad << (4u * (c & 7u))); st_ctl[1] = 1; }
      } else {
        uint32_t x = c << 4;
        uint32_t o = (x >> 2) + (x >> logR);
        tile[o] = v.x; tile[o + 1] = v.y; tile[o + 2] = v.z; tile[o + 3] = v.w;
      }
    }
    __syncthreads();

    const uint64_t p0 = T0 + ((uint64_t)tid << logR);  // my first k-mer start position
    if (p0 < hp.range_hi) {
    const uint32_t nk = (hp.range_hi - p0) < R ? (uint32_t)(hp.range_hi - p0) : R;
    const uint32_t hi_ok = nk + (uint32_t)K - 1;        // base index i ends a window of my run iff K-1 <= i < hi_ok

    // record bookkeeping: `lim` = index of the first base of my run that is not inside the
    // current record's valid part
    uint32_t rec = 0;
    uint64_t cur_end = b.vend0;
    if (multi) {
      rec = find_record_in_tile(b, 0, tix, p0, &cur_end);
      if (PR) lthr = hp.thr_rec[rec];
    }
    uint32_t lim = 0;
    if (cur_end > p0) lim = (cur_end - p0) > 0xfffffffeull ? 0xffffffffu : (uint32_t)(cur_end - p0);
    uint32_t vstart = 0;      // first base of the run of valid bases that reaches the current base
    // clean groups: g_lo <= i0 < g_lo + g_span  (see the kernel comment)
    uint32_t g_lo = 0, g_span = 0;
    // PK: bit g = group g of my run touches a source dword that holds a byte other than ACGT (dirty bits of the tile,
    // one per aligned dword of the input; a group whose bases straddle two dwords looks at both)
    uint64_t dmask = 0;
    if (PK && st_ctl[1]) {
      const uint32_t b0 = (m + ((uint32_t)tid << logR)) >> 2;
      const uint32_t* dw = reinterpret_cast<const uint32_t*>(tile) + (nchunks_cap + (nchunks_cap >> (logR - 4)) + 2);
      const uint32_t i = b0 >> 5, sft = b0 & 31u;
      const uint64_t lo64 = ((uint64_t)dw[i + 1] << 32) | dw[i];
      const uint32_t d2 = dw[i + 2];
      dmask = (lo64 >> sft) | (sft ? ((uint64_t)d2 << (64 - sft)) : 0ull);
      if (m & 3u) dmask |= (dmask >> 1) | ((uint64_t)((d2 >> sft) & 1u) << 63);
    }
    auto set_clean_window = [&](bool warm, uint32_t from_i) {
      uint32_t lim2 = warm ? lim : (lim < hi_ok ? lim : hi_ok);
      if (PK) {
        const uint32_t g = from_i >> 2;
        const uint64_t rem = g < 64 ? (dmask >> g) : 0ull;
        if (rem) lim2 = min(lim2, (g + (uint32_t)__builtin_ctzll(rem)) * 4u);      // clean groups end before the next dirty one
      }
      g_lo = warm ? vstart : vstart + (uint32_t)K - 1;   // warm-up groups emit nothing: only validity matters
      g_span = lim2 >= g_lo + 4 ? lim2 - 3 - g_lo : 0u;
    };

    // tile address of the lane's current dword: one pad dword per R bytes; the running form adds 4
    // per group, and 4 more when the group crosses a multiple of R -- which it does for every lane
    // of the workgroup at once (lane runs start R apart), so the increment is a scalar
    const uint32_t xu = m & ~3u;                                     // uniform part of the lane's byte index
    uint32_t ta = xu + ((uint32_t)tid << logR);
    ta = ta + ((ta >> logR) << 2);                                   // byte offset in the padded tile
    const uint32_t sh = m & 3u;
    uint32_t cur = PK ? 0u : tile[ta >> 2];
    // PK: the codes of my run: chunk (16 bases, one dword) cq holds my base 16 * (cq - cq0) - (m & 15); w = the 16 bases of
    // the current quad of groups, aligned (base j of the quad in bits 2j), wc = their complements; both move down 8 bits per group
    uint32_t cq = (m >> 4) + ((uint32_t)tid << (logR - 4));
    const uint32_t sh2 = (m & 15u) * 2u;
    uint32_t pcur = PK ? tile[cq + (cq >> (logR - 4))] : 0u, w = 0, wc = 0;
    // two rolled windows of 2-bit digits in L limbs.  fle: the forward k-mer, first base least
    // significant.  cf: the COMPLEMENT of the forward k-mer, first base most significant -- which is
    // the reverse complement with ITS first base least significant.  Both candidates for the hashed
    // strand are therefore at hand as they are (first base low), and the reference's `kmer < rc`
    // (src/lib.rs:263; ASCII order A<C<G<T equals digit order) needs no complementing either: with
    // M = 4^k - 1, forward read first-base-most-significant is M - cf and the reverse complement read
    // that way is M - fle, so  kmer < rc  <=>  M - cf < M - fle  <=>  fle < cf.
    uint32_t cf[L], fle[L];
#pragma unroll
    for (int i = 0; i < L; i++) { cf[i] = 0; fle[i] = 0; }

    // one group of four bases; hashing = false for the warm-up groups
    auto group = [&](uint32_t i0, auto hashing) {
      constexpr bool kHash = decltype(hashing)::value;
      uint32_t code4 = 0, ccode4 = 0, diff4 = 0;
      if (PK) {
        if ((i0 & 12u) == 0) {                                       // uniform: a new dword of codes every 16 bases
          cq += 1;
          const uint32_t pnxt = tile[cq + (cq >> (logR - 4))];
          w = __builtin_amdgcn_alignbit(pnxt, pcur, sh2);
          pcur = pnxt;
          wc = ~w;
        }
      } else {
        const uint32_t xn = xu + i0 + 4;                             // uniform: byte index of the next dword
        ta += (xn & (R - 1)) == 0 ? 8u : 4u;
        const uint32_t nxt = tile[ta >> 2];
        const uint32_t d = __builtin_amdgcn_alignbyte(nxt, cur, sh);
        cur = nxt;
        // four bases at once: upper-case, 2-bit code (A0 C1 G2 T3), validity by re-encoding
        const uint32_t u4 = d & 0xDFDFDFDFu;
        const uint32_t c2 = (u4 >> 1) & 0x03030303u;
        code4 = c2 ^ ((c2 >> 1) & 0x01010101u);
        const uint32_t exp4 = __builtin_amdgcn_perm(0u, 0x54474341u, code4);
        diff4 = u4 ^ exp4;
        ccode4 = code4 ^ 0x03030303u;                                // complement digits
      }
      uint32_t okmask = 0xFu;                                        // windows ending at base q that may be emitted
      uint64_t tq[4];                                                // PR only: the threshold in force at each base
      if (PR) { tq[0] = lthr; tq[1] = lthr; tq[2] = lthr; tq[3] = lthr; }
      if (!((PK || diff4 == 0) && i0 - g_lo < g_span)) {
        if (PK && i0 < 256u && ((dmask >> (i0 >> 2)) & 1ull)) {
          // a dirty group: its four bytes again, from global memory, for the per-base validity below
          uint32_t d = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint64_t at = p0 + i0 + (uint32_t)q;
            if (at < b.len) d |= (uint32_t)b.seq[at] << (8 * q);
          }
          const uint32_t u4 = d & 0xDFDFDFDFu;
          const uint32_t c2 = (u4 >> 1) & 0x03030303u;
          const uint32_t cd = c2 ^ ((c2 >> 1) & 0x01010101u);
          diff4 = u4 ^ __builtin_amdgcn_perm(0u, 0x54474341u, cd);
        }
        // ---- not clean (rare): base by base, exactly the reference's conditions
        okmask = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t i = i0 + q;
          uint32_t bad = (diff4 >> (8 * q)) & 0xffu;
          if (i >= lim) {
            // at or past the end of the record's valid part: find where base p0+i belongs
            const uint64_t qpos = p0 + i;
            if (multi) {
              while (rec + 1 < b.nrec && qpos >= b.starts[rec + 1]) { rec++; vstart = i; }
              cur_end = b.vends ? b.vends[rec] : b.starts[rec + 1];
              if (PR) lthr = hp.thr_rec[rec];
            }
            if (qpos >= cur_end) { bad = 1; lim = i + 1; }
            else lim = (cur_end - p0) > 0xfffffffeull ? 0xffffffffu : (uint32_t)(cur_end - p0);
          }
          if (bad) vstart = i + 1;
          if (kHash) {
            const bool ok = (i + 1 >= vstart + (uint32_t)K) && (i < hi_ok);
            okmask |= ok ? (1u << q) : 0u;
            if (PR) tq[q] = lthr;
          }
        }
        set_clean_window(!kHash, i0 + 4);
      }
#pragma unroll
      for (int g0b = 0; g0b < 4; g0b += HB) {
        uint32_t X[HB][L];
#pragma unroll
        for (int q = 0; q < HB; q++) {
          const int bb = g0b + q;
          const uint32_t code = PK ? (w >> (2 * bb)) & 3u : (code4 >> (8 * bb)) & 3u;
          const uint32_t ccode = PK ? (wc >> (2 * bb)) & 3u : (ccode4 >> (8 * bb)) & 3u;
          // cf = ((cf << 2) | (3 - code)) & MASK ; fle = (fle >> 2) | code << (2K-2)   (limb-wise)
#pragma unroll
          for (int li = L - 1; li > 0; li--) cf[li] = __builtin_amdgcn_alignbit(cf[li], cf[li - 1], 30) & MASK[li];
          cf[0] = ((cf[0] << 2) | ccode) & MASK[0];
#pragma unroll
          for (int li = 0; li < L - 1; li++) fle[li] = __builtin_amdgcn_alignbit(fle[li + 1], fle[li], 2);
          fle[L - 1] >>= 2;
#pragma unroll
          for (int li = 0; li < L; li++)
            if (li == top_limb) fle[li] |= code << top_sh;
          if (kHash) {
            // canonical strand: kmer < rc  <=>  fle < cf (see above), decided from the top limb down
            bool fwd = false;
            if (L == 2) {
              fwd = (((uint64_t)fle[1] << 32) | fle[0]) < (((uint64_t)cf[1] << 32) | cf[0]);
            } else {
              bool decided = false;
#pragma unroll
              for (int li = L - 1; li >= 0; li--) {
                const uint32_t f = fle[li], r = cf[li];
                if (!decided && f != r) { fwd = f < r; decided = true; }
              }
            }
#pragma unroll
            for (int li = 0; li < L; li++) X[q][li] = fwd ? fle[li] : cf[li];  // chosen strand, first base low
          }
        }
        if (kHash && i0 + g0b + HB >= (uint32_t)K) {  // uniform: some window of this block is complete
          W2 ha[HB], hb[HB];
#pragma unroll
          for (int q = 0; q < HB; q++) {
            W2 M[2 * L];
#pragma unroll
            for (int wi = 0; wi < 2 * L; wi++) M[wi] = W2{0u, 0u};
#pragma unroll
            for (int g = 0; g < 4 * L; g++) {
              if (KT == 0 || 4 * g < K) {
                // byte offset of the group's table entry: its digits << 3 (entries are u64), masked to
                // the digits that belong to the k-mer; P1 / P2 by word parity, the partial table, or
                // (run-time k only) the zero entry
                const uint32_t xw = X[q][g >> 2];
                // byte (g & 3) of the limb, times 8, in ONE sub-dword-addressed shift (SDWA issues at full
                // rate on gfx950: 32.0 -> 30.8 ms per 10 GB).  No mask is needed, for any k: both windows are
                // zero above their 2k bits, so a partial group indexes inside its 4^nb-entry table and a
                // group past the k-mer reads entry 0 of the zero table.
                uint32_t off;                                        // digits << 3 (8-byte entries) or << 2 (packed low dwords)
                if ((g & 3) == 0) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(xw));
                else if ((g & 3) == 1) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "v"(xw));
                else if ((g & 3) == 2) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "v"(xw));
                else asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(xw));
                const char* at = reinterpret_cast<const char*>(lut) + gbase[g] + off;
                if ((g & 1) == 0) {                                  // low half of the word: full product
                  const uint2 e = *reinterpret_cast<const uint2*>(at);
                  M[g >> 1] = W2{e.x, e.y};
                } else {
                  M[g >> 1].hi += *reinterpret_cast<const uint32_t*>(at);   // high half: (entry << 32), low dword only
                }
              }
            }
            murmur_kmer_pre<L>(M, K, hp.seed, seedv, ha[q], hb[q]);
          }
#pragma unroll
          for (int q = 0; q < HB; q++)
            if (open_may_pass(ha[q], hb[q], PR ? open_thr(tq[g0b + q]) : thr_hi1)) {   // ~1 in `scaled` windows gets here
              uint32_t om = okmask;
              asm volatile("" : "+v"(om));                  // keeps the mask test inside this rare block
              const uint64_t h = open_finish(ha[q], hb[q]);
              if (h <= (PR ? tq[g0b + q] : thr) && ((om >> (g0b + q)) & 1u))
                stage_emit(stage, sink, h, hp.pos_base + p0 + (i0 + g0b + q + 1 - (uint32_t)K));
            }
        }
      }
      if (PK) { w >>= 8; wc >>= 8; }
    };

    set_clean_window(true, 0);
    uint32_t i0 = 0;
    for (; i0 < warm_end; i0 += 4) group(i0, std::false_type{});
    set_clean_window(false, i0);
    for (; i0 < nsteps; i0 += 4) group(i0, std::true_type{});
    }  // p0 < range_hi

    // ---- flush the staged candidates: ONE global atomic per tile, coalesced stores
    stage_flush(stage, sink, tid, THREADS);
  }
}

// ---------------------------------------------------------------------------------
// DNA arm, any ksize: one lane per k-mer start position
__global__ __launch_bounds__(256) void k_dna_generic(SeqBatch b, HashParams hp, CandSink sink) {
  const uint64_t K = hp.ksize;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t p = hp.range_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < hp.range_hi;
       p += stride) {
    uint64_t end = b.vend0, thr = hp.thr;
    if (b.starts) {
      uint32_t r = find_record(b.starts, b.nrec, p);
      end = b.vends ? b.vends[r] : b.starts[r + 1];
      if (hp.thr_rec) thr = hp.thr_rec[r];
    }
    if (p + K > end || p + K < p) continue;
    const uint8_t* s = b.seq + p;
    bool ok = true, decided = false, fwd = true;
    for (uint64_t i = 0; i < K; i++) {
      uint32_t f = upper(s[i]);
      if (!(f == 'A' || f == 'C' || f == 'G' || f == 'T')) { ok = false; break; }
      if (!decided) {
        uint32_t t = upper(s[K - 1 - i]);
        uint32_t rc = t == 'A' ? 'T' : t == 'T' ? 'A' : t == 'C' ? 'G' : t == 'G' ? 'C' : t;
        if (f != rc) { fwd = f < rc; decided = true; }
      }
    }
    if (!ok) continue;
    Mm3Stream st(hp.seed);
    for (uint64_t i = 0; i < K; i++) {
      uint32_t c;
      if (fwd) c = upper(s[i]);
      else {
        uint32_t t = upper(s[K - 1 - i]);
        c = t == 'A' ? 'T' : t == 'T' ? 'A' : t == 'C' ? 'G' : 'C';
      }
      st.push(c);
    }
    uint64_t h = st.finish();
    if (h <= thr) emit(sink, h, hp.pos_base + p);
  }
}

// ---------------------------------------------------------------------------------
// force == false: the smallest position of a byte outside [ACGTacgt] per record (atomicMin into vends, pre-filled with
// the record ends).  16 bytes per lane per step, aligned, validity four bytes at a time as in k_dna_rolling (10 GB in
// ~3 ms; the byte-at-a-time version took 11).
__global__ __launch_bounds__(256) void k_first_invalid(SeqBatch b, uint64_t* __restrict__ vends) {
  const uintptr_t p0 = (uintptr_t)b.seq, pa = p0 & ~(uintptr_t)15;
  const uint64_t head = p0 - pa;
  const uint64_t nch = (head + b.len + 15) >> 4;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nch; c += stride) {
    const uint4 v = *reinterpret_cast<const uint4*>(pa + (c << 4));
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t u4 = w[k] & 0xDFDFDFDFu;
      const uint32_t c2 = (u4 >> 1) & 0x03030303u;
      const uint32_t code4 = c2 ^ ((c2 >> 1) & 0x01010101u);
      const uint32_t diff4 = u4 ^ __builtin_amdgcn_perm(0u, 0x54474341u, code4);
      if (diff4 == 0) continue;
      for (int j = 0; j < 4; j++) {
        if (((diff4 >> (8 * j)) & 0xffu) == 0) continue;
        const uint64_t at = (c << 4) + 4 * k + j;            // offset from pa
        if (at < head || at - head >= b.len) continue;       // a byte of the 16-byte granule outside the batch
        const uint64_t q = at - head;
        const uint32_t r = b.starts ? find_record(b.starts, b.nrec, q) : 0;
        atomicMin((unsigned long long*)&vends[r], (unsigned long long)q);
      }
    }
  }
}
// the first record, in order, that is at least ksize long and whose valid part ends before its end (UINT64_MAX: none)
__global__ __launch_bounds__(256) void k_first_bad_record(const uint64_t* __restrict__ starts, const uint64_t* __restrict__ vends,
                                                          uint32_t nrec, uint32_t ksize, unsigned long long* __restrict__ out) {
  unsigned long long best = ~0ull;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t st = starts[r], en = starts[r + 1];
    if (vends[r] < en && en - st >= ksize && r < best) best = r;
  }
  for (int off = 32; off; off >>= 1) {
    const unsigned long long o = __shfl_down(best, off);
    best = o < best ? o : best;
  }
  if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(out, best);
}

// ---------------------------------------------------------------------------------
// protein arm, phase 1: translate.  Residue buffer layout: for record r, segments
// 6r+0..6r+5 = (frame 0 fwd, frame 0 rc, frame 1 fwd, frame 1 rc, frame 2 fwd, frame 2 rc),
// the order the reference walks them; an unknown codon becomes kDropped.
constexpr uint32_t kDropped = 0xFFu;
__constant__ char kCodonAA[65] =
    "FFLLSSSSYY**CC*W" "LLLLPPPPHHQQRRRR" "IIIMTTTTNNKKSSRR" "VVVVAAAADDEEGGGG";  // T,C,A,G order

__device__ __forceinline__ int tcag(uint32_t c) {
  return c == 'T' ? 0 : c == 'C' ? 1 : c == 'A' ? 2 : c == 'G' ? 3 : -1;
}
__device__ __forceinline__ uint32_t comp_upper(uint32_t c) {
  return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
}
// str::from_utf8 on a 3-byte chunk (reference src/lib.rs:787 unwraps it)
__device__ __forceinline__ bool utf8_ok3(uint32_t a, uint32_t c, uint32_t d) {
  auto cont = [](uint32_t v) { return v >= 0x80 && v <= 0xBF; };
  auto lead2 = [](uint32_t v) { return v >= 0xC2 && v <= 0xDF; };
  if (a < 0x80) {
    if (c < 0x80) return d < 0x80;
    return lead2(c) && cont(d);
  }
  if (lead2(a)) return cont(c) && d < 0x80;
  if (a == 0xE0) return c >= 0xA0 && c <= 0xBF && cont(d);
  if (a == 0xED) return c >= 0x80 && c <= 0x9F && cont(d);
  if (a >= 0xE1 && a <= 0xEF) return cont(c) && cont(d);
  return false;
}

// Six-frame translation in ONE read of the sequence.  Every base position p of a record is the first
// base of exactly one forward codon (frame p mod 3, residue p/3) and the last-read base of exactly one
// reverse-complement codon (frame (len-1-p) mod 3), so one lane per base position produces both
// from the five bytes p-2 .. p+2, served from an LDS tile that the workgroup loaded with coalesced
// 16-byte reads.  Segment order in the residue buffer: reference src/lib.rs:280-300.
constexpr int kTrThreads = 256;
// 12 consecutive bases per lane: every frame gets 4 consecutive residues from a lane (forward:
// ascending, reverse: descending), written as ONE unaligned 4-byte store; consecutive lanes write
// consecutive dwords of the same frame.  (One byte store per residue -- 25 G of them for 12.5 GB
// of DNA -- is what bounded the first version, not HBM.)
constexpr int kTrPerThread = 12;
constexpr int kTrTile = kTrThreads * kTrPerThread;  // bases per workgroup tile

__global__ __launch_bounds__(kTrThreads) void k_translate(SeqBatch b, const uint64_t* __restrict__ seg_off,
                                                          uint32_t nseg, uint32_t ksize, uint8_t* __restrict__ res,
                                                          uint32_t* __restrict__ bad_utf8) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[kTrTile + 64];   // raw bytes (only the UTF-8 check reads them)
  __shared__ __attribute__((aligned(16))) uint8_t code[kTrTile + 64];   // T0 C1 A2 G3 (complement = ^2), 0x80 = not a base
  __shared__ uint8_t lut_code[256];
  __shared__ uint8_t lut_aa[64];
  const int tid = threadIdx.x;
  {
    // byte -> base code, case-insensitive (the reference upper-cases first, src/lib.rs:253-256)
    const uint32_t c = (uint32_t)tid, u = upper(c);
    const int t = tcag(u);
    lut_code[tid] = t < 0 ? 0x80u : (uint8_t)t;
    if (tid < 64) lut_aa[tid] = (uint8_t)kCodonAA[tid];
  }
  const uint64_t ntiles = (b.len + kTrTile - 1) / kTrTile;
  const uintptr_t gend = ((uintptr_t)(b.seq + b.len) + 15) & ~(uintptr_t)15;
  // a workgroup owns a CONTIGUOUS run of tiles: the record holding the tile start is found by
  // binary search once and then walked forward (a search per lane per tile is a chain of ~14
  // dependent loads for 10^4 records -- it used to dominate this bandwidth-bound kernel)
  const uint64_t per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
  const uint64_t t_begin = (uint64_t)blockIdx.x * per_wg;
  const uint64_t t_end = t_begin + per_wg < ntiles ? t_begin + per_wg : ntiles;
  uint32_t rec0 = 0;
  if (b.starts && t_begin < t_end) rec0 = find_record(b.starts, b.nrec, t_begin * kTrTile);
  for (uint64_t tix = t_begin; tix < t_end; tix++) {
    const uint64_t T0 = tix * kTrTile;
    if (b.starts)
      while (rec0 + 1 < b.nrec && T0 >= b.starts[rec0 + 1]) rec0++;   // uniform
    // tile bytes [T0 - 16, T0 + kTrTile + 16) relative to an aligned base (a halo of 2 each side is needed)
    const uintptr_t g0 = (uintptr_t)(b.seq + T0);
    const uintptr_t ga = (g0 & ~(uintptr_t)15) - 16;
    const uint32_t m = (uint32_t)(g0 - ga);  // 16..31: offset of position T0 inside the tile
    __syncthreads();
    for (uint32_t c = tid; c < (kTrTile + 64) / 16; c += kTrThreads) {
      const uintptr_t addr = ga + ((uintptr_t)c << 4);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (addr >= ((uintptr_t)b.seq & ~(uintptr_t)15) && addr < gend) v = *reinterpret_cast<const uint4*>(addr);
      *reinterpret_cast<uint4*>(tile + (c << 4)) = v;
      uint32_t w[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
      for (int d = 0; d < 4; d++)
        o[d] = (uint32_t)lut_code[w[d] & 0xff] | ((uint32_t)lut_code[(w[d] >> 8) & 0xff] << 8) |
               ((uint32_t)lut_code[(w[d] >> 16) & 0xff] << 16) | ((uint32_t)lut_code[w[d] >> 24] << 24);
      *reinterpret_cast<uint4*>(code + (c << 4)) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    // record of the lane's first position; lanes then step forward (records are walked in order)
    uint32_t rec = 0;
    uint64_t rs = 0, re = b.len;
    const uint64_t pfirst = T0 + (uint64_t)tid * kTrPerThread;
    if (b.starts && pfirst < b.len) {
      rec = rec0;
      // many tiny records inside one tile: search instead of walking
      if (rec0 + 8 < b.nrec && pfirst >= b.starts[rec0 + 8]) rec = find_record(b.starts, b.nrec, pfirst);
      rs = b.starts[rec]; re = b.starts[rec + 1];
    }
    // fast path: the lane's 12 bases and their 2-base halos lie inside one record that is long enough
    bool packed = false;
    if (pfirst < b.len) {
      while (pfirst >= re) { rec++; rs = b.starts[rec]; re = b.starts[rec + 1]; }
      const uint64_t rl = re - rs, o = pfirst - rs;
      packed = rl >= ksize && o >= 2 && o + kTrPerThread + 2 <= rl;
      if (packed) {
        const uint32_t x0 = m + (uint32_t)(pfirst - T0);
        uint32_t k[kTrPerThread + 4];   // codes of bases pfirst-2 .. pfirst+13
        uint32_t anybad = 0;
#pragma unroll
        for (int i = 0; i < kTrPerThread + 4; i++) { k[i] = code[x0 - 2 + i]; anybad |= k[i]; }
        if (anybad & 0x80u) packed = false;   // a non-base byte: the per-base path sorts out dropped codons / UTF-8
        if (packed) {
          const uint32_t of = (uint32_t)(o % 3);
          const uint64_t back0 = rl - 1 - o;          // reverse index of the lane's first base
          const uint32_t bf = (uint32_t)(back0 % 3);
#pragma unroll
          for (int c = 0; c < 3; c++) {
            // forward codons starting at bases c, c+3, c+6, c+9 of the run: frame (o+c)%3, residues (o+c)/3 ..+3
            uint32_t v = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
              const int i = 2 + c + 3 * t;
              v |= (uint32_t)lut_aa[(k[i] << 4) | (k[i + 1] << 2) | k[i + 2]] << (8 * t);
            }
            uint32_t fr = of + c; fr = fr >= 3 ? fr - 3 : fr;
            __builtin_memcpy(res + seg_off[6 * rec + 2 * fr] + (o + c) / 3, &v, 4);
            // reverse codons whose first base is base c, c+3, c+6, c+9: reverse index back0-c-3t,
            // frame (back0-c)%3, residues (back0-c)/3 down to (back0-c)/3-3
            uint32_t w = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
              const int i = 2 + c + 3 * t;
              w |= (uint32_t)lut_aa[((k[i] << 4) | (k[i - 1] << 2) | k[i - 2]) ^ 0x2a] << (8 * (3 - t));
            }
            uint32_t br = bf + 3 - c; br = br >= 3 ? br - 3 : br;
            __builtin_memcpy(res + seg_off[6 * rec + 2 * br + 1] + (back0 - c) / 3 - 3, &w, 4);
          }
        }
      }
    }
    if (!packed) {
    for (int q = 0; q < kTrPerThread; q++) {
      const uint64_t p = pfirst + q;
      if (p >= b.len) break;
      while (p >= re) { rec++; rs = b.starts[rec]; re = b.starts[rec + 1]; }
      const uint64_t rl = re - rs, o = p - rs;
      if (rl < ksize) continue;  // reference src/lib.rs:257: shorter records add nothing
      const uint32_t x = m + (uint32_t)(p - T0);
      const uint32_t k0 = code[x];
      // forward codon starting at o
      if (o + 2 < rl) {
        const uint32_t k1 = code[x + 1], k2 = code[x + 2];
        const uint32_t seg = 6 * rec + 2 * (uint32_t)(o % 3);
        const uint32_t bad = (k0 | k1 | k2) & 0x80u;
        if (bad) {
          const uint32_t c0 = upper(tile[x]), c1 = upper(tile[x + 1]), c2 = upper(tile[x + 2]);
          if (((c0 | c1 | c2) & 0x80u) && !utf8_ok3(c0, c1, c2)) atomicOr(&bad_utf8[seg], 1u);
        }
        res[seg_off[seg] + o / 3] = bad ? (uint8_t)kDropped : lut_aa[(k0 << 4) | (k1 << 2) | k2];
      }
      // reverse-complement codon whose first base is the complement of position o
      if (o >= 2) {
        const uint32_t k1 = code[x - 1], k2 = code[x - 2];
        const uint64_t back = rl - 1 - o;
        const uint32_t seg = 6 * rec + 2 * (uint32_t)(back % 3) + 1;
        const uint32_t bad = (k0 | k1 | k2) & 0x80u;
        if (bad) {
          const uint32_t d0 = comp_upper(upper(tile[x])), d1 = comp_upper(upper(tile[x - 1])), d2 = comp_upper(upper(tile[x - 2]));
          if (((d0 | d1 | d2) & 0x80u) && !utf8_ok3(d0, d1, d2)) atomicOr(&bad_utf8[seg], 1u);
        }
        res[seg_off[seg] + back / 3] = bad ? (uint8_t)kDropped : lut_aa[(((k0 << 4) | (k1 << 2) | k2)) ^ 0x2a];
      }
    }
    }
  }
}

// ---------------------------------------------------------------------------------
// protein arm, fused: translation and hashing in ONE pass over the DNA, no residue buffer.
//
// A window of `W` residues of any of the six frames is the translation of 3W consecutive bases --
// read forward, or read backward and complemented.  Every start position `a` of a record with
// a + 3W <= length carries exactly one forward window (frame a mod 3) and exactly one
// reverse-complement window, so the arm is the DNA kernel's walk with k = 3W and TWO hashes per
// position instead of a canonical choice.  A lane keeps, for each of the three reading frames that
// pass through its run, the last W residues of the forward translation and of the reverse-complement
// translation (byte strings in registers).  Each base completes one codon: its six digit bits index a
// 64-entry LDS table that holds the codon's residue and the residue of its reverse complement; the
// forward string of that frame shifts down a byte and takes the new residue on top, the
// reverse-complement string shifts up and takes it at the bottom (reference src/lib.rs:280-300 reads
// revcomp(sequence) forward, which is the record backward), and both strings are hashed
// (MurmurHash3 x64_128 of W <= 16 bytes: no full block, k1 and k2 straight from the registers).
//
// That covers every window whose 3W bases are all ACGT inside one record.  Windows that SKIP dropped
// codons (to_aa drops a codon holding anything else and splices its neighbours together, quirk Q8)
// are rare: the lane notices that a span is not clean with the same group test as the DNA kernel
// and puts its start position on a list; k_spliced_windows, a second tiny launch, walks the bytes
// of those spans in global memory exactly like the reference.  (A routine called from inside the
// hashing kernel needs a stack, and a kernel with a stack costs ~110 us more PER LAUNCH on this
// stack -- more than hashing a 5 Mbp genome.)  A byte >= 0x80 anywhere in the batch (where
// str::from_utf8 could panic, src/lib.rs:787) or a full list raises `high_flag`: the host then
// discards this launch and takes the two-pass path (k_translate + k_hash_windows), which
// reproduces the panic semantics and has no limits.
//
// Candidate positions are residue indices of the six-frame layout (segment 6r + 2*frame + strand),
// the order the reference walks them; seg_off is that layout's segment table.

// The hashing kernel reports a candidate's position as (a << 1 | strand): the first base of the
// window's span in the batch, strand 1 = reverse complement -- no table look-up on its emit path (a
// dependent global load there, taken by one lane in five hundred, stalls the whole wave).  When the
// caller wants positions (order-dependent sketch modes, grouped batches) this kernel rewrites them
// as residue indices of the six-frame layout, the order the reference walks the windows in.
__global__ __launch_bounds__(256) void k_protein_positions(uint64_t* __restrict__ pos, const unsigned long long* __restrict__ count,
                                                           uint64_t capacity, const uint64_t* __restrict__ starts, uint32_t nrec,
                                                           uint64_t batch_len, const uint64_t* __restrict__ seg_off, uint32_t kb,
                                                           uint64_t pos_base) {
  const uint64_t n = *count < capacity ? *count : capacity;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t code = pos[i], a = code >> 1;
    uint32_t rec = 0;
    uint64_t rs = 0, re = batch_len;
    if (starts) { rec = find_record(starts, nrec, a); rs = starts[rec]; re = starts[rec + 1]; }
    const uint64_t a_rel = a - rs, len = re - rs;
    uint64_t g;
    if ((code & 1) == 0) {
      g = seg_off[6 * (uint64_t)rec + 2 * (a_rel % 3)] + a_rel / 3;
    } else {
      const uint64_t rcidx = len - a_rel - kb;              // index, in revcomp(record), of the window's first base
      g = seg_off[6 * (uint64_t)rec + 2 * (rcidx % 3) + 1] + rcidx / 3;
    }
    pos[i] = pos_base + g;
  }
}

__device__ __forceinline__ int dna_digit(uint32_t c) {      // A0 C1 G2 T3 (either case), -1 otherwise
  const uint32_t u = c & 0xDFu;
  return u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : -1;
}
__device__ __forceinline__ uint32_t aa_of_digits(int d0, int d1, int d2) {
  // kCodonAA is indexed in T C A G order
  const int t[4] = {2, 1, 3, 0};
  return (uint32_t)(uint8_t)kCodonAA[16 * t[d0] + 4 * t[d1] + t[d2]];
}

// The two windows that START (forward) / whose first residue lies (reverse complement) in the span
// [a, a + 3W) when that span is not all-ACGT-in-one-record: walk codon by codon, dropping codons
// that hold anything but ACGT, like to_aa + windows() of the reference (src/lib.rs:779-793, 289-300).
// (everything by value: a reference parameter of a non-inlined function would force the caller's
// kernel arguments into scratch memory)
// the windows that START at base `a` with dropped codons spliced out (one forward, one reverse complement), straight
// to the global sink
__device__ __forceinline__ void spliced_windows(const SeqBatch& b, const HashParams& hp, const CandSink& sink, uint32_t win, uint64_t a) {
  const uint32_t kb = 3 * win;
  uint32_t rec = 0;
  uint64_t rs = 0, re = b.len;
  if (b.starts) { rec = find_record(b.starts, b.nrec, a); rs = b.starts[rec]; re = b.starts[rec + 1]; }
  const uint64_t len = re - rs;
  if (len < hp.ksize || a < rs || a >= re) return;          // reference src/lib.rs:257
  // forward: first residue = codon [a, a+3) of frame a_rel % 3, then the following codons of that frame
  {
    const uint32_t f = (uint32_t)((a - rs) % 3);
    const uint64_t fend = rs + f + 3 * ((len - f) / 3);     // end of the frame's translated extent
    Mm3Stream st(hp.seed);
    uint32_t got = 0;
    for (uint64_t c = a; c + 3 <= fend && got < win; c += 3) {
      const int d0 = dna_digit(b.seq[c]), d1 = dna_digit(b.seq[c + 1]), d2 = dna_digit(b.seq[c + 2]);
      if (d0 < 0 || d1 < 0 || d2 < 0) { if (c == a) break; continue; }   // a window starts at a KEPT residue
      st.push(aa_of_digits(d0, d1, d2));
      got++;
    }
    if (got == win) {
      const uint64_t h = st.finish();
      if (h <= hp.thr) emit(sink, h, a << 1);
    }
  }
  // reverse complement: first residue = the codon read backward from base a + 3W - 1
  if (a + kb <= re) {
    const uint64_t e = a + kb - 1;
    Mm3Stream st(hp.seed);
    uint32_t got = 0;
    for (uint64_t c = e; c >= rs + 2 && got < win; c -= 3) {
      const int d0 = dna_digit(b.seq[c]), d1 = dna_digit(b.seq[c - 1]), d2 = dna_digit(b.seq[c - 2]);
      if (d0 < 0 || d1 < 0 || d2 < 0) { if (c == e) break; if (c < 3) break; continue; }
      st.push(aa_of_digits(3 - d0, 3 - d1, 3 - d2));
      got++;
      if (c < 3) break;                                     // (c -= 3 must not wrap)
    }
    if (got == win) {
      const uint64_t h = st.finish();
      if (h <= hp.thr) emit(sink, h, (a << 1) | 1u);
    }
  }
}

constexpr uint32_t kSplicedCap = 1u << 20;   // spans a launch may set aside (more: the two-pass path)
__global__ __launch_bounds__(256) void k_spliced_windows(SeqBatch b, HashParams hp, CandSink sink, uint32_t win,
                                                         const uint64_t* __restrict__ list, const uint32_t* __restrict__ count) {
  const uint32_t n = *count < kSplicedCap ? *count : kSplicedCap;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) spliced_windows(b, hp, sink, win, list[i]);
}

// murmur64 of a W-byte string, 1 <= W <= 16: no full block, k1 = bytes 0..7, k2 = bytes 8..W-1 (reference
// src/lib.rs:33-35 on aa.windows()).  k1c1 = k1 * c1 is handed in (the two strands get it differently, see
// k_protein_fused); k2 = the bytes past the eighth as they are; the digest is left open (w2_fmix_open).
// W == 9: k2 is one byte, so its whole contribution -- seed ^ mix_k2(byte) ^ W -- comes from a table (h2_ready).
template <int W>
__device__ __forceinline__ void murmur_short(W2 k1c1, W2 k2, W2 seedw /* seed ^ W, in vector registers */, uint64_t h2_ready, W2& a, W2& b) {
  W2 h1 = seedw, h2{(uint32_t)h2_ready, (uint32_t)(h2_ready >> 32)};
  if (W != 9) {
    h2 = seedw;
    if (W > 8) h2 = w2_xor(h2, w2_mul(w2_rotl(w2_mul(k2, kC2), 33), kC1));
  }
  h1 = w2_xor(h1, w2_mul(w2_rotl(k1c1, 31), kC2));
  w2_cross_add(h1, h2);
  a = w2_fmix_open(h1); b = w2_fmix_open(h2);
}

// byte BYTE of x, shifted left by SH, in one sub-dword-addressed instruction (full rate; see k_dna_rolling)
template <int SH, int BYTE>
__device__ __forceinline__ uint32_t byte_shl(uint32_t x) {
  static_assert(SH == 3 && BYTE >= 0 && BYTE < 4, "byte_shl");
  uint32_t r;
#define SMH_BSHL(SH_, B_) asm("v_lshlrev_b32_sdwa %0, " #SH_ ", %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #B_ : "=v"(r) : "v"(x))
  if (BYTE == 0) SMH_BSHL(3, 0); else if (BYTE == 1) SMH_BSHL(3, 1); else if (BYTE == 2) SMH_BSHL(3, 2); else SMH_BSHL(3, 3);
#undef SMH_BSHL
  return r;
}

#ifndef SMH_PF_MINW
#define SMH_PF_MINW 4
#endif
#ifndef SMH_PF_LOGR
#define SMH_PF_LOGR 7
#endif
template <int W, int THREADS>
__global__ __launch_bounds__(THREADS, SMH_PF_MINW) void k_protein_fused(SeqBatch b, HashParams hp, CandSink sink, int logR,
                                                                          uint32_t stage_cap, uint32_t* __restrict__ high_flag,
                                                                          uint64_t* __restrict__ slow_list, uint32_t* __restrict__ slow_count) {
  constexpr int KB = 3 * W;                          // bases per window
  constexpr int ND = (W + 3) / 4;                    // dwords of a residue string
  // static LDS: codon tables -- codon number (its first base's digit lowest) -> (residue of the reverse complement) * c1 |
  // W == 9: seed ^ mix_k2(residue) ^ W | residue + residue of the reverse complement << 8 -- as three arrays of 8-byte
  // entries read with ONE offset register (32-byte records would put every look-up of a wave on eight banks), and
  // seed ^ mix_k2(byte) ^ W of any byte (W == 9: the reverse strand's k2 is a residue met eight codons earlier)
  __shared__ uint64_t ctab[3 * 64];
  __shared__ uint64_t k2tab[W == 9 ? 256 : 1];
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* st_ctl = smem;
  uint64_t* st_hash = reinterpret_cast<uint64_t*>(st_ctl + 4);
  uint64_t* st_pos = st_hash + stage_cap;
  uint32_t* tile = reinterpret_cast<uint32_t*>(st_pos + (sink.pos ? stage_cap : 0));
  const Stage stage{st_ctl, st_hash, st_pos, stage_cap};

  const int tid0 = threadIdx.x;
  const uint32_t R = 1u << logR;
  const uint64_t TILE = (uint64_t)THREADS << logR;
  const bool multi = b.starts != nullptr;
  if (tid0 < 64) {
    const int d0 = tid0 & 3, d1 = (tid0 >> 2) & 3, d2 = (tid0 >> 4) & 3;
    const uint32_t af = aa_of_digits(d0, d1, d2), ar = aa_of_digits(3 - d2, 3 - d1, 3 - d0);
    const uint64_t arc1 = (uint64_t)ar * kC1, h2f = hp.seed ^ mix_k2((uint64_t)af) ^ (uint64_t)W;
    ctab[tid0] = arc1; ctab[64 + tid0] = h2f; ctab[128 + tid0] = af | (ar << 8);
  }
  if (W == 9) for (int e = tid0; e < 256; e += THREADS) k2tab[e] = hp.seed ^ mix_k2((uint64_t)e) ^ (uint64_t)W;
  if (tid0 == 0) st_ctl[0] = 0;

  const uint32_t thr_hi1 = open_thr(hp.thr);
  W2 seedw{(uint32_t)hp.seed ^ (uint32_t)W, (uint32_t)(hp.seed >> 32)};   // in vector registers: see k_dna_rolling
  asm volatile("" : "+v"(seedw.lo), "+v"(seedw.hi));
  const uint64_t ntiles = (b.len + TILE - 1) / TILE;
  const uintptr_t gend = ((uintptr_t)(b.seq + b.len) + 15) & ~(uintptr_t)15;
  const uint32_t nsteps = ((R + (uint32_t)KB - 1 + 11) / 12) * 12;   // bases walked per lane: whole triples of dwords
  const uint32_t warm_end = ((uint32_t)(KB - 1) / 12) * 12;          // iterations [0, warm_end) end before any window is complete
  // the end of the valid part of record r: a record shorter than ksize adds nothing (src/lib.rs:257)
  auto valid_end = [&](uint32_t r) -> uint64_t {
    const uint64_t s0 = b.starts[r], s1 = b.starts[r + 1];
    return s1 - s0 >= hp.ksize ? s1 : s0;
  };

  for (uint64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    // The lane number, re-read per tile behind an opaque asm: the compiler otherwise computes tid << logR and friends
    // once, ahead of this loop, and -- the main loop holding every register -- spills them; their reload at every tile
    // came from memory (the streamed input evicts the scratch lines from the L2 in between): 11 % more bytes fetched
    // than the input holds, 26 % for W = 10 (TCC_EA0_RDREQ, profiles/r02_ea_read_requests.txt).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const uint64_t T0 = tix * TILE;
    const uintptr_t g0 = (uintptr_t)(b.seq + T0);
    const uintptr_t ga = g0 & ~(uintptr_t)15;
    const uint32_t m = (uint32_t)(g0 - ga);
    const uint32_t nchunks = (m + (uint32_t)TILE + (uint32_t)KB + 12 + 8 + 15) >> 4;
    __syncthreads();
    uint32_t high = 0;
    for (uint32_t c = tid; c < nchunks; c += THREADS) {
      uintptr_t addr = ga + ((uintptr_t)c << 4);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (addr < gend) v = *reinterpret_cast<const uint4*>(addr);
      high |= v.x | v.y | v.z | v.w;
      uint32_t x = c << 4;
      uint32_t o = (x >> 2) + (x >> logR);
      tile[o] = v.x; tile[o + 1] = v.y; tile[o + 2] = v.z; tile[o + 3] = v.w;
    }
    if (high & 0x80808080u) atomicOr(high_flag, 1u);      // (bytes of the 16-byte granules around the batch may flag too: harmless)
    __syncthreads();

    const uint64_t p0 = T0 + ((uint64_t)tid << logR);     // my first window start position
    if (p0 < b.len) {
    const uint32_t nk = (b.len - p0) < R ? (uint32_t)(b.len - p0) : R;
    const uint32_t hi_ok = nk + (uint32_t)KB - 1;

    uint32_t rec = 0;
    uint64_t cur_end = b.vend0;
    if (multi) rec = find_record_in_tile(b, hp.ksize, tix, p0, &cur_end);
    uint32_t lim = 0;
    if (cur_end > p0) lim = (cur_end - p0) > 0xfffffffeull ? 0xffffffffu : (uint32_t)(cur_end - p0);
    uint32_t vstart = 0;                                  // first base index after the last record start / dropped base
    uint32_t rstart = 0;                                  // first base index of the record the walk is in
    uint32_t g_lo = 0, g_span = 0;
    auto set_clean_window = [&](bool warm) {
      const uint32_t lim2 = warm ? lim : (lim < hi_ok ? lim : hi_ok);
      g_lo = warm ? vstart : vstart + (uint32_t)KB - 1;
      g_span = lim2 >= g_lo + 4 ? lim2 - 3 - g_lo : 0u;
    };

    const uint32_t xu = m & ~3u;
    uint32_t ta = xu + ((uint32_t)tid << logR);
    ta = ta + ((ta >> logR) << 2);
    const uint32_t sh = m & 3u;
    uint32_t cur = tile[ta >> 2];
    ta += ((xu + 4u) & (R - 1)) == 0 ? 8u : 4u;
    uint32_t nxt = tile[ta >> 2];                         // always one dword ahead: its latency hides behind the hashing
    uint32_t pcode = 0;                                   // digits of the previous group, one per byte
    // per reading frame (base index mod 3): the forward and the reverse-complement residue strings (for W == 9 the
    // top dword of Sf is the last table entry as it came -- only its low byte is ever used -- and Sr keeps k1's 8 bytes)
    uint32_t Sf[3][ND], Sr[3][ND];
    uint64_t Mr[3];                                       // W >= 8: (first 8 residues of the reverse-complement string) * c1
#pragma unroll
    for (int t = 0; t < 3; t++) {
      Mr[t] = 0;
#pragma unroll
      for (int d = 0; d < ND; d++) { Sf[t][d] = 0; Sr[t][d] = 0; }
    }

    // four bases (one dword of the tile); PH = (base index / 4) mod 3 fixes which frame each base completes
    auto group = [&](uint32_t i0, auto phase, auto hashing) {
      constexpr int PH = decltype(phase)::value;
      constexpr bool kHash = decltype(hashing)::value;
      // W == 9, reverse strand: k2 is the residue that leaves k1 -- byte 7 of the string as it stands before the base
      // (for the group's fourth base, which shares its frame with the first: byte 6 of the string before the group).
      // Nothing of this group is needed for it, so the four look-ups go first and overlap with everything below.
      uint64_t h2r[4] = {0, 0, 0, 0};
      if (kHash && W == 9) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t s1 = Sr[(4 * PH + q) % 3][1];
          h2r[q] = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(k2tab) + (q < 3 ? byte_shl<3, 3>(s1) : byte_shl<3, 2>(s1)));
        }
      }
      const uint32_t d = __builtin_amdgcn_alignbyte(nxt, cur, sh);
      cur = nxt;
      const uint32_t xn = xu + i0 + 8;
      ta += (xn & (R - 1)) == 0 ? 8u : 4u;
      nxt = tile[ta >> 2];                                 // for the next group
      const uint32_t u4 = d & 0xDFDFDFDFu;
      const uint32_t c2 = (u4 >> 1) & 0x03030303u;
      const uint32_t code4 = c2 ^ ((c2 >> 1) & 0x01010101u);
      const uint32_t exp4 = __builtin_amdgcn_perm(0u, 0x54474341u, code4);
      const uint32_t diff4 = u4 ^ exp4;
      uint32_t okmask = 0xFu, slowmask = 0;
      if (!(diff4 == 0 && i0 - g_lo < g_span)) {
        okmask = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t i = i0 + q;
          uint32_t bad = (diff4 >> (8 * q)) & 0xffu;
          if (i >= lim) {
            const uint64_t qpos = p0 + i;
            if (multi) {
              while (rec + 1 < b.nrec && qpos >= b.starts[rec + 1]) { rec++; vstart = i; rstart = i; }
              cur_end = valid_end(rec);
            }
            if (qpos >= cur_end) { bad = 1; lim = i + 1; }
            else lim = (cur_end - p0) > 0xfffffffeull ? 0xffffffffu : (uint32_t)(cur_end - p0);
          }
          if (bad) vstart = i + 1;
          if (kHash) {
            const bool inrun = i + 1 >= (uint32_t)KB && i < hi_ok;
            const bool simple = inrun && (i + 1 >= vstart + (uint32_t)KB);
            okmask |= simple ? (1u << q) : 0u;
            // A span that is not clean goes to the codon-by-codon walk -- unless it STARTS in an earlier record than
            // this base: its record then ends before 3W bases are through, and no window starts there.  (Every record
            // boundary makes 3W - 1 such spans; sending them through the walk cost a tenth of the kernel's time at
            // one boundary per 1 MB.)
            slowmask |= (inrun && !simple && i + 1 >= rstart + (uint32_t)KB) ? (1u << q) : 0u;
          }
        }
        set_clean_window(!kHash);
      }
      // Codon numbers of the codons that END at the group's four bases.  e4 = digits of bases i0-2 .. i0+1, one per
      // byte; x | x >> 6 | x >> 12 gathers three consecutive bytes' digits into the first one's byte (oldest lowest).
      const uint32_t e4 = __builtin_amdgcn_alignbyte(code4, pcode, 2);
      pcode = code4;
      const uint32_t ix01 = e4 | (e4 >> 6) | (e4 >> 12);              // bytes 0, 1: bases 0, 1
      const uint32_t ix23 = code4 | (code4 >> 6) | (code4 >> 12);     // bytes 0, 1: bases 2, 3
      // the four look-ups are issued together, ahead of the hashing they feed
      uint32_t eaa[4];
      uint64_t et1[4], h2f[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t off = (q & 1) ? byte_shl<3, 1>(q < 2 ? ix01 : ix23) : byte_shl<3, 0>(q < 2 ? ix01 : ix23);
        const char* at = reinterpret_cast<const char*>(ctab) + off;
        if (W >= 8) et1[q] = *reinterpret_cast<const uint64_t*>(at);
        if (kHash && W == 9) h2f[q] = *reinterpret_cast<const uint64_t*>(at + 512);
        eaa[q] = *reinterpret_cast<const uint32_t*>(at + 1024);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int t = (4 * PH + q) % 3;                    // reading frame completed by this base (lane-relative)
        const uint32_t e = eaa[q];                         // residue | residue of the reverse complement << 8
        // forward string: drop the oldest residue (byte 0), the new one becomes byte W-1
#pragma unroll
        for (int dd = 0; dd < ND - 1; dd++) Sf[t][dd] = __builtin_amdgcn_alignbyte(Sf[t][dd + 1], Sf[t][dd], 1);
        if (W == 9) Sf[t][ND - 1] = e;
        else Sf[t][ND - 1] = __builtin_amdgcn_perm(e, Sf[t][ND - 1], ((W - 1) & 3) == 0 ? 0x0c0c0c04u : ((W - 1) & 3) == 1 ? 0x0c0c0401u : ((W - 1) & 3) == 2 ? 0x0c040201u : 0x04030201u);
        // reverse-complement string: the new residue is its FIRST (byte 0), the oldest (byte W-1) falls off.
        // W >= 8: k1 * c1 rolls -- k1' = k1 << 8 | new (mod 2^64), and a left shift commutes with the multiplication:
        // (k1 * c1)' = (k1 * c1) << 8 + new * c1 (table), two instructions instead of a 64 x 64 multiply.
#pragma unroll
        for (int dd = ND - 1; dd > 0; dd--) {
          if (W == 9 && dd == 2) continue;
          const bool last = dd == ND - 1 && (W & 3) != 0;   // partial top dword: keep the bytes past W zero
          Sr[t][dd] = !last ? __builtin_amdgcn_alignbyte(Sr[t][dd], Sr[t][dd - 1], 3)
                            : __builtin_amdgcn_perm(Sr[t][dd], Sr[t][dd - 1], (W & 3) == 1 ? 0x0c0c0c03u : (W & 3) == 2 ? 0x0c0c0403u : 0x0c050403u);
        }
        Sr[t][0] = __builtin_amdgcn_perm(e, Sr[t][0], 0x02010005u);
        if (W >= 8) {
          uint64_t m16;
          asm("v_lshl_add_u64 %0, %1, 4, 0" : "=v"(m16) : "v"(Mr[t]));
          asm("v_lshl_add_u64 %0, %1, 4, %2" : "=v"(Mr[t]) : "v"(m16), "v"(et1[q]));
        }
        if (kHash && i0 + q + 1 >= (uint32_t)KB && i0 + q < R + (uint32_t)KB - 1) {   // uniform: a window of some lane's run can be complete here
          W2 fa, fb, ra, rb;
          const W2 k2f{ND > 2 ? Sf[t][ND > 2 ? 2 : 0] : 0u, ND > 3 ? Sf[t][ND > 3 ? 3 : 0] : 0u};
          const W2 k2r{ND > 2 ? Sr[t][ND > 2 ? 2 : 0] : 0u, ND > 3 ? Sr[t][ND > 3 ? 3 : 0] : 0u};
          murmur_short<W>(w2_mul(W2{Sf[t][0], Sf[t][1]}, kC1), k2f, seedw, h2f[q], fa, fb);
          murmur_short<W>(W >= 8 ? w2_split(Mr[t]) : w2_mul(W2{Sr[t][0], Sr[t][1]}, kC1), k2r, seedw, h2r[q], ra, rb);
          if (open_may_pass(fa, fb, thr_hi1) || open_may_pass(ra, rb, thr_hi1)) {   // ~2 in `scaled` positions get here
            uint32_t om = okmask;
            asm volatile("" : "+v"(om));
            const uint64_t hf = open_finish(fa, fb), hr = open_finish(ra, rb);
            if ((om >> q) & 1u) {
              const uint64_t a = p0 + i0 + q + 1 - KB;      // first base of the span; see k_protein_positions
              if (hf <= hp.thr) stage_emit(stage, sink, hf, a << 1);
              if (hr <= hp.thr) stage_emit(stage, sink, hr, (a << 1) | 1u);
            }
          }
        }
      }
      if (kHash && slowmask) {
#pragma unroll 1
        for (int q = 0; q < 4; q++)
          if ((slowmask >> q) & 1u) {                     // for k_spliced_windows
            const uint32_t at = atomicAdd(slow_count, 1u);
            if (at < kSplicedCap) slow_list[at] = p0 + i0 + q + 1 - KB;
            else atomicOr(high_flag, 2u);
          }
      }
    };

    set_clean_window(true);
    uint32_t i0 = 0;
    for (; i0 < warm_end; i0 += 12) {
      group(i0, std::integral_constant<int, 0>{}, std::false_type{});
      group(i0 + 4, std::integral_constant<int, 1>{}, std::false_type{});
      group(i0 + 8, std::integral_constant<int, 2>{}, std::false_type{});
    }
    set_clean_window(false);
    for (; i0 < nsteps; i0 += 12) {
      group(i0, std::integral_constant<int, 0>{}, std::true_type{});
      group(i0 + 4, std::integral_constant<int, 1>{}, std::true_type{});
      group(i0 + 8, std::integral_constant<int, 2>{}, std::true_type{});
    }
    }  // p0 < b.len

    stage_flush(stage, sink, tid, THREADS);
  }
}

// protein arm, phase 2: a window starts at every kept residue and takes the next `win` kept
// residues of the same segment (dropped codons are spliced out, quirk Q8).
__device__ __forceinline__ void hash_window_slow(const uint8_t* __restrict__ res, uint64_t g, uint64_t end, uint32_t win,
                                                 const HashParams& hp, uint64_t thr, const CandSink& sink,
                                                 const Stage& stage) {
  if (res[g] == kDropped) return;
  Mm3Stream st(hp.seed);
  uint32_t got = 0;
  for (uint64_t q = g; q < end && got < win; q++) {
    uint32_t c = res[q];
    if (c != kDropped) { st.push(c); got++; }
  }
  if (got < win) return;
  uint64_t h = st.finish();
  if (h <= thr) stage_emit(stage, sink, h, hp.pos_base + g);
}

constexpr int kWinRun = 8;
// W: compile-time window length (0 = run-time `win_rt`): with W known the loads, the byte shifts and
// murmur's block / tail structure are all static
template <int W>
__device__ __forceinline__ void hash_run(const uint8_t* __restrict__ res, const uint64_t* __restrict__ seg_off,
                                         uint32_t nseg, uint32_t win_rt, const HashParams& hp, uint64_t thr,
                                         const CandSink& sink, const Stage& stage, bool aligned, uint64_t g0,
                                         uint64_t blk_seg_end, const uint64_t* k2lut) {
  {
    const uint32_t win = W ? (uint32_t)W : win_rt;
    // the workgroup's first segment was looked up once with uniform (scalar) loads; only lanes
    // past its end search for their own
    uint64_t end = blk_seg_end;
    if (g0 >= end) end = seg_off[find_record(seg_off, nseg, g0) + 1];
    const uint64_t last = g0 + kWinRun < hp.range_hi ? g0 + kWinRun : hp.range_hi;  // window starts [g0, last)
    bool fast = aligned && last == g0 + kWinRun && g0 + kWinRun - 1 + win <= end;
    uint32_t D[12];  // 48 bytes: 8 starts + up to 32-byte windows, little-endian dwords
    if (fast) {
      const uint2* src = reinterpret_cast<const uint2*>(res + g0);
      const int nw = (kWinRun + (int)win - 1 + 7) >> 3;  // 8-byte words covering the stretch
      uint32_t anydrop = 0;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        uint2 v = make_uint2(0, 0);
        if (i < nw) v = src[i];
        D[2 * i] = v.x; D[2 * i + 1] = v.y;
        if (i < nw) {
          // a byte equal to 0xFF anywhere in the loaded words (bytes past the stretch may flag too: harmless)
          uint32_t nx = ~v.x, ny = ~v.y;
          anydrop |= ((nx - 0x01010101u) & ~nx & 0x80808080u) | ((ny - 0x01010101u) & ~ny & 0x80808080u);
        }
      }
      if (anydrop) fast = false;
    }
    if (!fast) {
      for (uint64_t g = g0; g < last; g++) {
        uint64_t e = end;
        if (g >= end) e = seg_off[find_record(seg_off, nseg, g) + 1];
        hash_window_slow(res, g, e, win, hp, thr, sink, stage);
      }
      return;
    }
    const int nblocks = (int)win >> 4, tail = (int)win & 15;
#pragma unroll
    for (int j = 0; j < kWinRun; j++) {
      // window j = bytes [j, j+win) of the stretch
      uint32_t Wd[8];
#pragma unroll
      for (int d = 0; d < 8; d++) {
        const int lo = d + (j >> 2);
        Wd[d] = (4 * d < (int)win) ? __builtin_amdgcn_alignbyte(D[lo + 1 < 12 ? lo + 1 : 11], D[lo], j & 3) : 0u;
        const int nb = (int)win - 4 * d;
        if (nb > 0 && nb < 4) Wd[d] &= (1u << (8 * nb)) - 1u;
      }
      uint64_t h1 = hp.seed, h2 = hp.seed;
#pragma unroll
      for (int blk = 0; blk < 2; blk++) {
        const uint64_t k1 = Wd[4 * blk] | ((uint64_t)Wd[4 * blk + 1] << 32);
        const uint64_t k2 = Wd[4 * blk + 2] | ((uint64_t)Wd[4 * blk + 3] << 32);
        if (blk < nblocks) mm3_block(h1, h2, k1, k2);
        else if (blk == nblocks) {
          // nine residues: k2 is a single byte, its mix comes from a 256-entry table (2 of the 8
          // 64-bit multiplies of the hash)
          if (W == 9) h2 ^= k2lut[k2 & 0xffu];
          else if (tail > 8) h2 ^= mix_k2(k2);
          if (tail > 0) h1 ^= mix_k1(k1);
        }
      }
      const uint64_t h = mm3_finish(h1, h2, (uint64_t)win);
      if (h <= thr) stage_emit(stage, sink, h, hp.pos_base + g0 + j);
    }
  }
}

// One lane owns 8 consecutive window starts and reads the 8+win-1 residues they cover once
// (aligned 8-byte loads).  When that stretch lies inside one segment and holds no dropped codon --
// the normal case -- every window is a byte-shifted view of those registers; otherwise the lane
// falls back to the residue-by-residue walk.  win <= 32 for the fast path.
template <int W>
__global__ __launch_bounds__(256) void k_hash_windows(const uint8_t* __restrict__ res,
                                                      const uint64_t* __restrict__ seg_off,
                                                      uint32_t nseg, uint32_t win, HashParams hp,
                                                      CandSink sink, uint32_t stage_cap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t wsm[];
  __shared__ uint64_t k2lut[W == 9 ? 256 : 1];
  const Stage stage{wsm, reinterpret_cast<uint64_t*>(wsm + 4), reinterpret_cast<uint64_t*>(wsm + 4) + stage_cap, stage_cap};
  if (threadIdx.x == 0) wsm[0] = 0;
  if (W == 9) k2lut[threadIdx.x] = mix_k2((uint64_t)threadIdx.x);   // 256 threads, 256 byte values
  __syncthreads();
  const uint64_t thr = hp.thr;
  const bool aligned = ((hp.range_lo | (uintptr_t)res) & 7) == 0 && win <= 32 && win >= 1;
  // a workgroup owns a contiguous run of passes (2048 window starts each): the segment of the
  // pass start is searched once and then walked forward with uniform loads
  const uint64_t pass = (uint64_t)blockDim.x * kWinRun;
  const uint64_t npass = (hp.range_hi - hp.range_lo + pass - 1) / pass;
  const uint64_t per_wg = (npass + gridDim.x - 1) / gridDim.x;
  const uint64_t p_begin = (uint64_t)blockIdx.x * per_wg;
  const uint64_t p_end = p_begin + per_wg < npass ? p_begin + per_wg : npass;
  uint32_t seg = 0;
  if (p_begin < p_end) seg = find_record(seg_off, nseg, hp.range_lo + p_begin * pass);
  // the loop bound is the same for every lane of the workgroup: the flush inside synchronises
  for (uint64_t ps = p_begin; ps < p_end; ps++) {
    const uint64_t b0 = hp.range_lo + ps * pass;
    const uint64_t g0 = b0 + (uint64_t)threadIdx.x * kWinRun;
    while (seg + 1 < nseg && b0 >= seg_off[seg + 1]) seg++;   // b0 is uniform
    const uint64_t blk_seg_end = seg_off[seg + 1];
    if (g0 < hp.range_hi) hash_run<W>(res, seg_off, nseg, win, hp, thr, sink, stage, aligned, g0, blk_seg_end, k2lut);
    // a pass covers only 2048 windows: flushing (a returning global atomic) every pass would cost
    // more than the hashing; wait until the stage is half full
    stage_flush(stage, sink, threadIdx.x, blockDim.x, stage_cap / 2);
  }
  stage_flush(stage, sink, threadIdx.x, blockDim.x);
}

__global__ __launch_bounds__(256) void k_hash_segments(const uint8_t* __restrict__ bytes,
                                                       const uint64_t* __restrict__ off, uint32_t nseg,
                                                       uint64_t seed, uint64_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  Mm3Stream st(seed);
  for (uint64_t q = off[i]; q < off[i + 1]; q++) st.push(bytes[q]);
  out[i] = st.finish();
}

// ---------------------------------------------------------------------------------
// add_many (reference src/lib.rs:412-417) in bulk: the hashes already exist, only the filter and
// the append remain; the stream position of hash i is i.
__global__ __launch_bounds__(256) void k_filter_hashes(const uint64_t* __restrict__ hashes, HashParams hp, CandSink sink) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t thr = hp.thr;
  for (uint64_t i = hp.range_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hp.range_hi; i += stride) {
    const uint64_t h = hashes[i];
    if (h <= thr) emit(sink, h, hp.pos_base + i);
  }
}

// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t seed, uint64_t index) {
  uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// one lane per 64-bit generator word = 32 bases; `start` is a multiple of 32
__global__ __launch_bounds__(256) void k_synth_dna(uint8_t* __restrict__ out, uint64_t start, uint64_t len,
                                                   uint64_t seed, uint64_t n_every) {
  const uint64_t nwords = (len + 31) / 32;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    uint64_t bits = splitmix64(seed, (start >> 5) + w);
    uint64_t p = start + 32 * w;
    uint64_t nrem = n_every ? p % n_every : 0;
    uint64_t o = 32 * w;
    for (int j = 0; j < 32 && o + j < len; j++) {
      uint32_t c = (0x54474341u >> (8 * ((bits >> (2 * j)) & 3))) & 0xffu;
      if (n_every) {
        if (nrem == n_every - 1) c = 'N';
        nrem = nrem + 1 == n_every ? 0 : nrem + 1;
      }
      out[o + j] = (uint8_t)c;
    }
  }
}

inline int grid_for(uint64_t items, int per_block, int cap) {
  uint64_t g = (items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > (uint64_t)cap) g = cap;
  return (int)g;
}

}  // namespace

// ---------------------------------------------------------------------------------
// launchers

// launch geometry of the rolling kernel: 512 lanes, runs of 128 positions, two hashes per block.
// Other geometries exist only in experiment builds (-DSMH_EXPERIMENTS, SOURMASH_AMD_DNA_CFG="threads,logR,hb").
struct DnaCfg { int threads, logR, hb; bool packed = true; };
static DnaCfg dna_cfg() {
  static DnaCfg cfg = [] {
    DnaCfg c{512, 7, 2};
#ifdef SMH_EXPERIMENTS
    if (const char* e = std::getenv("SOURMASH_AMD_DNA_PK")) c.packed = std::atoi(e) != 0;     // 0: the byte tile (A/B)
#endif
#ifdef SMH_EXPERIMENTS
    if (const char* e = std::getenv("SOURMASH_AMD_DNA_CFG")) {
      int t = 0, r = 0, h = 0;
      if (sscanf(e, "%d,%d,%d", &t, &r, &h) == 3 && (t == 256 || t == 512) && r >= 5 && r <= 7 &&
          (h == 1 || h == 2 || h == 4)) c = DnaCfg{t, r, h, false};
    }
#endif
    return c;
  }();
  return cfg;
}

template <int KT, int L>
static void launch_rolling(const SeqBatch& b, const HashParams& p, const CandSink& sink, int grid, size_t lds,
                           int logR, uint32_t stage_cap, const DnaCfg& c, hipStream_t s) {
#define SMH_LAUNCH(T, H) hipLaunchKernelGGL((k_dna_rolling<KT, T, H, L>), dim3(grid), dim3(T), lds, s, b, p, sink, logR, stage_cap)
  if (p.thr_rec) {   // per-record thresholds: default geometry only
    if (c.packed) hipLaunchKernelGGL((k_dna_rolling<KT, 512, 2, L, true, true>), dim3(grid), dim3(512), lds, s, b, p, sink, logR, stage_cap);
    else hipLaunchKernelGGL((k_dna_rolling<KT, 512, 2, L, true>), dim3(grid), dim3(512), lds, s, b, p, sink, logR, stage_cap);
    return;
  }
  if (c.packed) {    // the product geometry: 512 lanes, two hashes per block, the tile packed to two bits per base
#ifdef SMH_EXPERIMENTS
    if (const char* e = std::getenv("SOURMASH_AMD_DNA_PKV")) {     // "minw,hb" of the packed kernel (A/B)
      int mw = 0, hb = 0;
      if (sscanf(e, "%d,%d", &mw, &hb) == 2) {
#define SMH_PKV(M_, H_) if (mw == M_ && hb == H_) { hipLaunchKernelGGL((k_dna_rolling<KT, 512, H_, L, false, true, M_>), dim3(grid), dim3(512), lds, s, b, p, sink, logR, stage_cap); return; }
        SMH_PKV(8, 1) SMH_PKV(7, 1) SMH_PKV(6, 1) SMH_PKV(5, 1) SMH_PKV(4, 1) SMH_PKV(8, 2) SMH_PKV(6, 2) SMH_PKV(5, 2) SMH_PKV(4, 2) SMH_PKV(8, 4) SMH_PKV(6, 4)
#undef SMH_PKV
      }
    }
#endif
    // (six waves per SIMD, one hash per block: 80 vector registers keep the scalar values out of vector lanes; 27.2 against
    // 27.6 ms per 10 GB with eight waves and two hashes per block, profiles/r04_pmc_dna_rolling.json "variants_measured")
    hipLaunchKernelGGL((k_dna_rolling<KT, 512, 1, L, false, true, 6>), dim3(grid), dim3(512), lds, s, b, p, sink, logR, stage_cap);
    return;
  }
#ifdef SMH_EXPERIMENTS
  if (c.threads == 512) { if (c.hb == 4) SMH_LAUNCH(512, 4); else if (c.hb == 2) SMH_LAUNCH(512, 2); else SMH_LAUNCH(512, 1); }
  else { if (c.hb == 4) SMH_LAUNCH(256, 4); else if (c.hb == 2) SMH_LAUNCH(256, 2); else SMH_LAUNCH(256, 1); }
#else
  SMH_LAUNCH(512, 2);
#endif
#undef SMH_LAUNCH
}

// the batch with its per-tile record table (k_tile_records) for a launch of `ntiles` tiles from position `base`
static SeqBatch with_tile_records(const SeqBatch& b, uint32_t min_len, uint64_t base, uint64_t tile, uint64_t ntiles, Device& dev,
                                  hipStream_t s) {
  SeqBatch r = b;
  if (!b.starts || b.nrec < 2) return r;
  dev.tile_rec.ensure((size_t)ntiles * sizeof(TileRec));
  hipLaunchKernelGGL(k_tile_records, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, s, b.starts, b.vends, b.nrec, min_len, base,
                     tile, ntiles, dev.tile_rec.as<TileRec>());
  r.tile_rec = dev.tile_rec.as<TileRec>();
  return r;
}

void launch_dna_hash(const SeqBatch& b_in, const HashParams& p, const CandSink& sink, Device& dev,
                     hipStream_t s, bool force_generic) {
  if (p.range_hi <= p.range_lo) return;
  const uint64_t span = p.range_hi - p.range_lo;
  if (p.ksize >= 1 && p.ksize <= 128 && !force_generic) {
    // run length per lane: long runs amortise the k-1 warm-up bases; short inputs use short
    // runs so that the launch still covers the chip
    DnaCfg c = dna_cfg();
    if (p.thr_rec) { c.threads = 512; c.logR = 7; c.hb = 2; }     // the per-record variant exists in one geometry
    int logR = c.logR;
    while (logR > 5 && (span >> logR) < (uint64_t)dev.cu_count() * c.threads * 2) logR--;
    const uint64_t tile = (uint64_t)c.threads << logR;
    const uint64_t ntiles = (span + tile - 1) / tile;
    const SeqBatch b = with_tile_records(b_in, 0, p.range_lo, tile, ntiles, dev, s);
    dev.prof_begin(s);
    int grid = (int)(ntiles < (uint64_t)dev.cu_count() * 8 ? ntiles : (uint64_t)dev.cu_count() * 8);
    const int limbs = p.ksize <= 32 ? 2 : (p.ksize <= 64 ? 4 : 8);
    const uint32_t x_bytes = (uint32_t)tile + 16 * limbs + 96;
    // LDS stage for the survivors of one tile: twice the expectation under a uniform hash, within
    // [128, 2048] entries; anything beyond goes straight to the global sink
    const uint64_t thr = p.thr;
    long double expect = (long double)tile * (((long double)thr + 1.0L) / 18446744073709551616.0L);
    uint32_t stage_cap = expect * 2.0L + 64.0L > 2048.0L ? 2048u : (uint32_t)(expect * 2.0L + 64.0L);
    if (stage_cap < 128) stage_cap = 128;
    size_t lds = 16 + (size_t)stage_cap * 8 * (sink.pos ? 2 : 1) + x_bytes +   // dynamic part; the tables are static LDS
                 4 * ((x_bytes >> logR) + 2);
    if (c.packed) {
      // two bits per base + a pad dword per run + a dirty bit per source dword (the kernel's nchunks_cap)
      const uint32_t ncap = (15u + (uint32_t)tile + 16u * (uint32_t)limbs + 8u + 15u + 32u) >> 4;
      lds = 16 + (size_t)stage_cap * 8 * (sink.pos ? 2 : 1) + 4 * (size_t)(ncap + (ncap >> (logR - 4)) + 2) + 4 * (size_t)((ncap >> 3) + 8);
    }
    // 8 workgroups per CU may be resident with the packed tile: keep a few tiles per workgroup
    if (p.ksize == 31) launch_rolling<31, 2>(b, p, sink, grid, lds, logR, stage_cap, c, s);
    else if (p.ksize == 21) launch_rolling<21, 2>(b, p, sink, grid, lds, logR, stage_cap, c, s);
    else if (p.ksize == 51) launch_rolling<51, 4>(b, p, sink, grid, lds, logR, stage_cap, c, s);
    else if (p.ksize <= 32) launch_rolling<0, 2>(b, p, sink, grid, lds, logR, stage_cap, c, s);
    else if (p.ksize <= 64) launch_rolling<0, 4>(b, p, sink, grid, lds, logR, stage_cap, c, s);
    else launch_rolling<0, 8>(b, p, sink, grid, lds, logR, stage_cap, c, s);
    HIP_CHECK(hipGetLastError());
    dev.prof_end("dna_rolling", s);
  } else {
    dev.prof_begin(s);
    hipLaunchKernelGGL(k_dna_generic, dim3(grid_for(span, 256, dev.cu_count() * 16)), dim3(256), 0, s,
                       b_in, p, sink);
    HIP_CHECK(hipGetLastError());
    dev.prof_end("dna_generic", s);
  }
}

// out[0] = records of at least ksize bases, out[1] = positions of the protein arm's six-frame layout (2 (len - 2) per such
// record: len / 3 + (len - 1) / 3 + (len - 2) / 3 == len - 2).  A batch of reads has tens of millions of records; a
// host pass over their offsets costs more than hashing them.
__global__ __launch_bounds__(256) void k_record_stats(const uint64_t* __restrict__ starts, uint32_t nrec, uint32_t ksize,
                                                      unsigned long long* __restrict__ out) {
  unsigned long long nlong = 0, total = 0;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t len = starts[r + 1] - starts[r];
    if (len >= ksize) {
      nlong++;
      if (len >= 2) total += 2 * (len - 2);
      else for (uint32_t f = 0; f < 3; f++) total += len >= f ? 2 * ((len - f) / 3) : 0;
    }
  }
  for (int off = 32; off; off >>= 1) { nlong += __shfl_down(nlong, off); total += __shfl_down(total, off); }
  if ((threadIdx.x & 63) == 0 && (nlong | total)) { atomicAdd(&out[0], nlong); atomicAdd(&out[1], total); }
}
void launch_record_stats(const uint64_t* starts, uint32_t nrec, uint32_t ksize, uint64_t* out2, hipStream_t s) {
  HIP_CHECK(hipMemsetAsync(out2, 0, 16, s));
  hipLaunchKernelGGL(k_record_stats, dim3(grid_for(nrec, 256, 2048)), dim3(256), 0, s, starts, nrec, ksize,
                     reinterpret_cast<unsigned long long*>(out2));
  HIP_CHECK(hipGetLastError());
}

void launch_first_bad_record(const uint64_t* starts, const uint64_t* vends, uint32_t nrec, uint32_t ksize, uint64_t* out, hipStream_t s) {
  HIP_CHECK(hipMemsetAsync(out, 0xff, 8, s));
  hipLaunchKernelGGL(k_first_bad_record, dim3(grid_for(nrec, 256, 2048)), dim3(256), 0, s, starts, vends, nrec, ksize,
                     reinterpret_cast<unsigned long long*>(out));
  HIP_CHECK(hipGetLastError());
}

void launch_first_invalid(const SeqBatch& b, uint64_t* vends_out, hipStream_t s) {
  if (b.len == 0) return;
  hipLaunchKernelGGL(k_first_invalid, dim3(grid_for(b.len + 32, 256 * 16, 8192)), dim3(256), 0, s, b,
                     vends_out);
  HIP_CHECK(hipGetLastError());
}

void launch_translate(const SeqBatch& b, const uint64_t* seg_off, uint32_t nseg, uint64_t total, uint32_t ksize,
                      uint8_t* residues, uint32_t* bad_utf8, hipStream_t s) {
  if (total == 0) return;
  hipLaunchKernelGGL(k_translate, dim3(grid_for(b.len, kTrTile, 16384)), dim3(kTrThreads), 0, s, b, seg_off, nseg, ksize,
                     residues, bad_utf8);
  HIP_CHECK(hipGetLastError());
}

bool launch_protein_fused(const SeqBatch& b_in, const uint64_t* seg_offsets, uint32_t win, const HashParams& p,
                          const CandSink& sink, uint32_t* high_flag, Device& dev, hipStream_t s) {
  if (b_in.len == 0) return true;
  if (!(win == 7 || win == 9 || win == 10)) return false;     // the usual protein k-mer sizes (ksize 21 / 27 / 30)
  int logR = SMH_PF_LOGR;
  while (logR > 5 && (b_in.len >> logR) < (uint64_t)dev.cu_count() * 512 * 2) logR--;
  const uint64_t tile = 512ull << logR;
  const uint64_t ntiles = (b_in.len + tile - 1) / tile;
  const int grid = (int)(ntiles < (uint64_t)dev.cu_count() * 8 ? ntiles : (uint64_t)dev.cu_count() * 8);
  const uint32_t x_bytes = (uint32_t)tile + 3 * win + 12 + 96;
  // two windows per position pass with probability (thr + 1) / 2^64 each
  long double expect = 2.0L * (long double)tile * (((long double)p.thr + 1.0L) / 18446744073709551616.0L);
  uint32_t stage_cap = expect * 2.0L + 64.0L > 2048.0L ? 2048u : (uint32_t)(expect * 2.0L + 64.0L);
  if (stage_cap < 128) stage_cap = 128;
  const size_t lds = 16 + (size_t)stage_cap * 8 * (sink.pos ? 2 : 1) + x_bytes + 4 * ((x_bytes >> logR) + 2);
  const SeqBatch b = with_tile_records(b_in, p.ksize, 0, tile, ntiles, dev, s);
  // the list of span starts for k_spliced_windows: [count][starts], in the shared scratch (used up before this returns'
  // successors on the stream -- the fold's sort -- claim it)
  dev.scratch.ensure(16 + (size_t)kSplicedCap * 8);
  uint32_t* slow_count = dev.scratch.as<uint32_t>();
  uint64_t* slow_list = reinterpret_cast<uint64_t*>(dev.scratch.as<char>() + 16);
  HIP_CHECK(hipMemsetAsync(slow_count, 0, 4, s));
#define SMH_PF(W_) hipLaunchKernelGGL((k_protein_fused<W_, 512>), dim3(grid), dim3(512), lds, s, b, p, sink, logR, stage_cap, high_flag, \
                                      slow_list, slow_count)
  if (win == 7) SMH_PF(7); else if (win == 9) SMH_PF(9); else SMH_PF(10);
#undef SMH_PF
  hipLaunchKernelGGL(k_spliced_windows, dim3(64), dim3(256), 0, s, b, p, sink, win, slow_list, slow_count);
  if (sink.pos)   // (a << 1 | strand) -> residue index of the six-frame layout
    hipLaunchKernelGGL(k_protein_positions, dim3(dev.cu_count() * 4), dim3(256), 0, s, sink.pos, sink.count, sink.capacity, b.starts,
                       b.nrec, b.len, seg_offsets, 3 * win, p.pos_base);
  HIP_CHECK(hipGetLastError());
  return true;
}

void launch_hash_windows(const uint8_t* bytes, uint64_t total, const uint64_t* seg_offsets,
                         uint32_t nseg, uint32_t win, const HashParams& p, const CandSink& sink,
                         hipStream_t s) {
  (void)total;
  if (p.range_hi <= p.range_lo) return;
  const uint32_t stage_cap = 1024;
  const size_t lds = 16 + (size_t)stage_cap * 8 * (sink.pos ? 2 : 1);
  const dim3 grid(grid_for((p.range_hi - p.range_lo + kWinRun - 1) / kWinRun, 256, 16384));
#define SMH_HW(W_) hipLaunchKernelGGL(k_hash_windows<W_>, grid, dim3(256), lds, s, bytes, seg_offsets, nseg, win, p, sink, stage_cap)
  // the usual protein k-mer sizes (ksize 21 / 27 / 30 nucleotides) get static instantiations
  if (win == 7) SMH_HW(7); else if (win == 9) SMH_HW(9); else if (win == 10) SMH_HW(10); else SMH_HW(0);
#undef SMH_HW
  HIP_CHECK(hipGetLastError());
}

void launch_hash_segments(const uint8_t* bytes, const uint64_t* seg_offsets, uint32_t nseg,
                          uint64_t seed, uint64_t* out, hipStream_t s) {
  if (nseg == 0) return;
  hipLaunchKernelGGL(k_hash_segments, dim3((nseg + 255) / 256), dim3(256), 0, s, bytes, seg_offsets,
                     nseg, seed, out);
  HIP_CHECK(hipGetLastError());
}

void launch_filter_hashes(const uint64_t* hashes, const HashParams& p, const CandSink& sink, hipStream_t s) {
  if (p.range_hi <= p.range_lo) return;
  hipLaunchKernelGGL(k_filter_hashes, dim3(grid_for(p.range_hi - p.range_lo, 256, 4096)), dim3(256), 0, s, hashes, p, sink);
  HIP_CHECK(hipGetLastError());
}

void launch_synth_dna(uint8_t* out, uint64_t start, uint64_t len, uint64_t seed, uint64_t n_every,
                      hipStream_t s) {
  if (len == 0) return;
  if (start & 31) throw_internal("synth_dna: start must be a multiple of 32");
  hipLaunchKernelGGL(k_synth_dna, dim3(grid_for((len + 31) / 32, 256, 8192)), dim3(256), 0, s, out,
                     start, len, seed, n_every);
  HIP_CHECK(hipGetLastError());
}

}  // namespace smh
