// kernels.hpp -- launch interfaces of the HIP kernels (sketch_kernels.hip, sort.hip,
// compare_kernels.hip).  Plain structs and pointers; no torch types anywhere.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device.hpp"

namespace smh {

// Where the hash kernels append the hashes that pass the threshold.  `count` keeps counting
// past `capacity`, so an overflowing launch reports exactly how much room a re-run needs.
struct CandSink {
  uint64_t* hash = nullptr;
  uint64_t* pos = nullptr;  // stream position of the k-mer (only when order matters: Q3/Q4)
  unsigned long long* count = nullptr;
  uint64_t capacity = 0;
};

// One flat byte buffer holding 1..nrec records.  Record r is [starts[r], starts[r+1]);
// bytes at or beyond vends[r] are not part of it (force=false truncation, reference
// src/lib.rs:268-273).  starts == nullptr means a single record [0, len) ending at vend0.
struct SeqBatch {
  const uint8_t* seq = nullptr;
  uint64_t len = 0;
  const uint64_t* starts = nullptr;
  const uint64_t* vends = nullptr;
  uint32_t nrec = 1;
  uint64_t vend0 = 0;
  // set by the launchers of the tiled kernels: per tile of the launch, the record that holds its first
  // position (see k_tile_records)
  const struct TileRec* tile_rec = nullptr;
};

struct TileRec {
  uint32_t rec;    // record of the tile's first position
  uint32_t last;   // record of the first position of the NEXT tile (== rec: the tile lies inside one record)
  uint64_t end;    // end of the valid part of `rec`
};

struct HashParams {
  uint32_t ksize = 31;
  uint64_t seed = 42;
  const uint64_t* thr_rec = nullptr;  // DNA arm: per-record thresholds (device, nrec entries) or null
  uint64_t thr = ~0ull;               // the threshold when thr_rec is null (and the LDS stage estimate)
  uint64_t pos_base = 0;
  uint64_t range_lo = 0, range_hi = 0;  // k-mer start positions [lo, hi) handled by this launch
};

// DNA arm of add_sequence (reference src/lib.rs:258-274): canonical k-mer + murmur64 + filter.
// Picks the rolling 2-bit kernel for ksize <= 32 and the byte-wise kernel otherwise.
void launch_dna_hash(const SeqBatch& b, const HashParams& p, const CandSink& sink, Device& dev,
                     hipStream_t s, bool force_generic = false);

// smallest position of a byte outside [ACGTacgt] per record -> vends (atomicMin); vends must be
// pre-filled with the record ends.  (reference src/lib.rs:795-804 _checkdna, applied per byte)
void launch_first_invalid(const SeqBatch& b, uint64_t* vends_out, hipStream_t s);
// *out (device) = the first record that is at least ksize long and whose vends entry lies before its end, else UINT64_MAX
void launch_first_bad_record(const uint64_t* starts, const uint64_t* vends, uint32_t nrec, uint32_t ksize, uint64_t* out, hipStream_t s);
// out2 (device): [0] records of at least ksize bases, [1] positions of the protein arm's six-frame layout
void launch_record_stats(const uint64_t* starts, uint32_t nrec, uint32_t ksize, uint64_t* out2, hipStream_t s);

// launch_hash_segments: murmur64 of whole byte strings (add_word / hash_murmur, reference
// src/lib.rs:247-250, src/ffi.rs:15-24) -> out[i].
// launch_hash_windows: every window of `win` kept residues inside each segment of the translated
// buffer (protein arm second phase, reference src/lib.rs:289-300) -> candidate sink.
void launch_hash_segments(const uint8_t* bytes, const uint64_t* seg_offsets, uint32_t nseg,
                          uint64_t seed, uint64_t* out, hipStream_t s);
void launch_hash_windows(const uint8_t* bytes, uint64_t total, const uint64_t* seg_offsets,
                         uint32_t nseg, uint32_t win, const HashParams& p, const CandSink& sink,
                         hipStream_t s);

// six-frame translation (reference src/lib.rs:277-301, 779-793).  seg_offsets (device, nseg+1
// entries, nseg = 6*nrec) gives where each (record, frame, strand) segment lives in `residues`;
// unknown codons are written as 0xFF and skipped by launch_hash_windows; bad_utf8[seg] is set when
// a codon chunk is not valid UTF-8 (the reference panics there).
void launch_translate(const SeqBatch& b, const uint64_t* seg_offsets, uint32_t nseg, uint64_t total, uint32_t ksize,
                      uint8_t* residues, uint32_t* bad_utf8, hipStream_t s);

// Protein arm in ONE pass over the DNA (translation + hashing, no residue buffer): every window of
// `win` residues of the six frames of the whole batch (p.range_* are not used); candidate positions,
// when the sink wants them, are indices of the six-frame layout (seg_offsets, as above) + p.pos_base.
// b.vend0 (single record) must be 0
// when the record is shorter than p.ksize.  *high_flag is OR-ed with 1 when the batch holds a byte
// >= 0x80: the result must then be discarded and the two-pass path taken (UTF-8 panic semantics).
// Returns false (nothing launched) for window lengths it has no instantiation for.
bool launch_protein_fused(const SeqBatch& b, const uint64_t* seg_offsets, uint32_t win, const HashParams& p,
                          const CandSink& sink, uint32_t* high_flag, Device& dev, hipStream_t s);

// hashes[lo..hi) that are <= thr go to the sink with their index as stream position
// (bulk add_many, reference src/lib.rs:412-417)
void launch_filter_hashes(const uint64_t* hashes, const HashParams& p, const CandSink& sink, hipStream_t s);

// synthetic DNA generator (same definition as oracle osynth_dna)
void launch_synth_dna(uint8_t* out, uint64_t start, uint64_t len, uint64_t seed, uint64_t n_every,
                      hipStream_t s);

// --- sort.hip -----------------------------------------------------------------------
// LSD radix sort, ping-pong between (k0,v0) and (k1,v1); returns which pair holds the result.
// Only the byte passes [first_pass, last_pass) are run (default: all eight): a stable sort by a bit
// field of the key.
int radix_sort_u64(uint64_t* k0, uint64_t* k1, uint64_t* v0, uint64_t* v1, size_t n,
                   DeviceBuffer& scratch, hipStream_t s, int first_pass = 0, int last_pass = 8, uint32_t pass_mask = 0);
// The whole fold of a small batch in ONE launch that reads the candidate count from device memory (no host round trip in
// front of it): up to kSmallFoldMax candidates -> distinct keys ascending in uniq, their run starts in starts (both with room
// for kSmallFoldMax entries), result_dev[0] = candidates, result_dev[1] = runs.  More candidates than it takes (kSmallFoldMax,
// or half of that when `expected` is small enough for the half-size instance), or than `capacity`: result_dev[1] = ~0 and
// nothing else is written.
constexpr uint32_t kSmallFoldMax = 16384;
void small_fold_async(const uint64_t* keys, const unsigned long long* count_dev, uint64_t capacity, uint64_t* uniq, uint32_t* starts,
                      unsigned long long* result_dev, uint32_t expected, hipStream_t s);
// the same with a 32-bit payload (indices): a quarter less traffic per pass
// pass_mask != 0: run exactly the byte passes whose bit is set instead of reading the digit histograms
// back to skip constant bytes -- no host synchronisation inside the sort
int radix_sort_u64_v32(uint64_t* k0, uint64_t* k1, uint32_t* v0, uint32_t* v1, size_t n, DeviceBuffer& scratch,
                       hipStream_t s, uint32_t pass_mask = 0);
int radix_sort_u64_keys(uint64_t* k0, uint64_t* k1, size_t n, DeviceBuffer& scratch, hipStream_t s, uint32_t pass_mask);
// keys[0 .. n) (read only) sorted into k0 or k1 (the return value says which), v0 / v1 alongside = where every sorted key
// was in `keys`: the first pass reads `keys` and numbers them itself -- no copy of the keys, no index array.  pass_mask != 0.
// shift_dev (nullable): a device word added to every pass's shift -- pass p sorts by bits [*shift_dev + 8 p, + 8): the caller's
// passes start at a bit only the device knows (the low end of the 32 most significant bits that vary, see k_key_span).
int radix_sort_u64_place(const uint64_t* keys, uint64_t* k0, uint64_t* k1, uint32_t* v0, uint32_t* v1, size_t n, DeviceBuffer& scratch,
                         hipStream_t s, uint32_t pass_mask, const uint32_t* shift_dev = nullptr);
// run_length_encode_u64 without its read-back: *nruns_dev (device) receives the number of runs;
// skip (device, nullable): non-zero = the launches do nothing
void run_length_encode_u64_async(const uint64_t* keys, size_t n, uint64_t* uniq, uint32_t* starts, DeviceBuffer& scratch,
                                 hipStream_t s, const uint32_t* origin, uint32_t* rank_out, uint32_t* nruns_dev,
                                 const uint32_t* skip, uint32_t* runid_out = nullptr,    // runid_out[i] = run of sorted position i
                                 bool rank_flags = false);   // rank_out values carry bit 31 = run longer than 1, bit 30 = first of its run
// in-place exclusive scan of m u32 counters; *total (device, nullable) receives their sum
void exclusive_scan_u32_dev(uint32_t* d, size_t m, uint32_t* total, DeviceBuffer& scratch, hipStream_t s);
// unique keys + run start indices of a sorted array; returns the number of runs (syncs).
// origin / rank_out (optional): also write rank_out[origin[i]] = run id of sorted position i.
// key2 / uniq2 (optional): a more significant second key (array sorted by (key2, keys)); a run
// ends where either changes and uniq2[run] receives its key2.
uint32_t run_length_encode_u64(const uint64_t* keys, size_t n, uint64_t* uniq, uint32_t* starts,
                               DeviceBuffer& scratch, hipStream_t s, const uint32_t* origin = nullptr,
                               uint32_t* rank_out = nullptr, const uint64_t* key2 = nullptr,
                               uint64_t* uniq2 = nullptr, int key2_shift = 0);   // key2 is compared as key2 >> key2_shift
// Union of a device-resident sketch S (ascending distinct hashes + optional u64 counts, which are UPDATED in place for the
// hashes D also holds) with the fold D of one more batch (ascending distinct hashes + optional run starts): out / out_cnt
// (room for n_s + n_d entries) receive the merged sketch, *n_new_dev the number of hashes that were new (result size =
// n_s + that).  tmp: (2 n_d + n_s + 1) u32 of work space.  No sort, no host round trip.
void sorted_union_async(const uint64_t* S, uint64_t* S_cnt, uint32_t n_s, const uint64_t* D, const uint32_t* D_starts,
                        const uint64_t* D_cnt /* counts of D: these, or run starts, when S_cnt is given */, uint32_t n_d, uint32_t d_total, uint64_t* out, uint64_t* out_cnt, uint32_t* n_new_dev, DeviceBuffer& tmp, DeviceBuffer& scratch,
                        hipStream_t s);
// counts[i] = starts[i+1] - starts[i] (the last run ends at total)
void starts_to_counts(const uint32_t* starts, uint32_t n, uint32_t total, uint64_t* counts, hipStream_t s);
// cand_pos[i] (a k-mer start position of the batch) -> the sketch group of the record holding it
// keep_bits != 0: the position is kept in the low keep_bits bits, the group goes above them
void launch_pos_to_group(uint64_t* pos, uint64_t n, const uint64_t* rec_starts, uint32_t nrec,
                         const uint32_t* group_of_rec, hipStream_t s, int keep_bits = 0);
// reduces the first `nruns` of `total_runs` runs (run u ends at starts[u+1], the last one at n)
void run_reduce(const uint32_t* starts, uint32_t nruns, uint32_t total_runs, uint32_t n,
                const uint64_t* weights, const uint64_t* pos, uint64_t* out_sum, uint64_t* out_minpos,
                hipStream_t s);

// --- compare_kernels.hip ---------------------------------------------------------------
// Sketches as CSR: hashes[offsets[i] .. offsets[i+1]) ascending and unique.
struct SketchSet {
  const uint64_t* hashes = nullptr;
  const uint64_t* offsets = nullptr;  // n+1 entries (device)
  uint32_t n = 0;
  const uint64_t* h_offsets = nullptr;  // the same n+1 entries in host memory when the caller has them (saves a read-back)
};
struct CompareOut {
  uint64_t* common = nullptr;  // |A ^ B ^ bottom_n(A u B)|   (reference src/lib.rs:470-499)
  uint64_t* size = nullptr;    // |bottom_n(A u B)|
  double* jaccard = nullptr;   // common / max(1, size)        (reference src/lib.rs:501-508)
  uint64_t* count_common = nullptr;  // |A ^ B| untruncated     (reference src/lib.rs:428-436)
  double* containment = nullptr;     // |A ^ B| / |A|           (reference src/index.rs:146-154)
};
// rows x cols block; num = truncation length of the union walk (row's `num`; 0 = unbounded),
// row_nums (device, nullable) overrides it per row.
// max_row_len / max_col_len: longest sketch on each side (decides LDS staging).
// one ordered pair from plain device pointers; out = {|A u B|, |A n B|, common within the first n of the union}
struct PairOut { unsigned long long tot_u, tot_c, common; };
void launch_compare_pair(const uint64_t* A, uint32_t la, const uint64_t* B, uint32_t lb, uint64_t n, PairOut* out_dev,
                         Device& dev, hipStream_t s);

// Which kernel serves an N x M block is chosen from its shape.  The choice never changes a result;
// it can be pinned (parity tests force every route over the same inputs, callers may know better).
enum CompareRoute : uint32_t { kRouteAuto = 0, kRouteWave = 1, kRouteFew = 2, kRouteComponents = 3, kRouteTiled = 4 };
struct CompareTuning {
  uint32_t route = kRouteAuto;             // kRouteComponents / kRouteTiled both mean "the block path"; the two
                                           // are told apart by comp_pairs_limit
  uint32_t visit_all_tiles = 0;            // 1: launch every tile, not only those that can hold sharing pairs
  uint32_t use_symmetry = 1;               // all-vs-all with one num: compute the upper triangle, mirror the rest
  uint32_t split_frequent = 1;             // hashes held by a large share of the sketches do not make components
  uint32_t dictionary = 0;                 // 1: the pooled sort of the dictionary with all eight byte passes (default: four + tie fix)
  uint32_t no_range_masks = 0;             // 1: the tiled kernel walks every pair from the first range on (default: range masks)
  uint64_t comp_pairs_limit = 96ull << 10;  // at most this many sharing pairs: per-component pair kernel, else tiled (the pair kernel
                                           // takes ~3.7 ns per pair of num = 2000 sketches, one round of tiles ~0.45 ms: profiles/r03_tile_shape.txt)
};
struct CompareStats {                      // of the last block compare
  uint32_t route = 0;                      // CompareRoute that ran
  uint32_t rows_per_tile = 0;
  uint64_t tiles_visited = 0, tiles_total = 0, pairs_per_tile = 0;   // components route: pairs walked / pairs / 1
  uint64_t lds_overflow_steps = 0;         // tiled: (tile, range) steps merged from global memory instead of LDS
  uint32_t frequent_hashes = 0;            // hashes set aside as frequent (decided from per-sketch positions, not walked)
  uint32_t pipelined = 0;                  // tiled: k_compare_tiled_pf walked the tiles
  uint32_t span_halvings = 0;              // pipelined kernel: stretches rebuilt with a halved span (their speculative span did not fit LDS)
  uint32_t prefetched_after_halving = 0;   // ... tiles in which prefetched crossings were used after such a rebuild
};
void compare_set_tuning(const CompareTuning& t);
CompareTuning compare_get_tuning();
CompareStats compare_last_stats();
void release_compare_scratch();   // frees the tiled kernel's pre-pass buffers
// The dictionary of one collection of sketches resident in HBM (CSR; element t of the collection is hashes_dev[offsets[0] + t]):
// dense ranks of all its hashes, components, frequent hashes, range tables -- built once, by one owner or by `world`
// cooperating owners (each holding the same CSR; owner `rank` sorts slice `rank` of hash space) with ONE all-gather of
// collection_share() between collection_begin and collection_finish; then any number of block compares over it.
// own_mode of collection_compare: 0 every pair of the block; 1 rows == columns == the whole collection with one num
// (upper triangle + mirrors); 2 the rows are one rank's block of the all-vs-all matrix, columns == everything, one num:
// only pairs (i, j) with (j - i) mod N < N/2 (ties: i < j) must be computed here, the others arrive from their owner's rank.
struct CollectionDict;
CollectionDict* collection_begin(const uint64_t* hashes_dev, const uint64_t* offsets_dev /*nullable*/, const uint64_t* offsets_host,
                                 uint32_t n, uint32_t world, uint32_t rank, Device& dev, hipStream_t s);
uint64_t collection_share_bytes(const CollectionDict* d);
const void* collection_share(const CollectionDict* d);
uint32_t collection_len(const CollectionDict* d);
void collection_finish(CollectionDict* d, const void* gathered_dev, Device& dev, hipStream_t s);
void collection_compare(CollectionDict* d, uint32_t row_lo, uint32_t row_hi, uint32_t col_lo, uint32_t col_hi, uint32_t num,
                        const uint32_t* row_nums, uint32_t own_mode, const CompareOut& out, Device& dev, hipStream_t s);
void collection_free(CollectionDict* d);

// the mirrored-block exchange of a matrix computed with own_mode 2 (8-byte elements): the block this rank sends to the rank
// holding rows [col_lo, col_hi) -- its own block's columns, transposed -- and what it keeps of a block it received
void launch_mirror_pack(const void* out, uint32_t n_local, uint32_t n_total, uint32_t col_lo, uint32_t col_hi, void* packed, hipStream_t s);
void launch_mirror_apply(void* out, uint32_t row_lo, uint32_t n_local, uint32_t n_total, uint32_t peer_lo, uint32_t peer_hi, const void* recv,
                         hipStream_t s);

void launch_compare_block(const SketchSet& rows, const SketchSet& cols, uint32_t num,
                          const uint32_t* row_nums, const CompareOut& out, Device& dev,
                          hipStream_t s, uint32_t max_row_len, uint32_t max_col_len, uint64_t nr_elems,
                          uint64_t nc_elems, bool same_sets = false);   // same_sets: rows and cols are one CSR

}  // namespace smh
