// minhash.hpp -- host-side KmerMinHash and the sketching engine that drives the HIP kernels.
//
// Mirrors the reference's `pub struct KmerMinHash` (src/lib.rs:37-46) field for field and its
// methods (src/lib.rs:141-513).  Scalar methods (add_hash, merge, check_compatible) are plain
// host code like the reference's; everything that touches sequence bytes or compares sketches
// runs on the GPU and throws Error(kInternal) when no device is usable -- there is no CPU
// fallback for those.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"
#include "device.hpp"
#include "kernels.hpp"

namespace smh {

// result of hashing one chunk on the device: what must be merged into a sketch
struct Delta {
  std::vector<uint64_t> uniq;       // ascending distinct hashes that passed the filter (kept prefix)
  std::vector<uint32_t> run_start;  // uniq.size()+1 entries: run k is [run_start[k], run_start[k+1])
  std::vector<uint64_t> minpos;     // first stream position of each (empty when not requested)
  int sorted_buf = 0;               // which candidate ping-pong half holds the sorted chunk
  uint64_t pos_mask = ~0ull;        // positions in cand_pos carry a tag above these bits (grouped batches)
  uint64_t n = 0;                   // candidates in the chunk
};

// A sketch that was just built on the GPU stays in HBM (distinct hashes + run boundaries, from
// which the abundances follow) until something on the host asks for it; `mins` / `abunds` are then
// filled by materialize().  Keeps a 10^7-hash scaled sketch out of PCIe unless it is wanted.
struct DeviceSketch {
  DeviceBuffer uniq;    // n ascending distinct u64
  DeviceBuffer starts;  // has_runs: n u32 run starts (abundance k = starts[k+1]-starts[k], last ends at total)
  DeviceBuffer counts;  // has_counts: n u64 abundances (a sketch that has absorbed more than one batch)
  uint64_t n = 0;
  uint64_t total = 0;
  bool has_runs = false;
  bool has_counts = false;
};

// The `mins` vector with a generation counter: reads go through the const forwarding methods, every
// mutation goes through w(), which bumps the generation BEFORE handing out the vector.  A device
// mirror made at generation g is valid exactly while generation() == g -- an O(1), exact test (no
// non-const access to the vector exists outside w(), so no mutation route can be missed).
class TrackedMins {
 public:
  using Vec = std::vector<uint64_t>;
  size_t size() const { return v_.size(); }
  bool empty() const { return v_.empty(); }
  size_t capacity() const { return v_.capacity(); }
  const uint64_t& operator[](size_t i) const { return v_[i]; }
  const uint64_t& back() const { return v_.back(); }
  const uint64_t* data() const { return v_.data(); }
  Vec::const_iterator begin() const { return v_.begin(); }
  Vec::const_iterator end() const { return v_.end(); }
  const Vec& get() const { return v_; }
  operator const Vec&() const { return v_; }
  Vec& w() { ++gen_; return v_; }
  TrackedMins& operator=(const Vec& o) { ++gen_; v_ = o; return *this; }
  TrackedMins& operator=(Vec&& o) { ++gen_; v_ = std::move(o); return *this; }
  TrackedMins& operator=(const TrackedMins& o) { ++gen_; v_ = o.v_; return *this; }
  TrackedMins() = default;
  TrackedMins(const TrackedMins& o) : v_(o.v_) {}
  uint64_t generation() const { return gen_; }
 private:
  Vec v_;
  uint64_t gen_ = 1;
};

// Device copy of a host-resident sketch's `mins`, kept between pairwise calls: compare / count_common
// through the one-pair-per-call ABI would otherwise upload both sketches every time.  Valid while the
// sketch's TrackedMins generation is the one it was made at; never modified in place once published.
struct DeviceMirror {
  void* ptr = nullptr;   // from the device block pool, or adopted from a DeviceSketch
  size_t cap = 0;
  size_t n = 0;
  uint64_t gen = 0;      // TrackedMins::generation() of the vector this is a copy of
  DeviceMirror() = default;
  DeviceMirror(const DeviceMirror&) = delete;
  DeviceMirror& operator=(const DeviceMirror&) = delete;
  ~DeviceMirror();
};

struct KmerMinHash {
  uint32_t num = 1000;
  uint32_t ksize = 21;
  bool is_protein = false;
  uint64_t seed = 42;
  uint64_t max_hash = 0;
  mutable TrackedMins mins;
  bool has_abunds = false;         // Option<Vec<u64>>::is_some()
  mutable std::vector<uint64_t> abunds;
  mutable std::shared_ptr<DeviceSketch> dev;  // non-null: the state lives here, mins/abunds are empty
  mutable std::shared_ptr<DeviceMirror> mirror;  // see DeviceMirror (not copied by Clone)
  // Small add_sequence calls (a read at a time through the legacy ABI) are queued here and hashed
  // in one device batch when the state is next observed or the queue is large: same result as one
  // launch per call, without the per-launch latency.  Errors are still raised by the call itself.
  mutable std::vector<uint8_t> pend_seq;
  mutable std::vector<uint64_t> pend_off;
  // add_word calls are queued the same way: the words are hashed by ONE device launch when the state is next observed
  // (or 64 K words are waiting) and then go through add_hash in the order they came.  The two queues never hold work at
  // the same time (whichever call comes next drains the other first), so the stream order of the calls is kept.
  mutable std::vector<uint8_t> pend_words;
  mutable std::vector<uint64_t> pend_woff;
  void flush_pending() const;
  void flush_words() const;

  KmerMinHash() { mins.w().reserve(1000); }  // Default, src/lib.rs:48-60
  KmerMinHash(uint32_t n, uint32_t k, bool prot, uint64_t seed_, uint64_t mx, bool track);  // 142-174
  KmerMinHash(const KmerMinHash& o);             // Clone: brings a device-resident state to the host first
  KmerMinHash& operator=(const KmerMinHash& o);
  void materialize() const;                      // device-resident state -> mins / abunds
  void to_device_state();                        // scaled sketches: host-resident state -> HBM (uniq + u64 counts)

  void check_compatible(const KmerMinHash& other) const;                 // 176-190
  void add_hash(uint64_t h);                                             // 192-245
  void add_word(const uint8_t* w, size_t len);                           // 247-250 (hash on device)
  void add_sequence(const uint8_t* seq, size_t len, bool force);         // 252-305 (host bytes)
  void merge(const KmerMinHash& other);                                  // 307-403
  void add_from(const KmerMinHash& other);                               // 405-410
  void add_many(const uint64_t* hashes, size_t n);                       // 412-417
  void add_many_bulk(const uint64_t* hashes, size_t n);                  // same, through the device fold
  void add_many_with_abund(const uint64_t* hashes, const uint64_t* counts, size_t n);  // 419-426 ((hash, count) pairs)
  bool merge_on_device(const KmerMinHash& other);                        // large well-formed merges
  uint64_t count_common(const KmerMinHash& other) const;                 // 428-436 (device)
  void intersection_size(const KmerMinHash& other, uint64_t* common, uint64_t* size) const;  // 470-499
  void intersection(const KmerMinHash& other, std::vector<uint64_t>* common, uint64_t* size) const;   // 438-468
  double compare(const KmerMinHash& other) const;                        // 501-508 (device)
  size_t size() const { flush_pending(); return dev ? (size_t)dev->n : mins.size(); }

  // --- batch entry points (additive C ABI) ---
  // Records live in ONE device buffer; h_offsets has nrec+1 host entries.  Semantics: as if
  // add_sequence were called on every record in order; the first record that would have
  // returned Err is reported through *first_error (nullable) after all records were processed.
  void add_sequences_device(const uint8_t* d_seq, uint64_t total_len, const uint64_t* h_offsets,
                            uint32_t nrec, bool force, hipStream_t stream, Error* first_error);
  // Records in HOST memory (what the reference's boundary hands over): uploads them and calls the
  // above; large scaled-DNA batches with force=true are uploaded in chunks on a second stream while
  // the chunks already there are being hashed.
  void add_sequences_host(const uint8_t* h_seq, uint64_t total_len, const uint64_t* h_offsets, uint32_t nrec, bool force);
};

// Many sketches from one batch (additive C ABI smh_add_sequences_grouped): record r feeds
// sketches[group_of_rec[r]].  Semantics per sketch: add_sequences_device over its records in
// order.  Scaled DNA sketches with equal (ksize, seed, max_hash) share one hashing launch and one
// sort; anything else is served sketch by sketch.
void add_sequences_grouped(KmerMinHash* const* sketches, uint32_t n_sketches, const uint8_t* d_seq, uint64_t total_len,
                           const uint64_t* h_offsets, const uint32_t* group_of_rec, uint32_t nrec, bool force,
                           hipStream_t stream, Error* first_error);

// One process-wide workspace: candidate buffers, sort ping-pong, small staging areas.
// Entry points take the Device mutex (recursive).
using HashSourceRef = void*;  // HashSource* of minhash.cpp
class Engine {
 public:
  static Engine& get();

  // hash positions [lo, hi) of a source, keep hashes <= thr: candidates land in cand_hash[0]
  // (and cand_pos[0]); returns how many.  Re-runs once with an exact buffer on overflow.
  uint64_t run_chunk(HashSourceRef src, uint64_t lo, uint64_t hi, uint64_t thr, bool want_pos, hipStream_t s);
  // A chunk expected to leave a few thousand candidates: hash AND fold with one synchronisation (k_small_fold reads the
  // candidate count on the device).  True: `out` holds the sorted distinct hashes and their run starts.  False: *n_out
  // candidates were produced and wait in cand_hash[0] (if they fit *cap_out) for the general path.
  bool run_chunk_small(HashSourceRef src, uint64_t lo, uint64_t hi, uint64_t thr, uint32_t expected, hipStream_t s,
                       DeviceSketch* out, uint64_t* n_out, uint64_t* cap_out);
  // sort the chunk by hash, collapse runs, keep the first `keep` runs (0 = all), fetch them
  // key_bound (0 = unknown): no candidate exceeds it -- the sort then knows which byte passes can differ without reading
  // the digit histograms back (one host round trip less per fold)
  void reduce_chunk(uint64_t n, uint32_t keep, bool have_pos, bool want_minpos, hipStream_t s, Delta* out,
                    DeviceSketch* keep_on_device = nullptr, uint64_t key_bound = 0);

  // murmur64 of whole byte strings on the device (host pointers in, host pointer out)
  void hash_words(const uint8_t* bytes, const uint64_t* offsets, uint32_t n, uint64_t seed, uint64_t* out);

  // one ordered pair (self = a) on mirrored device copies: what the pairwise reference API needs
  struct PairResult { uint64_t common, size, count_common; double jaccard, containment; };
  void compare_pair(const KmerMinHash& a, const KmerMinHash& b, uint32_t num, PairResult* out);
  // the sketch's hashes in device memory: the device-resident state itself, or the mirror of the host vector
  const uint64_t* device_mins(const KmerMinHash& mh, size_t* n, hipStream_t s);
  // a device-resident scaled sketch absorbs the fold of one more batch without leaving HBM (sort.hip sorted_union_async)
  void union_into_device_sketch(KmerMinHash& mh, DeviceSketch& delta, hipStream_t s);
  // the same with the delta given as plain device arrays (ascending distinct hashes, u64 counts or null = 1 each)
  void union_arrays_into_device_sketch(KmerMinHash& mh, const uint64_t* d_mins, const uint64_t* d_counts, uint64_t n, hipStream_t s);

  // block compare of host-resident sketches (uploads them); outputs are row-major rows x cols
  void compare_host(const std::vector<const KmerMinHash*>& rows, const std::vector<const KmerMinHash*>& cols,
                    const uint32_t* row_nums_host, uint32_t num, uint64_t* common, uint64_t* size,
                    double* jaccard, uint64_t* count_common, double* containment);

  // frees every grow-only workspace buffer (they are re-created on demand)
  void release_workspace();

  // CSR upload of host-resident sketches (materialises them; gathers through pinned staging)
  void pack_sketches(const std::vector<const KmerMinHash*>& v, DeviceBuffer& data, DeviceBuffer& offs, SketchSet* out,
                     uint32_t* maxlen, std::vector<uint64_t>* h_off, hipStream_t s);

  DeviceBuffer cand_hash[2], cand_pos[2], counter, uniq, uniq2, starts, red_b, misc, seqbuf, offbuf, vendbuf, vendbuf2, grpbuf, union_tmp;
  PinnedBuffer pin_a, pin_b, pin_pair;
  DeviceBuffer pair_out;
  DeviceBuffer resbuf, segbuf, badbuf, cmp_a, cmp_b, cmp_oa, cmp_ob, cmp_out;

 private:
  Engine() = default;
};

}  // namespace smh
