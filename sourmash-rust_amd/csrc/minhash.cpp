// minhash.cpp -- host logic of KmerMinHash + the engine that feeds the HIP kernels.
//
// Division of labour (DESIGN.md "fold"):
//   device : k-mer walk, canonicalisation, murmur64, threshold filter (sketch_kernels.hip);
//            radix sort, distinct + run boundaries + first position, bottom-n cut (sort.hip);
//            every sketch comparison (compare_kernels.hip)
//   host   : applying the resulting delta (<= num entries in num mode, the distinct retained
//            hashes in scaled mode) to the sorted vector, with the reference's abundance
//            quirk Q3, and the scalar API (add_hash, merge, check_compatible).
#include "minhash.hpp"

#include <algorithm>
#include <mutex>
#include <map>
#include <cstring>

namespace smh {

// ------------------------------------------------------------------------------------
// scalar host methods

KmerMinHash::KmerMinHash(uint32_t n, uint32_t k, bool prot, uint64_t seed_, uint64_t mx, bool track)
    : num(n), ksize(k), is_protein(prot), seed(seed_), max_hash(mx), has_abunds(track) {
  mins.w().reserve(n > 0 ? n : 1000);
  if (track) abunds.reserve(mins.capacity());
}

KmerMinHash::KmerMinHash(const KmerMinHash& o)
    : num(o.num), ksize(o.ksize), is_protein(o.is_protein), seed(o.seed), max_hash(o.max_hash),
      has_abunds(o.has_abunds) {
  o.materialize();   // also drains o's queued sequences
  mins = o.mins;
  abunds = o.abunds;
}

KmerMinHash& KmerMinHash::operator=(const KmerMinHash& o) {
  if (this == &o) return *this;
  o.materialize();
  pend_seq.clear(); pend_off.clear(); pend_words.clear(); pend_woff.clear();
  num = o.num; ksize = o.ksize; is_protein = o.is_protein; seed = o.seed; max_hash = o.max_hash;
  has_abunds = o.has_abunds; mins = o.mins; abunds = o.abunds; dev.reset(); mirror.reset();
  return *this;
}

// Mirror buffers come from the device block pool: a sketch that is created, compared once and
// dropped must not pay a hipMalloc + hipFree per compare.  (No wait for the device when one goes
// back: every call that reads a mirror has synchronised before it returns.)
DeviceMirror::~DeviceMirror() {
  if (ptr) device_pool_free(ptr, cap, false);
}

void KmerMinHash::materialize() const {
  flush_pending();
  if (!dev) return;
  Device& d = Device::get();
  std::lock_guard<std::recursive_mutex> lock(d.mutex());
  hipStream_t s = d.stream();
  const size_t n = (size_t)dev->n;
  d.count("sketch_to_host");   // (evidence for the tests: a sketch that accumulates in HBM must not pass here between batches)
  // Large states come over in 16 MB pieces through two page-locked buffers (link speed instead of a pageable bounce copy),
  // and the vectors are APPENDED to: `resize` would first write 80 MB of zeros for a 10^7-hash sketch.
  Engine& E = Engine::get();
  constexpr size_t kPiece = 16u << 20;
  auto fetch = [&](const void* src, size_t bytes, auto&& sink) {     // sink(const uint8_t* piece, size_t piece_bytes), in order
    if (bytes <= (1u << 20)) {
      std::vector<uint8_t> tmp(bytes);
      if (bytes) HIP_CHECK(hipMemcpyAsync(tmp.data(), src, bytes, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      sink(tmp.data(), bytes);
      return;
    }
    PinnedBuffer* pin[2] = {&E.pin_a, &E.pin_b};
    pin[0]->ensure(kPiece); pin[1]->ensure(kPiece);
    size_t off = 0, prev_bytes = 0;
    int which = 0;
    while (off < bytes || prev_bytes) {
      const size_t cur = std::min(kPiece, bytes - off);
      if (cur) HIP_CHECK(hipMemcpyAsync(pin[which]->ptr, (const uint8_t*)src + off, cur, hipMemcpyDeviceToHost, s));
      if (prev_bytes) sink(pin[which ^ 1]->as<uint8_t>(), prev_bytes);    // the piece before this one, while this one is in flight
      HIP_CHECK(hipStreamSynchronize(s));
      off += cur; prev_bytes = cur; which ^= 1;
    }
  };
  std::vector<uint64_t>& hm = mins.w();
  hm.clear(); hm.reserve(n);
  fetch(dev->uniq.ptr, n * 8, [&](const uint8_t* p, size_t b) { hm.insert(hm.end(), (const uint64_t*)p, (const uint64_t*)(p + b)); });
  if (has_abunds) {
    abunds.clear(); abunds.reserve(n);
    if (dev->has_counts) {
      fetch(dev->counts.ptr, n * 8, [&](const uint8_t* p, size_t b) { abunds.insert(abunds.end(), (const uint64_t*)p, (const uint64_t*)(p + b)); });
    } else {
      // run starts -> abundances, piece by piece (the end of a piece's last run is the next piece's first start)
      bool have = false;
      uint32_t last = 0;
      fetch(dev->starts.ptr, n * 4, [&](const uint8_t* p, size_t b) {
        const uint32_t* st = (const uint32_t*)p;
        const size_t k = b / 4;
        for (size_t i = 0; i < k; i++) {
          if (have) abunds.push_back((uint64_t)(st[i] - last));
          last = st[i]; have = true;
        }
      });
      if (have) abunds.push_back((uint64_t)((uint32_t)dev->total - last));
    }
  }
  // the device copy stays on as the mirror of the host vector: a compare right after needs no upload
  auto m = std::make_shared<DeviceMirror>();
  std::swap(m->ptr, dev->uniq.ptr);
  std::swap(m->cap, dev->uniq.bytes);
  m->n = n;
  m->gen = mins.generation();
  mirror = m;
  dev.reset();
}

void KmerMinHash::check_compatible(const KmerMinHash& o) const {
  if (ksize != o.ksize) throw_mismatch(kMismatchKSizes);
  if (is_protein != o.is_protein) throw_mismatch(kMismatchDNAProt);
  if (max_hash != o.max_hash) throw_mismatch(kMismatchMaxHash);
  if (seed != o.seed) throw_mismatch(kMismatchSeed);
}

// reference src/lib.rs:192-245 (quirks Q3, Q4)
void KmerMinHash::add_hash(uint64_t hash) {
  materialize();
  const uint64_t current_max = mins.empty() ? UINT64_MAX : mins.back();
  if (!(hash <= max_hash || max_hash == 0)) return;
  if (mins.empty()) {
    mins.w().push_back(hash);
    if (has_abunds) abunds.push_back(1);
    return;
  }
  if (hash <= max_hash || current_max > hash || (uint32_t)mins.size() < num) {
    size_t pos = std::lower_bound(mins.begin(), mins.end(), hash) - mins.begin();
    if (pos == mins.size()) {
      mins.w().push_back(hash);
      if (has_abunds) abunds.push_back(1);
    } else if (mins[pos] != hash) {
      { auto& v = mins.w(); v.insert(v.begin() + pos, hash); }
      if (has_abunds) {
        if (pos > abunds.size()) throw_panic("insertion index is out of bounds");
        abunds.insert(abunds.begin() + pos, 1);
      }
      if (num != 0 && mins.size() > (size_t)num) {
        mins.w().pop_back();
        if (has_abunds && !abunds.empty()) abunds.pop_back();
      }
    } else if (has_abunds) {
      if (pos >= abunds.size()) throw_panic("index out of bounds");
      abunds[pos] += 1;
    }
  }
}

void KmerMinHash::add_from(const KmerMinHash& other) {
  other.materialize();
  for (uint64_t h : other.mins) add_hash(h);
}
// reference src/lib.rs:419-426: `for item in hashes { for _ in 0..item.1 { self.add_hash(item.0) } }`.
// After the first add_hash of an item the sketch either holds the hash or is unchanged, and every
// further add_hash of the same value meets the same state (only one abundance moves), so the inner
// loop's repeats 2..count collapse into one more add_hash plus an O(1) bump of the slot it hit --
// same result, panics included, without looping over an abundance of millions.
void KmerMinHash::add_many_with_abund(const uint64_t* hashes, const uint64_t* counts, size_t n) {
  materialize();
  for (size_t i = 0; i < n; i++) {
    const uint64_t h = hashes[i], c = counts[i];
    if (c == 0) continue;
    add_hash(h);
    if (c == 1) continue;
    if (!has_abunds) {
      // untracked: a repeat can still change nothing but is evaluated once for its panics
      add_hash(h);
      continue;
    }
    const size_t pos = std::lower_bound(mins.begin(), mins.end(), h) - mins.begin();
    const uint64_t before = (pos < mins.size() && mins[pos] == h && pos < abunds.size()) ? abunds[pos] : 0;
    add_hash(h);                                  // repeat no. 2 (may panic exactly like the reference)
    const bool bumped = pos < mins.size() && mins[pos] == h && pos < abunds.size() && abunds[pos] == before + 1;
    if (bumped) abunds[pos] += c - 2;             // repeats 3..c hit the same slot
  }
}

void KmerMinHash::add_many(const uint64_t* hashes, size_t n) {
  // a handful of hashes: the reference's loop.  A bulk array: same result through the device fold
  // (filter + sort + distinct + count), which does not degrade quadratically like Vec::insert.
  if (n < 4096) {
    for (size_t i = 0; i < n; i++) add_hash(hashes[i]);
    return;
  }
  add_many_bulk(hashes, n);
}

// reference src/lib.rs:307-403 (quirks Q5, Q6): the abundance iterators advance exactly as there
void KmerMinHash::merge(const KmerMinHash& other) {
  check_compatible(other);
  materialize();
  other.materialize();
  if (merge_on_device(other)) return;
  std::vector<uint64_t> merged, mab;
  merged.reserve(mins.size() + other.mins.size());
  mab.reserve(mins.size() + other.mins.size());
  size_t si = 0, oi = 0, sai = 0, oai = 0;
  const bool s_has = has_abunds, o_has = other.has_abunds;
  while (si < mins.size()) {
    const uint64_t value = mins[si];
    if (oi >= other.mins.size()) {
      merged.insert(merged.end(), mins.begin() + si, mins.end());
      si = mins.size();
      if (s_has && sai < abunds.size()) mab.insert(mab.end(), abunds.begin() + sai, abunds.end());
      sai = abunds.size();
      break;
    }
    const uint64_t x = other.mins[oi];
    if (x < value) {
      merged.push_back(x); oi++;
      if (o_has && oai < other.abunds.size()) mab.push_back(other.abunds[oai++]);
    } else if (x == value) {
      merged.push_back(x); oi++; si++;
      if (o_has && oai < other.abunds.size()) {
        uint64_t v = other.abunds[oai++];
        if (s_has && sai < abunds.size()) mab.push_back(v + abunds[sai++]);
      }
    } else {
      merged.push_back(value); si++;
      if (s_has && sai < abunds.size()) mab.push_back(abunds[sai++]);
    }
  }
  merged.insert(merged.end(), other.mins.begin() + oi, other.mins.end());
  if (o_has && oai < other.abunds.size())
    mab.insert(mab.end(), other.abunds.begin() + oai, other.abunds.end());
  if (!(merged.size() < (size_t)num || num == 0)) merged.resize(num);
  mins.w().swap(merged);
  abunds.swap(mab);
  has_abunds = true;  // Q5: Some(..) even when nothing was tracked, and never truncated
}

// ------------------------------------------------------------------------------------
// engine: hashing a position range into candidates, reducing candidates to a delta

Engine& Engine::get() {
  static Engine* e = new Engine();
  return *e;
}

namespace {

// something whose k-mers / windows can be hashed over an index range of its position space
struct HashSource {
  virtual ~HashSource() = default;
  virtual uint64_t positions() const = 0;
  virtual void launch(uint64_t lo, uint64_t hi, uint64_t thr, const CandSink& sink, hipStream_t s) = 0;
  // The same without any host round trip inside.  Returns null, or a device word the caller reads back together with
  // its own results: non-zero means this launch's candidates must be discarded and launch() used instead.
  virtual const uint32_t* launch_optimistic(uint64_t lo, uint64_t hi, uint64_t thr, const CandSink& sink, hipStream_t s) {
    launch(lo, hi, thr, sink, s);
    return nullptr;
  }
};

struct DnaSource : HashSource {
  SeqBatch b;
  uint32_t ksize = 0;
  uint64_t seed = 0;
  Device* dev = nullptr;
  uint64_t positions() const override { return b.len; }
  const uint64_t* thr_rec = nullptr;   // per-record thresholds (grouped bottom-num batches)
  void launch(uint64_t lo, uint64_t hi, uint64_t thr, const CandSink& sink, hipStream_t s) override {
    HashParams p;
    p.ksize = ksize; p.seed = seed; p.thr = thr; p.thr_rec = thr_rec; p.range_lo = lo; p.range_hi = hi;
    launch_dna_hash(b, p, sink, *dev, s);
  }
};

// The protein arm.  A launch over the WHOLE position space takes the fused kernel (one pass over
// the DNA, no residue buffer); if that kernel met a byte >= 0x80 (where the reference's
// str::from_utf8 may panic) its output is discarded and the launch is repeated on the two-pass path
// -- k_translate into E.resbuf (done once, on first need) + k_hash_windows -- which also serves
// launches over a part of the position space and window lengths the fused kernel has no
// instantiation for.
struct ProteinSource : HashSource {
  SeqBatch b;
  // The segment table (6 frames per record) of the six-frame layout: only the two-pass path and callers that want
  // positions need it, and for a batch of reads it is six entries per read -- built and uploaded on demand.
  std::vector<uint64_t> seg;           // host copy
  const uint64_t* seg_off = nullptr;   // device copy (null until ensure_segments)
  const uint64_t* h_offsets = nullptr; // the caller's record offsets (valid for the duration of the call)
  uint32_t nseg = 0;
  void ensure_segments(hipStream_t s);
  uint64_t total = 0;
  uint32_t win = 0, ksize = 0;
  uint64_t seed = 0;
  Device* dev = nullptr;
  Engine* eng = nullptr;
  bool* have_error = nullptr;
  Error* err = nullptr;
  bool translated = false;
  uint64_t positions() const override { return total; }
  void translate(hipStream_t s);
  // the fused kernel alone; the non-ASCII flag it raises is left for the caller to read (launch() decides then)
  const uint32_t* launch_optimistic(uint64_t lo, uint64_t hi, uint64_t thr, const CandSink& sink, hipStream_t s) override {
    if (translated || lo != 0 || hi != total || sink.pos) { launch(lo, hi, thr, sink, s); return nullptr; }
    HashParams p;
    p.seed = seed; p.thr = thr; p.ksize = ksize;
    p.range_lo = 0; p.range_hi = ~0ull;
    eng->badbuf.ensure(8);
    uint32_t* flag = eng->badbuf.as<uint32_t>();
    HIP_CHECK(hipMemsetAsync(flag, 0, 4, s));
    dev->prof_begin(s);
    if (!launch_protein_fused(b, seg_off, win, p, sink, flag, *dev, s)) {
      dev->prof_end("protein_fused_unsupported", s);
      launch(lo, hi, thr, sink, s);                         // other window lengths: the two-pass path
      return nullptr;
    }
    dev->prof_end("protein_fused", s);
    return flag;
  }
  void launch(uint64_t lo, uint64_t hi, uint64_t thr, const CandSink& sink, hipStream_t s) override {
    HashParams p;
    p.seed = seed; p.thr = thr; p.ksize = ksize;
    if (!translated && lo == 0 && hi == total) {
      p.range_lo = 0; p.range_hi = ~0ull;
      eng->badbuf.ensure(8);
      uint32_t* flag = eng->badbuf.as<uint32_t>();
      HIP_CHECK(hipMemsetAsync(flag, 0, 4, s));
      if (sink.pos) ensure_segments(s);                     // positions are residue indices of the six-frame layout
      dev->prof_begin(s);
      const bool ran = launch_protein_fused(b, seg_off, win, p, sink, flag, *dev, s);
      if (ran) {
        dev->prof_end("protein_fused", s);
        uint32_t high = 0;
        HIP_CHECK(hipMemcpyAsync(&high, flag, 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (!high) return;
        HIP_CHECK(hipMemsetAsync(sink.count, 0, 8, s));   // discard: the two-pass path decides what a non-ASCII byte does
      } else {
        dev->prof_end("protein_fused_unsupported", s);
      }
    }
    ensure_segments(s);
    if (!translated) translate(s);
    p.range_lo = lo; p.range_hi = hi;
    dev->prof_begin(s);
    launch_hash_windows(eng->resbuf.as<uint8_t>(), total, seg_off, nseg, win, p, sink, s);
    dev->prof_end("hash_windows", s);
  }
};

struct RawHashSource : HashSource {
  const uint64_t* hashes = nullptr;  // device
  uint64_t n = 0;
  uint64_t positions() const override { return n; }
  void launch(uint64_t lo, uint64_t hi, uint64_t thr, const CandSink& sink, hipStream_t s) override {
    HashParams p;
    p.thr = thr; p.range_lo = lo; p.range_hi = hi;
    launch_filter_hashes(hashes, p, sink, s);
  }
};

uint64_t estimate_capacity(uint64_t span, uint64_t thr) {
  // expected number of uniform 64-bit hashes <= thr among `span`, with head-room
  long double frac = ((long double)thr + 1.0L) / 18446744073709551616.0L;
  long double e = (long double)span * frac;
  uint64_t cap = (uint64_t)(e * 1.25L) + 65536;
  return cap < span ? cap : span;
}

}  // namespace

uint64_t Engine::run_chunk(HashSourceRef src_, uint64_t lo, uint64_t hi, uint64_t thr, bool want_pos,
                           hipStream_t s) {
  HashSource& src = *static_cast<HashSource*>(src_);
  uint64_t cap = estimate_capacity(hi - lo, thr);
  for (int attempt = 0; attempt < 2; attempt++) {
    if (cap >= (1ull << 31)) throw_internal("candidate set of one chunk exceeds 2^31 entries");
    if (cap == 0) cap = 1;
    cand_hash[0].ensure(cap * 8);
    cand_hash[1].ensure(cap * 8);
    if (want_pos) { cand_pos[0].ensure(cap * 8); cand_pos[1].ensure(cap * 8); }
    counter.ensure(8);
    HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 8, s));
    CandSink sink;
    sink.hash = cand_hash[0].as<uint64_t>();
    sink.pos = want_pos ? cand_pos[0].as<uint64_t>() : nullptr;
    sink.count = counter.as<unsigned long long>();
    sink.capacity = cap;
    src.launch(lo, hi, thr, sink, s);
    unsigned long long n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, counter.ptr, 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (n <= cap) return n;
    cap = n;  // the counter kept counting: exact size for the re-run
  }
  throw_internal("candidate buffer overflow after re-run");
}

bool Engine::run_chunk_small(HashSourceRef src_, uint64_t lo, uint64_t hi, uint64_t thr, uint32_t expected, hipStream_t s,
                             DeviceSketch* out, uint64_t* n_out, uint64_t* cap_out) {
  HashSource& src = *static_cast<HashSource*>(src_);
  uint64_t cap = estimate_capacity(hi - lo, thr);
  if (cap == 0) cap = 1;
  cand_hash[0].ensure(cap * 8);
  cand_hash[1].ensure(cap * 8);
  counter.ensure(8);
  misc.ensure(16);
  uniq.ensure((size_t)kSmallFoldMax * 8);
  starts.ensure(((size_t)kSmallFoldMax + 1) * 4);
  HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 8, s));
  CandSink sink;
  sink.hash = cand_hash[0].as<uint64_t>();
  sink.pos = nullptr;
  sink.count = counter.as<unsigned long long>();
  sink.capacity = cap;
  const uint32_t* redo_flag = src.launch_optimistic(lo, hi, thr, sink, s);
  small_fold_async(cand_hash[0].as<uint64_t>(), counter.as<unsigned long long>(), cap, uniq.as<uint64_t>(), starts.as<uint32_t>(),
                   misc.as<unsigned long long>(), expected, s);
  unsigned long long res[2] = {0, 0};
  uint32_t redo = 0;
  HIP_CHECK(hipMemcpyAsync(res, misc.ptr, 16, hipMemcpyDeviceToHost, s));
  if (redo_flag) HIP_CHECK(hipMemcpyAsync(&redo, redo_flag, 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (redo) {   // (a byte >= 0x80 in a protein batch: rare) the checked launch, then the general path
    HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 8, s));
    src.launch(lo, hi, thr, sink, s);
    unsigned long long n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, counter.ptr, 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    *n_out = n; *cap_out = cap;
    return false;
  }
  *n_out = res[0]; *cap_out = cap;
  if (res[1] == ~0ull) return false;
  std::swap(out->uniq.ptr, uniq.ptr);
  std::swap(out->uniq.bytes, uniq.bytes);
  std::swap(out->starts.ptr, starts.ptr);
  std::swap(out->starts.bytes, starts.bytes);
  out->n = res[1];
  out->total = res[0];
  out->has_runs = true;
  return true;
}

void Engine::reduce_chunk(uint64_t n, uint32_t keep, bool have_pos, bool want_minpos, hipStream_t s,
                          Delta* out, DeviceSketch* keep_on_device, uint64_t key_bound) {
  Device& dev = Device::get();
  out->uniq.clear(); out->run_start.clear(); out->minpos.clear();
  out->sorted_buf = 0; out->n = n;
  if (n == 0) return;
  uint32_t pass_mask = 0;   // the bytes up to the bound's highest non-zero one can differ; the ones above are zero
  if (key_bound) for (int p = 0; p < 8; p++) if ((key_bound >> (8 * p)) != 0) pass_mask |= 1u << p;
  int cur = radix_sort_u64(cand_hash[0].as<uint64_t>(), cand_hash[1].as<uint64_t>(),
                           have_pos ? cand_pos[0].as<uint64_t>() : nullptr,
                           have_pos ? cand_pos[1].as<uint64_t>() : nullptr, n, dev.scratch, s, 0, 8, pass_mask);
  out->sorted_buf = cur;
  uniq.ensure(n * 8);
  starts.ensure((n + 1) * 4);
  uint32_t nruns = run_length_encode_u64(cand_hash[cur].as<uint64_t>(), n, uniq.as<uint64_t>(),
                                         starts.as<uint32_t>(), dev.scratch, s);
  const uint32_t kept = (keep != 0 && nruns > keep) ? keep : nruns;
  if (keep_on_device) {
    // hand the two arrays over instead of copying them out: the sketch stays in HBM
    std::swap(keep_on_device->uniq.ptr, uniq.ptr);
    std::swap(keep_on_device->uniq.bytes, uniq.bytes);
    std::swap(keep_on_device->starts.ptr, starts.ptr);
    std::swap(keep_on_device->starts.bytes, starts.bytes);
    keep_on_device->n = kept;
    keep_on_device->total = n;
    keep_on_device->has_runs = true;
    return;
  }
  out->uniq.resize(kept);
  out->run_start.resize(kept + 1);
  HIP_CHECK(hipMemcpyAsync(out->uniq.data(), uniq.ptr, (size_t)kept * 8, hipMemcpyDeviceToHost, s));
  // run k is [run_start[k], run_start[k+1]); the end of the last kept run is the next run's
  // start, or n when nothing follows
  const uint32_t fetch = kept < nruns ? kept + 1 : kept;
  HIP_CHECK(hipMemcpyAsync(out->run_start.data(), starts.ptr, (size_t)fetch * 4, hipMemcpyDeviceToHost, s));
  if (want_minpos && have_pos && kept) {
    red_b.ensure((size_t)kept * 8);
    run_reduce(starts.as<uint32_t>(), kept, nruns, (uint32_t)n, nullptr, cand_pos[cur].as<uint64_t>(), nullptr,
               red_b.as<uint64_t>(), s);
    out->minpos.resize(kept);
    HIP_CHECK(hipMemcpyAsync(out->minpos.data(), red_b.ptr, (size_t)kept * 8, hipMemcpyDeviceToHost, s));
  }
  HIP_CHECK(hipStreamSynchronize(s));
  if (fetch == kept) out->run_start[kept] = (uint32_t)n;
}

namespace {

enum Mode { kScaled, kNum, kSequential };

Mode mode_of(const KmerMinHash& mh) {
  const bool ab_ok = !mh.has_abunds || mh.abunds.size() == mh.mins.size();
  if (mh.num == 0 && mh.max_hash > 0 && ab_ok) return kScaled;
  if (mh.num > 0 && mh.max_hash == 0 && mh.mins.size() <= (size_t)mh.num && ab_ok) return kNum;
  return kSequential;
}

// scaled mode (num == 0, max_hash > 0): set union, counts add up (reference add_hash with
// `hash <= max_hash` always true: insert or increment, never pop)
void apply_scaled(KmerMinHash& mh, Delta& d) {
  if (d.uniq.empty()) return;
  if (mh.mins.empty()) {
    mh.mins.w().swap(d.uniq);
    if (mh.has_abunds) {
      mh.abunds.resize(mh.mins.size());
      for (size_t k = 0; k < mh.mins.size(); k++) mh.abunds[k] = d.run_start[k + 1] - d.run_start[k];
    }
    return;
  }
  std::vector<uint64_t> nm, na;
  nm.reserve(mh.mins.size() + d.uniq.size());
  if (mh.has_abunds) na.reserve(nm.capacity());
  size_t i = 0, j = 0;
  auto cnt = [&](size_t k) { return (uint64_t)(d.run_start[k + 1] - d.run_start[k]); };
  while (i < mh.mins.size() || j < d.uniq.size()) {
    if (j >= d.uniq.size() || (i < mh.mins.size() && mh.mins[i] < d.uniq[j])) {
      nm.push_back(mh.mins[i]);
      if (mh.has_abunds) na.push_back(mh.abunds[i]);
      i++;
    } else if (i >= mh.mins.size() || d.uniq[j] < mh.mins[i]) {
      nm.push_back(d.uniq[j]);
      if (mh.has_abunds) na.push_back(cnt(j));
      j++;
    } else {
      nm.push_back(mh.mins[i]);
      if (mh.has_abunds) na.push_back(mh.abunds[i] + cnt(j));
      i++; j++;
    }
  }
  mh.mins.w().swap(nm);
  if (mh.has_abunds) mh.abunds.swap(na);
}

// num mode (num > 0, max_hash == 0).  mins = bottom-num of the union.  Abundances follow the
// closed form of quirk Q3: every final element gets old + new occurrences, except
// that when the final sketch is full its LAST element only counts the occurrences up to the
// stream position T* at which the sketch reached its final content (T* = latest first occurrence
// over the elements that were not present before; nothing new => no occurrence counts).
void apply_num(KmerMinHash& mh, const Delta& d, Engine& E, hipStream_t s) {
  if (d.uniq.empty()) return;
  const bool track = mh.has_abunds;
  auto cnt = [&](size_t k) { return (uint64_t)(d.run_start[k + 1] - d.run_start[k]); };
  std::vector<uint64_t> nm, na;
  nm.reserve(mh.num);
  if (track) na.reserve(mh.num);
  // provenance of the last survivor
  bool last_old = false, last_new = false;
  size_t last_j = 0;
  uint64_t last_old_ab = 0;
  bool any_new = false;
  uint64_t tstar = 0;
  size_t i = 0, j = 0;
  while (nm.size() < (size_t)mh.num && (i < mh.mins.size() || j < d.uniq.size())) {
    if (j >= d.uniq.size() || (i < mh.mins.size() && mh.mins[i] < d.uniq[j])) {
      nm.push_back(mh.mins[i]);
      if (track) { na.push_back(mh.abunds[i]); last_old = true; last_new = false; last_old_ab = mh.abunds[i]; }
      i++;
    } else if (i >= mh.mins.size() || d.uniq[j] < mh.mins[i]) {
      nm.push_back(d.uniq[j]);
      if (track) {
        na.push_back(cnt(j));
        last_old = false; last_new = true; last_j = j; last_old_ab = 0;
        any_new = true;
        tstar = std::max(tstar, d.minpos[j]);
      }
      j++;
    } else {
      nm.push_back(mh.mins[i]);
      if (track) {
        na.push_back(mh.abunds[i] + cnt(j));
        last_old = true; last_new = true; last_j = j; last_old_ab = mh.abunds[i];
      }
      i++; j++;
    }
  }
  if (track && nm.size() == (size_t)mh.num && last_new) {
    uint64_t c = 0;
    if (any_new) {
      const uint32_t lo = d.run_start[last_j], hi = d.run_start[last_j + 1];
      std::vector<uint64_t> pos(hi - lo);
      HIP_CHECK(hipMemcpyAsync(pos.data(), E.cand_pos[d.sorted_buf].as<uint64_t>() + lo, (size_t)(hi - lo) * 8,
                               hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      for (uint64_t p : pos) c += (p & d.pos_mask) <= tstar;
    }
    na.back() = (last_old ? last_old_ab : 0) + c;
  }
  mh.mins.w().swap(nm);
  if (track) mh.abunds.swap(na);
}

void ingest(KmerMinHash& mh, HashSource& src, hipStream_t s) {
  Engine& E = Engine::get();
  Device& dev = Device::get();
  const uint64_t P = src.positions();
  if (P == 0) return;
  const Mode mode = mode_of(mh);

  if (mode == kScaled) {
    // one launch over as many positions as keep the expected candidate count under 2^30: the
    // whole 10 GB benchmark batch is a single launch + a single sort + a single merge
    long double frac = ((long double)mh.max_hash + 1.0L) / 18446744073709551616.0L;
    long double span_ld = (long double)(1ull << 30) / frac;
    const uint64_t CH = span_ld > 4.0e12L ? (uint64_t)4e12 : (span_ld < 16777216.0L ? (1ull << 24) : (uint64_t)span_ld);
    for (uint64_t lo = 0; lo < P; lo += CH) {
      const uint64_t hi = std::min(P, lo + CH);
      // the state is in HBM, or there is none yet: it stays (or starts) there -- the batch's fold is united with it on
      // the device and nothing crosses the link until an accessor asks.  A state that an accessor has already brought to
      // the host is merged there (apply_scaled).
      const bool empty = mh.mins.empty() && !mh.dev;
      const bool in_hbm = empty || mh.dev;
      const bool whole_into_empty = empty && lo == 0 && hi == P;
      uint64_t n = 0;
      bool hashed = false;
      if (whole_into_empty && (long double)P * frac < 0.7L * kSmallFoldMax) {
        // one genome per call: a few thousand candidates -- hash, sort, collapse and count with ONE synchronisation
        auto ds = std::make_shared<DeviceSketch>();
        uint64_t cap = 0;
        if (E.run_chunk_small(&src, lo, hi, mh.max_hash, (uint32_t)((long double)P * frac), s, ds.get(), &n, &cap)) {
          if (n > 0) mh.dev = ds;
          return;
        }
        hashed = n <= cap;   // more than the small fold takes (repeats): the candidates are there for the general path
      }
      if (!hashed) n = E.run_chunk(&src, lo, hi, mh.max_hash, false, s);
      Delta d;
      if (in_hbm) {
        if (n == 0) continue;
        auto ds = std::make_shared<DeviceSketch>();
        E.reduce_chunk(n, 0, false, false, s, &d, ds.get(), mh.max_hash);
        if (!mh.dev) mh.dev = ds;          // empty sketch: the sorted distinct hashes ARE the new state
        else E.union_into_device_sketch(mh, *ds, s);
        continue;
      }
      E.reduce_chunk(n, 0, false, false, s, &d, nullptr, mh.max_hash);
      apply_scaled(mh, d);
    }
    return;
  }

  if (mode == kNum) {
    // The first chunk is small because every hash passes while the sketch is not full; after it
    // the sketch's own maximum is the filter and chunks grow geometrically.
    const bool track = mh.has_abunds;
    {
      // One pass when the input behaves like random sequence: keep hashes under the value below
      // which about 2 num + 64 of the P windows are expected (or under the full sketch's own
      // maximum when that is lower).  If at least num distinct hashes show up, they are the
      // bottom-num of the whole input; otherwise (few distinct k-mers) nothing has been applied
      // yet and the growing-chunk loop below takes over.
      const bool full = mh.mins.size() >= (size_t)mh.num;
      const uint64_t natural = full ? mh.mins.back() : UINT64_MAX;
      const long double want = 2.0L * mh.num + 64.0L;
      uint64_t thr = natural;
      if (want < 0.5L * (long double)P) {
        const uint64_t est = (uint64_t)(want / (long double)P * 18446744073709551616.0L);
        if (est < thr) thr = est;
      }
      if (P < (1ull << 40)) {
        uint64_t n = 0;
        bool hashed = false;
        if (!track && mh.mins.empty() && !mh.dev && want < 0.7L * kSmallFoldMax) {
          // one genome per call, no abundances: hash, sort, collapse and count with ONE synchronisation; the sketch is
          // the first `num` of the distinct candidates and stays in HBM until somebody looks at it
          auto ds = std::make_shared<DeviceSketch>();
          uint64_t cap = 0;
          if (E.run_chunk_small(&src, 0, P, thr, (uint32_t)want, s, ds.get(), &n, &cap)) {
            if (thr == natural || ds->n >= (uint64_t)mh.num) {
              if (ds->n > (uint64_t)mh.num) ds->n = mh.num;
              if (ds->n > 0) mh.dev = ds;
              return;
            }
            n = ~0ull;          // fewer than num distinct hashes: nothing applied, the growing-chunk loop takes over
          } else {
            hashed = n <= cap;  // too many candidates for the small fold (repeats): they wait in cand_hash[0]
          }
        }
        if (n != ~0ull) {
        if (!hashed) n = E.run_chunk(&src, 0, P, thr, track, s);
        Delta d;
        E.reduce_chunk(n, mh.num, track, track, s, &d, nullptr, thr == UINT64_MAX ? 0 : thr);
        if (thr == natural || d.uniq.size() >= (size_t)mh.num) {
          apply_num(mh, d, E, s);
          return;
        }
        }
      }
    }
    uint64_t chunk = std::max<uint64_t>(1u << 16, (uint64_t)mh.num * 64);
    for (uint64_t lo = 0; lo < P;) {
      const uint64_t hi = std::min(P, lo + chunk);
      const bool full = mh.mins.size() >= (size_t)mh.num;
      const uint64_t thr = full ? mh.mins.back() : UINT64_MAX;
      const uint64_t n = E.run_chunk(&src, lo, hi, thr, track, s);
      Delta d;
      E.reduce_chunk(n, mh.num, track, track, s, &d, nullptr, thr == UINT64_MAX ? 0 : thr);
      apply_num(mh, d, E, s);
      lo = hi;
      if (chunk < (1ull << 30)) chunk *= 8;
    }
    return;
  }

  // Order-dependent parameter combinations (num and max_hash both zero or both non-zero, or a
  // sketch whose vectors were pushed out of shape through the raw ABI): hash on the device,
  // then replay the reference's add_hash over the survivors in stream order.
  const uint64_t thr = mh.max_hash > 0 ? mh.max_hash : UINT64_MAX;
  const uint64_t CH = 1ull << 24;
  std::vector<uint64_t> hs;
  for (uint64_t lo = 0; lo < P; lo += CH) {
    const uint64_t hi = std::min(P, lo + CH);
    const uint64_t n = E.run_chunk(&src, lo, hi, thr, true, s);
    if (n == 0) continue;
    int cur = radix_sort_u64(E.cand_pos[0].as<uint64_t>(), E.cand_pos[1].as<uint64_t>(),
                             E.cand_hash[0].as<uint64_t>(), E.cand_hash[1].as<uint64_t>(), n, dev.scratch, s);
    hs.resize(n);
    HIP_CHECK(hipMemcpyAsync(hs.data(), E.cand_hash[cur].ptr, n * 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (uint64_t h : hs) mh.add_hash(h);
  }
}

}  // namespace

void KmerMinHash::add_many_bulk(const uint64_t* hashes, size_t n) {
  Device& dev = Device::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  Engine& E = Engine::get();
  hipStream_t s = dev.stream();
  materialize();
  E.seqbuf.ensure(n * 8);
  HIP_CHECK(hipMemcpyAsync(E.seqbuf.ptr, hashes, n * 8, hipMemcpyHostToDevice, s));
  RawHashSource src;
  src.hashes = E.seqbuf.as<uint64_t>();
  src.n = n;
  ingest(*this, src, s);
}

// merge (reference src/lib.rs:307-403) of two large, well-formed sketches on the device:
// concatenate, radix sort with the abundances as payload, collapse runs (sum), truncate to num.
// Used only when the host two-pointer loop would be the slow part; small or odd-shaped sketches
// (quirk Q5 states) stay on the statement-faithful host path.
bool KmerMinHash::merge_on_device(const KmerMinHash& other) {
  const size_t na = mins.size(), nb = other.mins.size(), n = na + nb;
  const bool both_tracked = has_abunds && other.has_abunds && abunds.size() == na && other.abunds.size() == nb;
  const bool none_tracked = !has_abunds && !other.has_abunds;
  if (n < (1u << 16) || n >= (1ull << 31) || !(both_tracked || none_tracked) || !Device::available()) return false;
  Device& dev = Device::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  Engine& E = Engine::get();
  hipStream_t s = dev.stream();
  E.cand_hash[0].ensure(n * 8); E.cand_hash[1].ensure(n * 8);
  HIP_CHECK(hipMemcpyAsync(E.cand_hash[0].ptr, mins.data(), na * 8, hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMemcpyAsync(E.cand_hash[0].as<uint64_t>() + na, other.mins.data(), nb * 8, hipMemcpyHostToDevice, s));
  if (both_tracked) {
    E.cand_pos[0].ensure(n * 8); E.cand_pos[1].ensure(n * 8);
    HIP_CHECK(hipMemcpyAsync(E.cand_pos[0].ptr, abunds.data(), na * 8, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(E.cand_pos[0].as<uint64_t>() + na, other.abunds.data(), nb * 8, hipMemcpyHostToDevice, s));
  }
  int cur = radix_sort_u64(E.cand_hash[0].as<uint64_t>(), E.cand_hash[1].as<uint64_t>(),
                           both_tracked ? E.cand_pos[0].as<uint64_t>() : nullptr,
                           both_tracked ? E.cand_pos[1].as<uint64_t>() : nullptr, n, dev.scratch, s);
  E.uniq.ensure(n * 8);
  E.starts.ensure((n + 1) * 4);
  const uint32_t nruns = run_length_encode_u64(E.cand_hash[cur].as<uint64_t>(), n, E.uniq.as<uint64_t>(),
                                               E.starts.as<uint32_t>(), dev.scratch, s);
  std::vector<uint64_t> nm(nruns), na_sum;
  HIP_CHECK(hipMemcpyAsync(nm.data(), E.uniq.ptr, (size_t)nruns * 8, hipMemcpyDeviceToHost, s));
  if (both_tracked) {
    E.red_b.ensure((size_t)nruns * 8);
    run_reduce(E.starts.as<uint32_t>(), nruns, nruns, (uint32_t)n, E.cand_pos[cur].as<uint64_t>(), nullptr,
               E.red_b.as<uint64_t>(), nullptr, s);
    na_sum.resize(nruns);
    HIP_CHECK(hipMemcpyAsync(na_sum.data(), E.red_b.ptr, (size_t)nruns * 8, hipMemcpyDeviceToHost, s));
  }
  HIP_CHECK(hipStreamSynchronize(s));
  if (!(nm.size() < (size_t)num || num == 0)) nm.resize(num);  // abundances are NOT truncated (Q5/Q6)
  mins.w().swap(nm);
  abunds.swap(na_sum);   // untracked inputs: Some(vec![]) like the reference
  has_abunds = true;
  return true;
}

namespace {

// String::from_utf8(kmer).unwrap() of reference src/lib.rs:270
bool utf8_valid(const uint8_t* s, size_t n) {
  size_t i = 0;
  while (i < n) {
    uint8_t c = s[i];
    if (c < 0x80) { i++; continue; }
    size_t need; uint8_t lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) need = 1;
    else if (c == 0xE0) { need = 2; lo = 0xA0; }
    else if (c >= 0xE1 && c <= 0xEC) need = 2;
    else if (c == 0xED) { need = 2; hi = 0x9F; }
    else if (c >= 0xEE && c <= 0xEF) need = 2;
    else if (c == 0xF0) { need = 3; lo = 0x90; }
    else if (c >= 0xF1 && c <= 0xF3) need = 3;
    else if (c == 0xF4) { need = 3; hi = 0x8F; }
    else return false;
    if (i + need >= n) return false;
    if (s[i + 1] < lo || s[i + 1] > hi) return false;
    for (size_t j = 2; j <= need; j++)
      if (s[i + j] < 0x80 || s[i + j] > 0xBF) return false;
    i += need + 1;
  }
  return true;
}

}  // namespace

namespace {
// force == false: where each record stops being valid DNA.  The windows before that byte are
// added, the first window holding it is the error (reference src/lib.rs:261-273, quirk Q1);
// the first such record in order is reported.  Leaves the cut points in b.vends / b.vend0.
void dna_validate(SeqBatch& b, const uint8_t* d_seq, const uint64_t* h_offsets, uint32_t nrec, uint32_t ksize, Engine& E,
                  hipStream_t s, bool* have_error, Error* err) {
  // vends = the record ends (a device copy of starts + 1: no host vector, a batch of reads has tens of millions of
  // records), lowered by k_first_invalid; the first offending record is found on the device too
  E.vendbuf.ensure((size_t)nrec * 8);
  if (nrec > 1) HIP_CHECK(hipMemcpyAsync(E.vendbuf.ptr, b.starts + 1, (size_t)nrec * 8, hipMemcpyDeviceToDevice, s));
  else HIP_CHECK(hipMemcpyAsync(E.vendbuf.ptr, h_offsets + 1, 8, hipMemcpyHostToDevice, s));
  launch_first_invalid(b, E.vendbuf.as<uint64_t>(), s);
  uint64_t first = ~0ull, vend0 = 0;
  if (nrec > 1) {
    E.misc.ensure(16);
    launch_first_bad_record(b.starts, E.vendbuf.as<uint64_t>(), nrec, ksize, E.misc.as<uint64_t>(), s);
    HIP_CHECK(hipMemcpyAsync(&first, E.misc.ptr, 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  } else {
    HIP_CHECK(hipMemcpyAsync(&vend0, E.vendbuf.ptr, 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (vend0 < h_offsets[1] && h_offsets[1] - h_offsets[0] >= ksize) first = 0;
  }
  if (first != ~0ull && !*have_error) {
    const uint32_t r = (uint32_t)first;
    const uint64_t st = h_offsets[r];
    uint64_t bad = vend0;
    if (nrec > 1) {
      HIP_CHECK(hipMemcpyAsync(&bad, E.vendbuf.as<uint64_t>() + r, 8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
    }
    const uint64_t ws = bad + 1 >= st + ksize ? bad + 1 - ksize : st;  // first window holding `bad`
    std::vector<uint8_t> kmer(ksize);
    HIP_CHECK(hipMemcpyAsync(kmer.data(), d_seq + ws, ksize, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (auto& c : kmer) if (c >= 'a' && c <= 'z') c -= 32;
    *have_error = true;
    if (!utf8_valid(kmer.data(), kmer.size()))
      *err = Error(kPanic, "sourmash panicked: called `Result::unwrap()` on an `Err` value: FromUtf8Error");
    else
      *err = Error(kInvalidDNA, "invalid DNA character in input k-mer: " + std::string(kmer.begin(), kmer.end()));
  }
  // records shorter than ksize must not lose anything: they add nothing either way
  if (nrec > 1) b.vends = E.vendbuf.as<uint64_t>();
  else b.vend0 = vend0;
}
// Protein arm set-up (reference src/lib.rs:277-301): the segment table (6 frames per record) of the
// six-frame layout that defines the arm's position space.  Returns false when there is nothing to hash.
bool prepare_protein(const SeqBatch& b, const uint64_t* h_offsets, uint32_t nrec, uint32_t ksize, uint64_t seed, Engine& E,
                     Device& dev, hipStream_t s, ProteinSource* src, bool* have_error, Error* err,
                     const uint64_t* known_total = nullptr) {
  (void)s;
  const uint32_t aa_k = ksize / 3;
  // residues over the six frames of every record: frame f of a record of `len` bases holds (len - f) / 3, twice
  // (forward and reverse complement); a record shorter than ksize adds nothing
  // (len / 3 + (len - 1) / 3 + (len - 2) / 3 == len - 2 for len >= 2)
  uint64_t total = 0;
  if (known_total) total = *known_total;
  else
    for (uint32_t r = 0; r < nrec; r++) {
      const uint64_t len = h_offsets[r + 1] - h_offsets[r];
      if (len < ksize) continue;
      if (len >= 2) total += 2 * (len - 2);
      else for (uint32_t frame = 0; frame < 3; frame++) total += len >= frame ? 2 * ((len - frame) / 3) : 0;
    }
  if (aa_k == 0) throw_panic("window size must be non-zero");  // aa.windows(0), quirk Q8
  if (total == 0) return false;
  src->b = b;
  if (nrec == 1) src->b.vend0 = (h_offsets[1] - h_offsets[0]) >= ksize ? b.len : 0;
  src->seg.clear(); src->seg_off = nullptr; src->h_offsets = h_offsets; src->nseg = 6 * nrec;
  src->total = total; src->win = aa_k; src->ksize = ksize; src->seed = seed; src->dev = &dev; src->eng = &E;
  src->have_error = have_error; src->err = err; src->translated = false;
  return true;
}

void ProteinSource::ensure_segments(hipStream_t s) {
  if (seg_off) return;
  const uint32_t nrec = nseg / 6;
  seg.assign((size_t)nseg + 1, 0);
  for (uint32_t r = 0; r < nrec; r++) {
    const uint64_t len = h_offsets[r + 1] - h_offsets[r];
    for (uint32_t f = 0; f < 6; f++) {
      const uint32_t frame = f >> 1;
      const uint64_t nres = (len >= ksize && len >= frame) ? (len - frame) / 3 : 0;
      seg[6 * (size_t)r + f + 1] = seg[6 * (size_t)r + f] + nres;
    }
  }
  eng->segbuf.ensure((size_t)(nseg + 1) * 8);
  HIP_CHECK(hipMemcpyAsync(eng->segbuf.ptr, seg.data(), (size_t)(nseg + 1) * 8, hipMemcpyHostToDevice, s));
  seg_off = eng->segbuf.as<uint64_t>();
}

}  // namespace

// Six-frame translation into E.resbuf, and the reference's UTF-8 panic: a codon chunk that is not
// UTF-8 makes from_utf8().unwrap() panic in that frame -- frames before it were added, it and the
// rest of the record were not.
void ProteinSource::translate(hipStream_t s) {
  Engine& E = *eng;
  E.badbuf.ensure((size_t)nseg * 4);
  E.resbuf.ensure(total + 64);  // k_hash_windows reads whole 8-byte words past the last start
  HIP_CHECK(hipMemsetAsync(E.badbuf.ptr, 0, (size_t)nseg * 4, s));
  SeqBatch tb = b;
  tb.vend0 = b.len;
  dev->prof_begin(s);
  launch_translate(tb, seg_off, nseg, total, ksize, E.resbuf.as<uint8_t>(), E.badbuf.as<uint32_t>(), s);
  dev->prof_end("translate", s);
  std::vector<uint32_t> bad(nseg);
  HIP_CHECK(hipMemcpyAsync(bad.data(), E.badbuf.ptr, (size_t)nseg * 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  const uint32_t nrec = nseg / 6;
  for (uint32_t r = 0; r < nrec; r++) {
    for (uint32_t f = 0; f < 6; f++) {
      if (!bad[6 * r + f]) continue;
      const uint64_t lo = seg[6 * r + f], hi = seg[6 * r + 6];
      if (hi > lo) HIP_CHECK(hipMemsetAsync(E.resbuf.as<uint8_t>() + lo, 0xFF, hi - lo, s));
      if (!*have_error) {
        *have_error = true;
        *err = Error(kPanic, "sourmash panicked: called `Result::unwrap()` on an `Err` value: Utf8Error");
      }
      break;
    }
  }
  translated = true;
}

// ------------------------------------------------------------------------------------
// add_sequence front ends

void KmerMinHash::add_sequences_device(const uint8_t* d_seq, uint64_t total_len, const uint64_t* h_offsets,
                                       uint32_t nrec, bool force, hipStream_t stream, Error* first_error) {
  if (nrec == 0 || total_len == 0) return;
  Device& dev = Device::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  Engine& E = Engine::get();
  hipStream_t s = stream ? stream : dev.stream();
  // drains queued small sequences first: stream order is part of the semantics.  A scaled sketch that lives in HBM stays
  // there (ingest() unites the batch with it on the device); any other state is handled on the host.
  flush_pending();
  if (!(num == 0 && max_hash > 0)) materialize();

  const uint64_t* d_starts = nullptr;
  if (nrec > 1) {
    E.offbuf.ensure((size_t)(nrec + 1) * 8);
    HIP_CHECK(hipMemcpyAsync(E.offbuf.ptr, h_offsets, (size_t)(nrec + 1) * 8, hipMemcpyHostToDevice, s));
    d_starts = E.offbuf.as<uint64_t>();
  }
  // records shorter than ksize add nothing (reference src/lib.rs:257); the protein arm also needs its position count.
  // A batch of reads is tens of millions of records: the device counts them from the offsets it has just received.
  uint64_t n_long = 0, protein_total = 0;
  if (nrec >= 65536) {
    E.misc.ensure(16);
    launch_record_stats(d_starts, nrec, ksize, E.misc.as<uint64_t>(), s);
    uint64_t st[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(st, E.misc.ptr, 16, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    n_long = st[0]; protein_total = st[1];
  } else {
    for (uint32_t r = 0; r < nrec; r++) {
      const uint64_t len = h_offsets[r + 1] - h_offsets[r];
      if (len < ksize) continue;
      n_long++;
      if (len >= 2) protein_total += 2 * (len - 2);    // len / 3 + (len - 1) / 3 + (len - 2) / 3 == len - 2
      else for (uint32_t frame = 0; frame < 3; frame++) protein_total += len >= frame ? 2 * ((len - frame) / 3) : 0;
    }
  }
  if (n_long == 0) return;
  SeqBatch b;
  b.seq = d_seq; b.len = total_len; b.starts = d_starts; b.nrec = nrec; b.vend0 = total_len;

  bool have_error = false;
  Error err(kNoError, "");

  if (!is_protein) {
    if (ksize == 0) throw_panic("window size must be non-zero");  // slice::windows(0)
    if (!force) dna_validate(b, d_seq, h_offsets, nrec, ksize, E, s, &have_error, &err);
    DnaSource src;
    src.b = b; src.ksize = ksize; src.seed = seed; src.dev = &dev;
    ingest(*this, src, s);
  } else {
    ProteinSource src;
    if (!prepare_protein(b, h_offsets, nrec, ksize, seed, E, dev, s, &src, &have_error, &err, &protein_total)) return;
    ingest(*this, src, s);
  }

  if (have_error) {
    if (first_error) *first_error = err;
    else throw err;
  }
}

void KmerMinHash::add_sequences_host(const uint8_t* h_seq, uint64_t total, const uint64_t* h_offsets, uint32_t nrec,
                                     bool force) {
  if (nrec == 0 || total == 0) return;
  Device& dev = Device::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  Engine& E = Engine::get();
  hipStream_t s = dev.stream();
  flush_pending();
  if (!(num == 0 && max_hash > 0)) materialize();
  E.seqbuf.ensure(total + 64);
  uint8_t* d_seq = E.seqbuf.as<uint8_t>();

  constexpr uint64_t kChunk = 128ull << 20;
  bool pipelined = force && !is_protein && ksize >= 1 && total >= 2 * kChunk && mode_of(*this) == kScaled;
  if (pipelined) {
    // the whole batch must be one hashing pass (as in ingest(): candidates under 2^30)
    long double frac = ((long double)max_hash + 1.0L) / 18446744073709551616.0L;
    pipelined = (long double)total * frac < (long double)(1ull << 30);
  }
  if (!pipelined) {
    HIP_CHECK(hipMemcpyAsync(d_seq, h_seq, total, hipMemcpyHostToDevice, s));
    add_sequences_device(d_seq, total, h_offsets, nrec, force, s, nullptr);
    return;
  }

  bool any_long = false;
  for (uint32_t r = 0; r < nrec; r++) any_long |= (h_offsets[r + 1] - h_offsets[r]) >= ksize;
  if (!any_long) return;
  SeqBatch b;
  b.seq = d_seq; b.len = total; b.nrec = nrec; b.vend0 = total;
  if (nrec > 1) {
    E.offbuf.ensure((size_t)(nrec + 1) * 8);
    HIP_CHECK(hipMemcpyAsync(E.offbuf.ptr, h_offsets, (size_t)(nrec + 1) * 8, hipMemcpyHostToDevice, s));
    b.starts = E.offbuf.as<uint64_t>();
  }
  DnaSource src;
  src.b = b; src.ksize = ksize; src.seed = seed; src.dev = &dev;
  const uint64_t P = total;
  uint64_t cap = estimate_capacity(P, max_hash);
  E.cand_hash[0].ensure(cap * 8); E.cand_hash[1].ensure(cap * 8);
  E.counter.ensure(8);
  HIP_CHECK(hipMemsetAsync(E.counter.ptr, 0, 8, s));
  CandSink sink;
  sink.hash = E.cand_hash[0].as<uint64_t>(); sink.pos = nullptr;
  sink.count = E.counter.as<unsigned long long>(); sink.capacity = cap;

  // chunk c is copied on the copy stream; the hashing stream waits for its event and hashes the
  // k-mer starts whose last base is already on the device: [done, copied - (k-1))
  const uint32_t nchunks = (uint32_t)((total + kChunk - 1) / kChunk);
  std::vector<hipEvent_t> ev(nchunks, nullptr);
  hipStream_t cs = dev.copy_stream();
  uint64_t done = 0;
  try {
    for (uint32_t c = 0; c < nchunks; c++) {
      const uint64_t lo = (uint64_t)c * kChunk, hi = std::min(total, lo + kChunk);
      HIP_CHECK(hipEventCreateWithFlags(&ev[c], hipEventDisableTiming));
      HIP_CHECK(hipMemcpyAsync(d_seq + lo, h_seq + lo, hi - lo, hipMemcpyHostToDevice, cs));
      HIP_CHECK(hipEventRecord(ev[c], cs));
      HIP_CHECK(hipStreamWaitEvent(s, ev[c], 0));
      const uint64_t upto = (c + 1 == nchunks) ? P : (hi >= ksize ? hi - (ksize - 1) : 0);
      if (upto > done) {
        src.launch(done, upto, max_hash, sink, s);
        done = upto;
      }
    }
    unsigned long long n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, E.counter.ptr, 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (auto& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (n > cap) n = E.run_chunk(&src, 0, P, max_hash, false, s);   // rare: hash again into an exact-size buffer
    Delta d;
    if ((mins.empty() || this->dev) && n > 0) {
      auto ds = std::make_shared<DeviceSketch>();
      E.reduce_chunk(n, 0, false, false, s, &d, ds.get(), max_hash);
      if (!this->dev) this->dev = ds;
      else E.union_into_device_sketch(*this, *ds, s);
    } else if (n > 0) {
      E.reduce_chunk(n, 0, false, false, s, &d, nullptr, max_hash);
      apply_scaled(*this, d);
    }
  } catch (...) {
    (void)hipStreamSynchronize(cs);
    (void)hipStreamSynchronize(s);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    throw;
  }
}

void add_sequences_grouped(KmerMinHash* const* mhs, uint32_t n_mh, const uint8_t* d_seq, uint64_t total_len,
                           const uint64_t* h_offsets, const uint32_t* grp, uint32_t nrec, bool force, hipStream_t stream,
                           Error* first_error) {
  if (nrec == 0 || total_len == 0) return;
  Device& dev = Device::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  Engine& E = Engine::get();
  hipStream_t s = stream ? stream : dev.stream();
  for (uint32_t r = 0; r < nrec; r++)
    if (grp[r] >= n_mh) throw_internal("record group out of range");
  bool have_error = false;
  Error err(kNoError, "");

  // every maximal run of consecutive records of one group
  struct RecRun { uint32_t r0, r1; };
  std::vector<RecRun> runs;
  for (uint32_t r0 = 0; r0 < nrec;) {
    uint32_t r1 = r0 + 1;
    while (r1 < nrec && grp[r1] == grp[r0]) r1++;
    runs.push_back({r0, r1});
    r0 = r1;
  }
  // sketch by sketch: each run is one batch for its sketch (any parameters, any mode)
  std::vector<uint64_t> offs;
  auto serve_run = [&](const RecRun& rr) {
    offs.assign(h_offsets + rr.r0, h_offsets + rr.r1 + 1);
    const uint64_t base = offs[0];
    for (auto& o : offs) o -= base;
    Error e(kNoError, "");
    mhs[grp[rr.r0]]->add_sequences_device(d_seq + base, offs.back(), offs.data(), rr.r1 - rr.r0, force, s, &e);
    if (e.code != kNoError && !have_error) { have_error = true; err = e; }
  };

  // Shared-launch paths: sketches of one molecule type with equal (ksize, seed), all scaled with one max_hash, or
  // all bottom-num without abundance tracking (per-record thresholds).
  enum { kSlow, kSharedScaled, kSharedNum } path = kSlow;
  const KmerMinHash& m0 = *mhs[0];
  if (m0.ksize > 0 && (!m0.is_protein || m0.ksize >= 3) && runs.size() > 1) {
    bool same = true, all_scaled = true, all_num = !m0.is_protein;   // per-record thresholds exist in the DNA kernel only
    for (uint32_t g = 0; g < n_mh && same; g++) {
      mhs[g]->materialize();
      const KmerMinHash& m = *mhs[g];
      same = m.is_protein == m0.is_protein && m.ksize == m0.ksize && m.seed == m0.seed;
      const int mode = mode_of(m);
      all_scaled &= mode == kScaled && m.max_hash == m0.max_hash;
      all_num &= mode == kNum;
    }
    if (same) same = n_mh < (1u << 24);   // the group tag shares a word with the position (24 + 40 bits)
    if (same) {
      // one sketch listed twice is served in order instead
      std::vector<const KmerMinHash*> sorted(mhs, mhs + n_mh);
      std::sort(sorted.begin(), sorted.end());
      same = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
    }
    if (same && all_scaled) path = kSharedScaled;
    else if (same && all_num && total_len < (1ull << 40)) path = kSharedNum;   // positions share a word with the group tag
  }
  if (path == kSlow) {
    for (const RecRun& rr : runs) serve_run(rr);
    if (have_error) { if (first_error) *first_error = err; else throw err; }
    return;
  }

  const uint32_t ksize = m0.ksize;
  bool any_long = false;
  for (uint32_t r = 0; r < nrec; r++) any_long |= (h_offsets[r + 1] - h_offsets[r]) >= ksize;
  if (!any_long) return;
  E.offbuf.ensure((size_t)(nrec + 1) * 8);
  E.grpbuf.ensure((size_t)nrec * 4);
  HIP_CHECK(hipMemcpyAsync(E.offbuf.ptr, h_offsets, (size_t)(nrec + 1) * 8, hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMemcpyAsync(E.grpbuf.ptr, grp, (size_t)nrec * 4, hipMemcpyHostToDevice, s));
  SeqBatch b;
  b.seq = d_seq; b.len = total_len; b.starts = E.offbuf.as<uint64_t>(); b.nrec = nrec; b.vend0 = total_len;
  DnaSource dna;
  ProteinSource prot;
  HashSource* srcp = &dna;
  // candidate positions -> groups: DNA positions are bases (table = record starts), protein positions
  // are residues of the six-frame buffer (table = segment starts, six segments per record)
  const uint64_t* pos_table = E.offbuf.as<uint64_t>();
  uint32_t pos_entries = nrec;
  const uint32_t* pos_groups = E.grpbuf.as<uint32_t>();
  if (!m0.is_protein) {
    if (!force) dna_validate(b, d_seq, h_offsets, nrec, ksize, E, s, &have_error, &err);
    dna.b = b; dna.ksize = ksize; dna.seed = m0.seed; dna.dev = &dev;
  } else {
    if (!prepare_protein(b, h_offsets, nrec, ksize, m0.seed, E, dev, s, &prot, &have_error, &err)) return;
    srcp = &prot;
    std::vector<uint32_t> g6((size_t)nrec * 6);
    for (uint32_t r = 0; r < nrec; r++) for (int f = 0; f < 6; f++) g6[(size_t)6 * r + f] = grp[r];
    E.grpbuf.ensure(g6.size() * 4);
    HIP_CHECK(hipMemcpyAsync(E.grpbuf.ptr, g6.data(), g6.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));   // g6 is a stack-lifetime staging vector
    prot.ensure_segments(s);              // the grouped fold maps positions to groups through the segment table
    pos_table = E.segbuf.as<uint64_t>(); pos_entries = 6 * nrec; pos_groups = E.grpbuf.as<uint32_t>();
  }
  HashSource& src = *srcp;
  const uint64_t P = src.positions();
  bool any_track = false;
  for (uint32_t g = 0; g < n_mh; g++) any_track |= mhs[g]->has_abunds;

  // n candidates (hash, position) in cand_hash[0] / cand_pos[0] -> per group: its distinct hashes
  // ascending (+ run starts when some sketch tracks abundance), handed to per_group(g, hashes,
  // count, starts_or_null, end_of_last_run)
  // keep_pos: the payload becomes (group << 40 | position) so that the stream positions survive
  // the two sorts (quirk Q3 of abundance-tracking bottom-num sketches needs them); the second sort
  // then runs over the group bytes only
  constexpr int kPosBits = 40;
  int fold_cur = 0;   // which ping-pong half holds the sorted candidates after a fold
  auto fold_groups = [&](uint64_t n, bool keep_pos, auto&& per_group) {
    if (n == 0) return;
    // (hash, position) -> (hash, group); sort by hash, then stably by group: (group, hash) order
    launch_pos_to_group(E.cand_pos[0].as<uint64_t>(), n, pos_table, pos_entries, pos_groups, s, keep_pos ? kPosBits : 0);
    const int c1 = radix_sort_u64(E.cand_hash[0].as<uint64_t>(), E.cand_hash[1].as<uint64_t>(), E.cand_pos[0].as<uint64_t>(),
                                  E.cand_pos[1].as<uint64_t>(), n, dev.scratch, s);
    const int c2 = radix_sort_u64(E.cand_pos[c1].as<uint64_t>(), E.cand_pos[c1 ^ 1].as<uint64_t>(),
                                  E.cand_hash[c1].as<uint64_t>(), E.cand_hash[c1 ^ 1].as<uint64_t>(), n, dev.scratch, s,
                                  keep_pos ? kPosBits / 8 : 0, 8);
    const int cur = c1 ^ c2;
    fold_cur = cur;
    E.uniq.ensure(n * 8); E.uniq2.ensure(n * 8); E.starts.ensure((n + 1) * 4);
    const uint32_t nruns = run_length_encode_u64(E.cand_hash[cur].as<uint64_t>(), n, E.uniq.as<uint64_t>(),
                                                 E.starts.as<uint32_t>(), dev.scratch, s, nullptr, nullptr,
                                                 E.cand_pos[cur].as<uint64_t>(), E.uniq2.as<uint64_t>(), keep_pos ? kPosBits : 0);
    if (nruns == 0) return;
    const uint64_t* h_minpos = nullptr;
    if (keep_pos) {
      // first stream position of every run (the group tag is the same within a run: min keeps it)
      E.cmp_out.ensure((size_t)nruns * 8);
      run_reduce(E.starts.as<uint32_t>(), nruns, nruns, (uint32_t)n, nullptr, E.cand_pos[cur].as<uint64_t>(), nullptr,
                 E.cmp_out.as<uint64_t>(), s);
      E.pin_pair.ensure((size_t)nruns * 8);
      HIP_CHECK(hipMemcpyAsync(E.pin_pair.ptr, E.cmp_out.ptr, (size_t)nruns * 8, hipMemcpyDeviceToHost, s));
      h_minpos = E.pin_pair.as<uint64_t>();
    }
    // group boundaries in run space: collapse the per-run group ids once more
    E.red_b.ensure((size_t)nruns * 8); E.misc.ensure(((size_t)nruns + 1) * 4);
    const uint32_t ngr = run_length_encode_u64(E.uniq2.as<uint64_t>(), nruns, E.red_b.as<uint64_t>(), E.misc.as<uint32_t>(),
                                               dev.scratch, s);
    E.pin_a.ensure((size_t)nruns * 8);
    E.pin_b.ensure((size_t)nruns * 4 + (size_t)ngr * 12 + 16);
    uint64_t* h_uniq = E.pin_a.as<uint64_t>();
    uint32_t* h_starts = E.pin_b.as<uint32_t>();                       // [nruns]   (only when some sketch tracks)
    uint32_t* h_gstart = h_starts + nruns;                             // [ngr]
    uint64_t* h_gid = reinterpret_cast<uint64_t*>(E.pin_b.as<char>() + (((size_t)nruns + ngr) * 4 + 7) / 8 * 8);  // [ngr]
    HIP_CHECK(hipMemcpyAsync(h_uniq, E.uniq.ptr, (size_t)nruns * 8, hipMemcpyDeviceToHost, s));
    if (any_track) HIP_CHECK(hipMemcpyAsync(h_starts, E.starts.ptr, (size_t)nruns * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(h_gstart, E.misc.ptr, (size_t)ngr * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(h_gid, E.red_b.ptr, (size_t)ngr * 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (uint32_t q = 0; q < ngr; q++) {
      const size_t a = h_gstart[q], e = q + 1 < ngr ? h_gstart[q + 1] : nruns;
      per_group((uint32_t)h_gid[q], h_uniq + a, e - a, any_track ? h_starts + a : nullptr,
                e < nruns && any_track ? h_starts[e] : (uint32_t)n, h_minpos ? h_minpos + a : nullptr);
    }
  };

  Delta d;
  if (path == kSharedScaled) {
    long double frac = ((long double)m0.max_hash + 1.0L) / 18446744073709551616.0L;
    long double span_ld = (long double)(1ull << 30) / frac;
    const uint64_t CH = span_ld > 4.0e12L ? (uint64_t)4e12 : (span_ld < 16777216.0L ? (1ull << 24) : (uint64_t)span_ld);
    for (uint64_t lo = 0; lo < P; lo += CH) {
      const uint64_t hi = std::min(P, lo + CH);
      const uint64_t n = E.run_chunk(&src, lo, hi, m0.max_hash, true, s);
      fold_groups(n, false, [&](uint32_t g, const uint64_t* hashes, size_t cnt, const uint32_t* starts, uint32_t end,
                                const uint64_t*) {
        KmerMinHash& mh = *mhs[g];
        d.uniq.assign(hashes, hashes + cnt);
        if (mh.has_abunds) {
          d.run_start.assign(starts, starts + cnt);
          d.run_start.push_back(end);
        }
        apply_scaled(mh, d);
      });
    }
  } else {
    // Bottom-num sketches: group g keeps hashes <= thr[g], chosen so that about 2 num + 64 of its
    // windows pass (a full sketch's own maximum when that is lower).  A group that still shows
    // fewer than num distinct hashes under a finite threshold (a repetitive genome) is served
    // again on its own afterwards; nothing was applied to it here.
    std::vector<uint64_t> windows(n_mh, 0), thr(n_mh, UINT64_MAX);
    for (uint32_t r = 0; r < nrec; r++) {
      const uint64_t len = h_offsets[r + 1] - h_offsets[r];
      if (len >= ksize) windows[grp[r]] += len - ksize + 1;
    }
    long double expect = 0;
    for (uint32_t g = 0; g < n_mh; g++) {
      const KmerMinHash& m = *mhs[g];
      const long double want = 2.0L * m.num + 64.0L;
      if (windows[g] > 0 && want < 0.5L * windows[g]) thr[g] = (uint64_t)(want / windows[g] * 18446744073709551616.0L);
      if (m.mins.size() >= (size_t)m.num && m.mins.back() < thr[g]) thr[g] = m.mins.back();
      expect += (long double)windows[g] * (((long double)thr[g] + 1.0L) / 18446744073709551616.0L);
    }
    uint64_t cap = (uint64_t)(expect * 1.25L) + 65536;
    // one launch over the whole batch: the kernel looks the threshold up per record
    std::vector<uint64_t> thr_of_rec(nrec);
    for (uint32_t r = 0; r < nrec; r++) thr_of_rec[r] = thr[grp[r]];
    E.vendbuf2.ensure((size_t)nrec * 8);
    HIP_CHECK(hipMemcpyAsync(E.vendbuf2.ptr, thr_of_rec.data(), (size_t)nrec * 8, hipMemcpyHostToDevice, s));
    dna.thr_rec = E.vendbuf2.as<uint64_t>();
    const long double favg = expect / (long double)(P ? P : 1);
    const uint64_t thr_avg = favg >= 1.0L ? UINT64_MAX : (uint64_t)(favg * 18446744073709551616.0L);   // sizes the LDS stage only
    uint64_t n = 0;
    for (int attempt = 0;; attempt++) {
      if (cap >= (1ull << 31)) throw_internal("candidate set of one grouped batch exceeds 2^31 entries");
      E.cand_hash[0].ensure(cap * 8); E.cand_hash[1].ensure(cap * 8);
      E.cand_pos[0].ensure(cap * 8); E.cand_pos[1].ensure(cap * 8);
      E.counter.ensure(8);
      HIP_CHECK(hipMemsetAsync(E.counter.ptr, 0, 8, s));
      CandSink sink;
      sink.hash = E.cand_hash[0].as<uint64_t>(); sink.pos = E.cand_pos[0].as<uint64_t>();
      sink.count = E.counter.as<unsigned long long>(); sink.capacity = cap;
      src.launch(0, P, thr_avg, sink, s);
      unsigned long long got = 0;
      HIP_CHECK(hipMemcpyAsync(&got, E.counter.ptr, 8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      n = got;
      if (n <= cap) break;
      if (attempt) throw_internal("candidate buffer overflow after re-run");
      cap = n;   // the counter kept counting: exact size for the re-run
    }
    std::vector<uint8_t> resolved(n_mh, 0);
    fold_groups(n, any_track, [&](uint32_t g, const uint64_t* hashes, size_t cnt, const uint32_t* starts, uint32_t end,
                                  const uint64_t* minpos) {
      KmerMinHash& mh = *mhs[g];
      if (cnt < (size_t)mh.num && thr[g] != UINT64_MAX && !(mh.mins.size() >= (size_t)mh.num && thr[g] == mh.mins.back()))
        return;
      resolved[g] = 1;
      const size_t kept = std::min(cnt, (size_t)mh.num);
      d.uniq.assign(hashes, hashes + kept);
      if (mh.has_abunds) {
        // run k of the group is [starts[k], starts[k+1]) in the sorted candidate arrays
        d.run_start.assign(starts, starts + kept);
        d.run_start.push_back(kept < cnt ? starts[kept] : end);
        d.minpos.resize(kept);
        for (size_t k = 0; k < kept; k++) d.minpos[k] = minpos[k] & ((1ull << kPosBits) - 1);
        d.sorted_buf = fold_cur;
        d.pos_mask = (1ull << kPosBits) - 1;
      }
      apply_num(mh, d, E, s);
    });
    // groups with no candidate at all under an exhaustive threshold have nothing to add
    for (uint32_t g = 0; g < n_mh; g++)
      if (thr[g] == UINT64_MAX || (mhs[g]->mins.size() >= (size_t)mhs[g]->num && thr[g] == mhs[g]->mins.back())) resolved[g] = 1;
    // (validation above already holds the batch's first error; serve_run keeps the earliest)
    for (const RecRun& rr : runs)
      if (!resolved[grp[rr.r0]] && windows[grp[rr.r0]] > 0) serve_run(rr);
  }
  if (have_error) {
    if (first_error) *first_error = err;
    else throw err;
  }
}

namespace {
constexpr size_t kLazyMaxRecord = 1u << 20;   // calls at least this long go straight to the device
constexpr size_t kLazyFlushBytes = 8u << 20;  // queued bytes that trigger a batch
}  // namespace

void KmerMinHash::flush_pending() const {
  if (pend_woff.size() > 1) flush_words();
  if (pend_off.size() <= 1) { pend_seq.clear(); pend_off.clear(); return; }
  // move the queue out first: add_sequences_device() calls back into materialize()
  std::vector<uint8_t> seqs;
  std::vector<uint64_t> offs;
  seqs.swap(pend_seq);
  offs.swap(pend_off);
  Device& dev = Device::get();
  Engine& E = Engine::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  E.seqbuf.ensure(seqs.size() + 64);
  HIP_CHECK(hipMemcpyAsync(E.seqbuf.ptr, seqs.data(), seqs.size(), hipMemcpyHostToDevice, dev.stream()));
  // queued DNA records were cut at their first invalid byte when force was false, so force=true
  // reproduces both settings; protein ignores force (reference src/lib.rs:275-302)
  const_cast<KmerMinHash*>(this)->add_sequences_device(E.seqbuf.as<uint8_t>(), seqs.size(), offs.data(),
                                                       (uint32_t)(offs.size() - 1), true, dev.stream(), nullptr);
}

void KmerMinHash::add_sequence(const uint8_t* seq, size_t len, bool force) {
  if (len < ksize) return;  // reference src/lib.rs:257
  if (!is_protein && ksize == 0) throw_panic("window size must be non-zero");
  if (is_protein && ksize / 3 == 0) throw_panic("window size must be non-zero");
  Device& dev = Device::get();   // raises here, not at the deferred batch, when there is no GPU
  if (pend_woff.size() > 1) flush_words();   // queued words came first
  bool direct = len >= kLazyMaxRecord || pend_off.size() >= (1u << 20);
  size_t use = len;
  bool have_err = false;
  Error err(kNoError, "");
  if (!direct && !is_protein && !force) {
    // force=false: the call itself must report the first window holding a non-ACGT byte (Q1); the
    // windows before it are still added, i.e. the record is cut at that byte
    for (size_t i = 0; i < len; i++) {
      const uint8_t u = seq[i] & 0xDFu;
      if (!(u == 'A' || u == 'C' || u == 'G' || u == 'T')) {
        const size_t ws = i + 1 >= (size_t)ksize ? i + 1 - ksize : 0;
        std::vector<uint8_t> kmer(seq + ws, seq + ws + ksize);
        for (auto& c : kmer) if (c >= 'a' && c <= 'z') c -= 32;
        have_err = true;
        if (!utf8_valid(kmer.data(), kmer.size()))
          err = Error(kPanic, "sourmash panicked: called `Result::unwrap()` on an `Err` value: FromUtf8Error");
        else
          err = Error(kInvalidDNA, "invalid DNA character in input k-mer: " + std::string(kmer.begin(), kmer.end()));
        use = i;
        break;
      }
    }
  }
  if (!direct && is_protein) {
    // a codon chunk that is not UTF-8 panics in the reference: take the synchronous path, which
    // reproduces which frames were added before the panic
    for (size_t i = 0; i < len; i++) if (seq[i] & 0x80u) { direct = true; break; }
  }
  if (direct) {
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    flush_pending();
    const uint64_t off[2] = {0, (uint64_t)len};
    add_sequences_host(seq, len, off, 1, force);
    return;
  }
  if (use >= ksize) {
    if (pend_off.empty()) pend_off.push_back(0);
    pend_seq.insert(pend_seq.end(), seq, seq + use);
    pend_off.push_back(pend_seq.size());
    if (pend_seq.size() >= kLazyFlushBytes) flush_pending();
  }
  if (have_err) throw err;
}

// ------------------------------------------------------------------------------------
// murmur64 of whole words (add_word, hash_murmur)

void Engine::hash_words(const uint8_t* bytes, const uint64_t* offsets, uint32_t n, uint64_t seed,
                        uint64_t* out) {
  if (n == 0) return;
  Device& dev = Device::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  hipStream_t s = dev.stream();
  const uint64_t total = offsets[n];
  misc.ensure(total + 64 + (size_t)(n + 1) * 8 + (size_t)n * 8);
  uint8_t* d_bytes = misc.as<uint8_t>();
  const size_t off_at = (total + 63) & ~(size_t)63;
  uint64_t* d_off = reinterpret_cast<uint64_t*>(d_bytes + off_at);
  uint64_t* d_out = d_off + (n + 1);
  if (total) HIP_CHECK(hipMemcpyAsync(d_bytes, bytes, total, hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMemcpyAsync(d_off, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
  launch_hash_segments(d_bytes, d_off, n, seed, d_out, s);
  HIP_CHECK(hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
}

// reference src/lib.rs:247-250: add_hash(_hash_murmur(word, seed)).  The hash is computed on the device; a call per word
// would be a launch and a synchronisation per word (the reference's own protein arm calls add_word per window), so the
// words are queued and hashed together (flush_words).
void KmerMinHash::add_word(const uint8_t* w, size_t len) {
  (void)Device::get();             // raises here, not at the deferred batch, when there is no GPU
  if (pend_off.size() > 1) flush_pending();   // queued sequences came first
  if (pend_woff.empty()) pend_woff.push_back(0);
  pend_words.insert(pend_words.end(), w, w + len);
  pend_woff.push_back(pend_words.size());
  if (pend_woff.size() > (1u << 16) || pend_words.size() > (8u << 20)) flush_words();
}

void KmerMinHash::flush_words() const {
  if (pend_woff.size() <= 1) { pend_words.clear(); pend_woff.clear(); return; }
  std::vector<uint8_t> bytes;
  std::vector<uint64_t> offs;
  bytes.swap(pend_words);          // moved out first: add_many() below observes the sketch (and would come back here)
  offs.swap(pend_woff);
  const uint32_t n = (uint32_t)(offs.size() - 1);
  std::vector<uint64_t> hashes(n);
  const uint8_t dummy = 0;
  Engine::get().hash_words(bytes.empty() ? &dummy : bytes.data(), offs.data(), n, seed, hashes.data());
  const_cast<KmerMinHash*>(this)->add_many(hashes.data(), n);   // add_hash in call order (bulk: the device fold, same result)
}

// ------------------------------------------------------------------------------------
// comparisons of host-resident sketches

void Engine::release_workspace() {
  Device& dev = Device::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  HIP_CHECK(hipDeviceSynchronize());
  for (DeviceBuffer* b : {&cand_hash[0], &cand_hash[1], &cand_pos[0], &cand_pos[1], &counter, &uniq, &uniq2, &starts, &red_b,
                          &misc, &seqbuf, &offbuf, &vendbuf, &vendbuf2, &grpbuf, &resbuf, &segbuf, &badbuf, &cmp_a, &cmp_b,
                          &cmp_oa, &cmp_ob, &cmp_out, &pair_out, &dev.scratch})
    b->release();
  release_compare_scratch();
  device_pool_trim();
}

void Engine::pack_sketches(const std::vector<const KmerMinHash*>& v, DeviceBuffer& data, DeviceBuffer& offs,
                         SketchSet* out, uint32_t* maxlen, std::vector<uint64_t>* h_off, hipStream_t s) {
  std::vector<uint64_t> off(v.size() + 1, 0);
  *maxlen = 0;
  for (size_t i = 0; i < v.size(); i++) v[i]->materialize();
  for (size_t i = 0; i < v.size(); i++) {
    off[i + 1] = off[i] + v[i]->mins.size();
    *maxlen = std::max<uint32_t>(*maxlen, (uint32_t)v[i]->mins.size());
  }
  data.ensure(off.back() * 8 + 8);
  offs.ensure(off.size() * 8);
  // gather through two page-locked staging buffers: one H2D per ~32 MB instead of one pageable
  // copy per sketch (10^5 sketches: 330 -> 190 ms), filling one buffer while the other is in flight
  constexpr size_t kStage = 32u << 20;
  PinnedBuffer* pin[2] = {&pin_a, &pin_b};
  int which = 0;
  for (size_t i = 0; i < v.size();) {
    if (v[i]->mins.size() * 8 > kStage) {   // a sketch larger than the stage goes on its own
      HIP_CHECK(hipMemcpyAsync(data.as<uint64_t>() + off[i], v[i]->mins.data(), v[i]->mins.size() * 8,
                               hipMemcpyHostToDevice, s));
      i++;
      continue;
    }
    pin[which]->ensure(kStage);
    uint8_t* hp = pin[which]->as<uint8_t>();
    size_t used = 0;
    const uint64_t first = off[i];
    while (i < v.size() && used + v[i]->mins.size() * 8 <= kStage) {
      if (!v[i]->mins.empty()) std::memcpy(hp + used, v[i]->mins.data(), v[i]->mins.size() * 8);
      used += v[i]->mins.size() * 8;
      i++;
    }
    if (used) HIP_CHECK(hipMemcpyAsync(data.as<uint64_t>() + first, hp, used, hipMemcpyHostToDevice, s));
    which ^= 1;
    if (which == 0) HIP_CHECK(hipStreamSynchronize(s));   // both buffers may be refilled now
  }
  HIP_CHECK(hipMemcpyAsync(offs.ptr, off.data(), off.size() * 8, hipMemcpyHostToDevice, s));
  HIP_CHECK(hipStreamSynchronize(s));  // `off` is a stack-lifetime staging buffer
  out->hashes = data.as<uint64_t>();
  out->offsets = offs.as<uint64_t>();
  out->n = (uint32_t)v.size();
  if (h_off) h_off->swap(off);
}

void Engine::compare_host(const std::vector<const KmerMinHash*>& rows, const std::vector<const KmerMinHash*>& cols,
                          const uint32_t* row_nums_host, uint32_t num, uint64_t* common, uint64_t* size,
                          double* jaccard, uint64_t* count_common, double* containment) {
  Device& dev = Device::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  hipStream_t s = dev.stream();
  if (rows.size() == 1 && cols.size() == 1) {   // one pair: mirrored copies, no CSR upload
    PairResult r;
    compare_pair(*rows[0], *cols[0], row_nums_host ? row_nums_host[0] : num, &r);
    if (common) *common = r.common;
    if (size) *size = r.size;
    if (jaccard) *jaccard = r.jaccard;
    if (count_common) *count_common = r.count_common;
    if (containment) *containment = r.containment;
    return;
  }
  SketchSet R, C;
  uint32_t mr = 0, mc = 0;
  // the same list on both axes (all-vs-all): one upload, and the block compare may use symmetry
  const bool same_sets = rows.size() == cols.size() && std::equal(rows.begin(), rows.end(), cols.begin());
  std::vector<uint64_t> h_off_r, h_off_c;
  pack_sketches(rows, cmp_a, cmp_oa, &R, &mr, &h_off_r, s);
  R.h_offsets = h_off_r.data();
  if (same_sets) { C = R; mc = mr; }
  else { pack_sketches(cols, cmp_b, cmp_ob, &C, &mc, &h_off_c, s); C.h_offsets = h_off_c.data(); }
  uint64_t row_total = 0, col_total = 0;
  for (auto* m : rows) row_total += m->mins.size();
  for (auto* m : cols) col_total += m->mins.size();
  const size_t np = rows.size() * cols.size();
  if (np == 0) return;
  cmp_out.ensure(np * 8 * 5 + rows.size() * 4 + 64);
  uint64_t* d_common = cmp_out.as<uint64_t>();
  uint64_t* d_size = d_common + np;
  double* d_jac = reinterpret_cast<double*>(d_size + np);
  uint64_t* d_cc = reinterpret_cast<uint64_t*>(d_jac + np);
  double* d_cont = reinterpret_cast<double*>(d_cc + np);
  uint32_t* d_rownum = reinterpret_cast<uint32_t*>(d_cont + np);
  // one num for every row: pass it as the launch-wide value
  if (row_nums_host) {
    bool uniform = true;
    for (size_t i = 1; i < rows.size(); i++) uniform &= row_nums_host[i] == row_nums_host[0];
    if (uniform) { num = row_nums_host[0]; row_nums_host = nullptr; }
  }
  if (row_nums_host)
    HIP_CHECK(hipMemcpyAsync(d_rownum, row_nums_host, rows.size() * 4, hipMemcpyHostToDevice, s));
  CompareOut o;
  // only what the caller asked for: without count_common / containment the kernels may stop at the cut
  o.common = common ? d_common : nullptr; o.size = size ? d_size : nullptr; o.jaccard = jaccard ? d_jac : nullptr;
  o.count_common = count_common ? d_cc : nullptr; o.containment = containment ? d_cont : nullptr;
  launch_compare_block(R, C, num, row_nums_host ? d_rownum : nullptr, o, dev, s, mr, mc, row_total, col_total, same_sets);
  if (common) HIP_CHECK(hipMemcpyAsync(common, d_common, np * 8, hipMemcpyDeviceToHost, s));
  if (size) HIP_CHECK(hipMemcpyAsync(size, d_size, np * 8, hipMemcpyDeviceToHost, s));
  if (jaccard) HIP_CHECK(hipMemcpyAsync(jaccard, d_jac, np * 8, hipMemcpyDeviceToHost, s));
  if (count_common) HIP_CHECK(hipMemcpyAsync(count_common, d_cc, np * 8, hipMemcpyDeviceToHost, s));
  if (containment) HIP_CHECK(hipMemcpyAsync(containment, d_cont, np * 8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
}

// The sketch's hashes in device memory.  A sketch whose state lives in HBM is compared where it is (nothing is brought to
// the host); otherwise the device copy of mh.mins: reused while the vector's generation is the one the copy was made at
// (TrackedMins: every mutation route bumps it -- exact, O(1), nothing is read), re-created (never overwritten: a published
// mirror is immutable) otherwise.
const uint64_t* Engine::device_mins(const KmerMinHash& mh, size_t* n_out, hipStream_t s) {
  mh.flush_pending();
  if (mh.dev) {
    *n_out = (size_t)mh.dev->n;
    return mh.dev->uniq.as<uint64_t>();
  }
  const size_t n = mh.mins.size();
  if (!mh.mirror || mh.mirror->gen != mh.mins.generation()) {
    auto m = std::make_shared<DeviceMirror>();
    m->ptr = device_pool_alloc(n * 8 + 8, &m->cap);
    if (n) HIP_CHECK(hipMemcpyAsync(m->ptr, mh.mins.data(), n * 8, hipMemcpyHostToDevice, s));  // pageable source: staged before return
    m->n = n;
    m->gen = mh.mins.generation();
    mh.mirror = m;
  }
  *n_out = n;
  return reinterpret_cast<const uint64_t*>(mh.mirror->ptr);
}

void Engine::compare_pair(const KmerMinHash& a, const KmerMinHash& b, uint32_t num, PairResult* out) {
  Device& dev = Device::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  hipStream_t s = dev.stream();
  size_t na = 0, nb = 0;
  const uint64_t* A = device_mins(a, &na, s);
  const uint64_t* B = device_mins(b, &nb, s);
  if (na >= (1ull << 32) || nb >= (1ull << 32)) throw_internal("compare: a sketch of more than 2^32 hashes");
  const uint32_t la = (uint32_t)na, lb = (uint32_t)nb;
  pair_out.ensure(sizeof(PairOut));
  pin_pair.ensure(sizeof(PairOut));
  launch_compare_pair(A, la, B, lb, num, pair_out.as<PairOut>(), dev, s);
  PairOut* h = pin_pair.as<PairOut>();
  HIP_CHECK(hipMemcpyAsync(h, pair_out.ptr, sizeof(PairOut), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  const bool cut = num != 0 && h->tot_u > num;
  out->count_common = h->tot_c;
  out->size = cut ? num : h->tot_u;
  out->common = cut ? h->common : h->tot_c;
  out->jaccard = (double)out->common / (double)(out->size > 1 ? out->size : 1);
  out->containment = (double)h->tot_c / (double)la;
}

// KmerMinHash::add_hash over one more batch (reference src/lib.rs:192-245) for a scaled sketch whose state lives in HBM:
// the batch's sorted distinct hashes with their counts (run starts of the fold, or plain u64 counts).  Present hashes have
// their counts raised, new ones are inserted.
static void union_core(KmerMinHash& mh, const uint64_t* d_mins, const uint32_t* d_starts, const uint64_t* d_counts, uint64_t n_delta,
                       uint64_t delta_total, Engine& E, hipStream_t s) {
  Device& dev = Device::get();
  DeviceSketch& S = *mh.dev;
  if (n_delta == 0) return;
  if (S.n + n_delta >= (1ull << 31) || delta_total >= (1ull << 32)) throw_internal("sketch union: more than 2^31 hashes");
  const uint32_t n_s = (uint32_t)S.n, n_d = (uint32_t)n_delta;
  const bool track = mh.has_abunds;
  if (track && !S.has_counts) {   // first union: run starts of the first batch -> counts
    S.counts.ensure((size_t)std::max<uint32_t>(n_s, 1) * 8);
    starts_to_counts(S.starts.as<uint32_t>(), n_s, (uint32_t)S.total, S.counts.as<uint64_t>(), s);
    S.has_counts = true; S.has_runs = false;
  }
  DeviceBuffer nu, nc;
  const size_t cap = (size_t)n_s + n_d;
  nu.ensure(cap * 8);
  if (track) nc.ensure(cap * 8);
  E.misc.ensure(16);
  sorted_union_async(S.uniq.as<uint64_t>(), track ? S.counts.as<uint64_t>() : nullptr, n_s, d_mins, track ? d_starts : nullptr,
                     track ? d_counts : nullptr, n_d, (uint32_t)delta_total, nu.as<uint64_t>(), track ? nc.as<uint64_t>() : nullptr,
                     E.misc.as<uint32_t>(), E.union_tmp, dev.scratch, s);
  uint32_t n_new = 0;
  HIP_CHECK(hipMemcpyAsync(&n_new, E.misc.ptr, 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  std::swap(S.uniq.ptr, nu.ptr); std::swap(S.uniq.bytes, nu.bytes);
  if (track) { std::swap(S.counts.ptr, nc.ptr); std::swap(S.counts.bytes, nc.bytes); }
  S.n = (uint64_t)n_s + n_new;
  S.total += delta_total;
  S.has_runs = false;
  dev.count("sketch_union_on_device");
}

void Engine::union_into_device_sketch(KmerMinHash& mh, DeviceSketch& delta, hipStream_t s) {
  union_core(mh, delta.uniq.as<uint64_t>(), delta.has_counts ? nullptr : delta.starts.as<uint32_t>(),
             delta.has_counts ? delta.counts.as<uint64_t>() : nullptr, delta.n, delta.total, *this, s);
}

void Engine::union_arrays_into_device_sketch(KmerMinHash& mh, const uint64_t* d_mins, const uint64_t* d_counts, uint64_t n, hipStream_t s) {
  if (n == 0) return;
  if (mh.has_abunds && !d_counts) throw_internal("sketch union: the part carries no abundances");
  if (!mh.dev) {                       // an empty sketch: the part IS the new state
    auto ds = std::make_shared<DeviceSketch>();
    ds->uniq.ensure(n * 8);
    HIP_CHECK(hipMemcpyAsync(ds->uniq.ptr, d_mins, n * 8, hipMemcpyDeviceToDevice, s));
    if (mh.has_abunds) {
      ds->counts.ensure(n * 8);
      HIP_CHECK(hipMemcpyAsync(ds->counts.ptr, d_counts, n * 8, hipMemcpyDeviceToDevice, s));
      ds->has_counts = true;
    }
    ds->n = n; ds->total = n;
    HIP_CHECK(hipStreamSynchronize(s));
    mh.dev = ds;
    return;
  }
  union_core(mh, d_mins, nullptr, d_counts, n, n, *this, s);
}

// scaled sketches: a host-resident state goes (back) to HBM as ascending hashes + u64 counts
void KmerMinHash::to_device_state() {
  flush_pending();
  if (dev || mins.empty()) return;
  if (!(num == 0 && max_hash > 0)) throw_internal("only a scaled sketch keeps its state in HBM");
  if (has_abunds && abunds.size() != mins.size()) throw_internal("sketch with mismatched abundance vector");
  Device& d = Device::get();
  std::lock_guard<std::recursive_mutex> lock(d.mutex());
  hipStream_t s = d.stream();
  auto ds = std::make_shared<DeviceSketch>();
  const size_t n = mins.size();
  ds->uniq.ensure(n * 8);
  HIP_CHECK(hipMemcpyAsync(ds->uniq.ptr, mins.data(), n * 8, hipMemcpyHostToDevice, s));
  if (has_abunds) {
    ds->counts.ensure(n * 8);
    HIP_CHECK(hipMemcpyAsync(ds->counts.ptr, abunds.data(), n * 8, hipMemcpyHostToDevice, s));
    ds->has_counts = true;
  }
  HIP_CHECK(hipStreamSynchronize(s));
  ds->n = n; ds->total = n;
  dev = ds;
  mins.w().clear(); abunds.clear(); mirror.reset();
}

uint64_t KmerMinHash::count_common(const KmerMinHash& other) const {
  check_compatible(other);
  Engine::PairResult r;
  Engine::get().compare_pair(*this, other, num, &r);
  return r.count_common;
}

void KmerMinHash::intersection_size(const KmerMinHash& other, uint64_t* common, uint64_t* size) const {
  check_compatible(other);
  Engine::PairResult r;
  Engine::get().compare_pair(*this, other, num, &r);
  if (common) *common = r.common;
  if (size) *size = r.size;
}

// reference src/lib.rs:438-468: the common hashes themselves -- (A ^ B) ^ bottom_n(A u B) -- and the
// size of the combined sketch.  The counts come from the device walk; the bottom-n of the union is a
// prefix of the sorted union, so the list is the first `common` elements of the sorted intersection.
void KmerMinHash::intersection(const KmerMinHash& other, std::vector<uint64_t>* common, uint64_t* size) const {
  uint64_t n_common = 0, sz = 0;
  intersection_size(other, &n_common, &sz);
  materialize();
  other.materialize();
  common->clear();
  common->reserve(n_common);
  size_t i = 0, j = 0;
  while (common->size() < n_common && i < mins.size() && j < other.mins.size()) {
    if (mins[i] < other.mins[j]) i++;
    else if (other.mins[j] < mins[i]) j++;
    else { common->push_back(mins[i]); i++; j++; }
  }
  if (size) *size = sz;
}

double KmerMinHash::compare(const KmerMinHash& other) const {
  check_compatible(other);
  Engine::PairResult r;
  Engine::get().compare_pair(*this, other, num, &r);
  return r.jaccard;
}

}  // namespace smh
