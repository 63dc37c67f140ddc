// ffi.cpp -- the C ABI: every symbol of include/sourmash.h (the reference's surface, restating
// src/ffi.rs and src/utils.rs) and the additive MI355X entry points of include/sourmash_amd.h.
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <set>
#include <vector>

#include "../../include/sourmash_amd.h"
#include "minhash.hpp"
#include "signature.hpp"

using smh::Error;

struct KmerMinHash : smh::KmerMinHash {
  using smh::KmerMinHash::KmerMinHash;
  KmerMinHash() = default;
  KmerMinHash(const smh::KmerMinHash& o) : smh::KmerMinHash(o) {}
};
struct Signature : smh::Signature {
  Signature() = default;
  Signature(const smh::Signature& o) : smh::Signature(o) {}
};

namespace {

bool g_panic_hook = false;  // sourmash_init() installs the hook that records panics (utils.rs:100-104)

void set_last_error(const Error& e) {
  // utils.rs:154-166: an Err is always stored; a panic only reaches the slot through the hook
  if (e.code == smh::kPanic && !g_panic_hook) return;
  auto& slot = smh::last_error();
  slot.set = true;
  slot.code = e.code;
  slot.message = e.message;
}

// landing pad: run f, on failure fill the slot and hand back an all-zero value
template <class R, class F>
R pad(F&& f) {
  try {
    return f();
  } catch (const Error& e) {
    set_last_error(e);
  } catch (const std::bad_alloc&) {
    set_last_error(Error(smh::kPanic, "sourmash panicked: memory allocation failed"));
  } catch (const std::exception& e) {
    set_last_error(Error(smh::kPanic, std::string("sourmash panicked: ") + e.what()));
  }
  R zero;
  memset(&zero, 0, sizeof zero);
  return zero;
}
template <class F>
void pad_void(F&& f) {
  (void)pad<int>([&] { f(); return 0; });
}
// additive ABI: 0 on success, else the error code (slot also set)
template <class F>
int pad_code(F&& f) {
  try {
    f();
    return 0;
  } catch (const Error& e) {
    auto& slot = smh::last_error();
    slot.set = true; slot.code = e.code; slot.message = e.message;
    return (int)e.code;
  } catch (const std::exception& e) {
    auto& slot = smh::last_error();
    slot.set = true; slot.code = smh::kPanic; slot.message = std::string("sourmash panicked: ") + e.what();
    return (int)smh::kPanic;
  }
}

void require(const void* p, const char* what) {
  if (!p) smh::throw_panic(std::string("assertion failed: !") + what + ".is_null()");
}

SourmashStr str_from_string(const std::string& s) {  // utils.rs:201-210 from_string: owned copy
  SourmashStr r;
  r.len = s.size();
  r.data = (char*)malloc(s.size() ? s.size() : 1);
  if (s.size()) memcpy(r.data, s.data(), s.size());
  r.owned = true;
  return r;
}

bool utf8_cstr_ok(const char* s) {
  const unsigned char* p = (const unsigned char*)s;
  while (*p) {
    unsigned char c = *p;
    int need;
    if (c < 0x80) { p++; continue; }
    else if (c >= 0xC2 && c <= 0xDF) need = 1;
    else if (c >= 0xE0 && c <= 0xEF) need = 2;
    else if (c >= 0xF0 && c <= 0xF4) need = 3;
    else return false;
    unsigned char lo = 0x80, hi = 0xBF;
    if (c == 0xE0) lo = 0xA0; else if (c == 0xED) hi = 0x9F; else if (c == 0xF0) lo = 0x90; else if (c == 0xF4) hi = 0x8F;
    if (p[1] < lo || p[1] > hi) return false;
    for (int j = 2; j <= need; j++) if (p[j] < 0x80 || p[j] > 0xBF) return false;
    p += need + 1;
  }
  return true;
}

template <class T>
T** leak_array(std::vector<T*>& v, uintptr_t* size) {
  T** arr = (T**)malloc((v.size() ? v.size() : 1) * sizeof(T*));
  for (size_t i = 0; i < v.size(); i++) arr[i] = v[i];
  *size = v.size();
  return arr;
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ hashing / sketching

uint64_t hash_murmur(const char* kmer, uint64_t seed) {
  return pad<uint64_t>([&] {
    require(kmer, "kmer");
    const uint64_t off[2] = {0, (uint64_t)strlen(kmer)};
    uint64_t h = 0;
    smh::Engine::get().hash_words((const uint8_t*)kmer, off, 1, seed, &h);
    return h;
  });
}

KmerMinHash* kmerminhash_new(uint32_t n, uint32_t k, bool prot, uint64_t seed, uint64_t mx, bool track_abundance) {
  return pad<KmerMinHash*>([&] { return new KmerMinHash(n, k, prot, seed, mx, track_abundance); });
}

void kmerminhash_free(KmerMinHash* ptr) { delete ptr; }

void kmerminhash_add_sequence(KmerMinHash* ptr, const char* sequence, bool force) {
  pad_void([&] {
    require(ptr, "ptr");
    require(sequence, "sequence");
    ptr->add_sequence((const uint8_t*)sequence, strlen(sequence), force);
  });
}

void kmerminhash_add_hash(KmerMinHash* ptr, uint64_t h) {
  pad_void([&] { require(ptr, "ptr"); ptr->add_hash(h); });
}

void kmerminhash_add_word(KmerMinHash* ptr, const char* word) {
  pad_void([&] {
    require(ptr, "ptr");
    require(word, "word");
    ptr->add_word((const uint8_t*)word, strlen(word));
  });
}

void kmerminhash_add_from(KmerMinHash* ptr, const KmerMinHash* other) {
  pad_void([&] { require(ptr, "ptr"); require(other, "other"); ptr->add_from(*other); });
}

void kmerminhash_merge(KmerMinHash* ptr, const KmerMinHash* other) {
  pad_void([&] { require(ptr, "ptr"); require(other, "other"); ptr->merge(*other); });
}

// ------------------------------------------------------------------ comparing

double kmerminhash_compare(KmerMinHash* ptr, const KmerMinHash* other) {
  return pad<double>([&] { require(ptr, "ptr"); require(other, "other"); return ptr->compare(*other); });
}

uint64_t kmerminhash_count_common(KmerMinHash* ptr, const KmerMinHash* other) {
  return pad<uint64_t>([&] { require(ptr, "ptr"); require(other, "other"); return ptr->count_common(*other); });
}

uint64_t kmerminhash_intersection(KmerMinHash* ptr, const KmerMinHash* other) {
  // src/ffi.rs:304-307: the SIZE of the combined sketch; any Err of intersection() reads as 0
  return pad<uint64_t>([&] {
    require(ptr, "ptr");
    require(other, "other");
    try {
      uint64_t common = 0, size = 0;
      ptr->intersection_size(*other, &common, &size);
      return size;
    } catch (const Error& e) {
      if (e.code == smh::kPanic || e.code == smh::kInternal) throw;
      return (uint64_t)0;
    }
  });
}

// ------------------------------------------------------------------ accessors

const uint64_t* kmerminhash_get_mins(KmerMinHash* ptr) {
  return pad<const uint64_t*>([&] {
    require(ptr, "ptr"); ptr->materialize();
    uint64_t* out = (uint64_t*)malloc((ptr->mins.size() ? ptr->mins.size() : 1) * sizeof(uint64_t));
    if (!ptr->mins.empty()) memcpy(out, ptr->mins.data(), ptr->mins.size() * sizeof(uint64_t));
    return (const uint64_t*)out;
  });
}

uintptr_t kmerminhash_get_mins_size(KmerMinHash* ptr) {
  return pad<uintptr_t>([&] { require(ptr, "ptr"); return (uintptr_t)ptr->size(); });
}

uint64_t kmerminhash_get_min_idx(KmerMinHash* ptr, uint64_t idx) {
  return pad<uint64_t>([&] {
    require(ptr, "ptr"); ptr->materialize();
    if (idx >= ptr->mins.size()) smh::throw_panic("index out of bounds");
    return ptr->mins[idx];
  });
}

void kmerminhash_mins_push(KmerMinHash* ptr, uint64_t val) {
  pad_void([&] { require(ptr, "ptr"); ptr->materialize(); ptr->mins.w().push_back(val); });
}

const uint64_t* kmerminhash_get_abunds(KmerMinHash* ptr) {
  return pad<const uint64_t*>([&] {
    require(ptr, "ptr"); ptr->materialize();
    if (!ptr->has_abunds) return (const uint64_t*)nullptr;
    uint64_t* out = (uint64_t*)malloc((ptr->abunds.size() ? ptr->abunds.size() : 1) * sizeof(uint64_t));
    if (!ptr->abunds.empty()) memcpy(out, ptr->abunds.data(), ptr->abunds.size() * sizeof(uint64_t));
    return (const uint64_t*)out;
  });
}

uintptr_t kmerminhash_get_abunds_size(KmerMinHash* ptr) {
  return pad<uintptr_t>([&] { require(ptr, "ptr"); ptr->materialize(); return (uintptr_t)(ptr->has_abunds ? ptr->abunds.size() : 0); });
}

uint64_t kmerminhash_get_abund_idx(KmerMinHash* ptr, uint64_t idx) {
  return pad<uint64_t>([&] {
    require(ptr, "ptr"); ptr->materialize();
    if (!ptr->has_abunds) return (uint64_t)0;
    if (idx >= ptr->abunds.size()) smh::throw_panic("index out of bounds");
    return ptr->abunds[idx];
  });
}

void kmerminhash_abunds_push(KmerMinHash* ptr, uint64_t val) {
  pad_void([&] { require(ptr, "ptr"); ptr->materialize(); if (ptr->has_abunds) ptr->abunds.push_back(val); });
}

bool kmerminhash_is_protein(KmerMinHash* ptr) { return pad<bool>([&] { require(ptr, "ptr"); return ptr->is_protein; }); }
uint64_t kmerminhash_seed(KmerMinHash* ptr) { return pad<uint64_t>([&] { require(ptr, "ptr"); return ptr->seed; }); }
bool kmerminhash_track_abundance(KmerMinHash* ptr) { return pad<bool>([&] { require(ptr, "ptr"); return ptr->has_abunds; }); }
uint32_t kmerminhash_num(KmerMinHash* ptr) { return pad<uint32_t>([&] { require(ptr, "ptr"); return ptr->num; }); }
uint32_t kmerminhash_ksize(KmerMinHash* ptr) { return pad<uint32_t>([&] { require(ptr, "ptr"); return ptr->ksize; }); }
uint64_t kmerminhash_max_hash(KmerMinHash* ptr) { return pad<uint64_t>([&] { require(ptr, "ptr"); return ptr->max_hash; }); }

// ------------------------------------------------------------------ Signature

Signature* signature_new(void) { return pad<Signature*>([&] { return new Signature(); }); }
void signature_free(Signature* ptr) { delete ptr; }

void signature_set_name(Signature* ptr, const char* name) {
  pad_void([&] {
    require(ptr, "ptr"); require(name, "name");
    if (utf8_cstr_ok(name)) { ptr->has_name = true; ptr->name = name; }  // ffi.rs:357-359: silently ignored otherwise
  });
}
void signature_set_filename(Signature* ptr, const char* name) {
  pad_void([&] {
    require(ptr, "ptr"); require(name, "name");
    if (utf8_cstr_ok(name)) { ptr->has_filename = true; ptr->filename = name; }
  });
}
void signature_push_mh(Signature* ptr, const KmerMinHash* other) {
  pad_void([&] { require(ptr, "ptr"); require(other, "other"); ptr->signatures.push_back(*other); });
}
void signature_set_mh(Signature* ptr, const KmerMinHash* other) {
  pad_void([&] { require(ptr, "ptr"); require(other, "other"); ptr->signatures.assign(1, *other); });
}
SourmashStr signature_get_name(Signature* ptr) {
  return pad<SourmashStr>([&] { require(ptr, "ptr"); return str_from_string(ptr->has_name ? ptr->name : ""); });
}
SourmashStr signature_get_filename(Signature* ptr) {
  return pad<SourmashStr>([&] { require(ptr, "ptr"); return str_from_string(ptr->has_filename ? ptr->filename : ""); });
}
SourmashStr signature_get_license(Signature* ptr) {
  return pad<SourmashStr>([&] { require(ptr, "ptr"); return str_from_string(ptr->license); });
}
KmerMinHash* signature_first_mh(Signature* ptr) {
  return pad<KmerMinHash*>([&] {
    require(ptr, "ptr");
    if (!ptr->signatures.empty()) return new KmerMinHash(ptr->signatures[0]);
    return new KmerMinHash();  // ffi.rs:468-471 "this is totally wrong": a Default sketch
  });
}
bool signature_eq(Signature* ptr, Signature* other) {
  return pad<bool>([&] { require(ptr, "ptr"); require(other, "other"); return smh::signature_equal(*ptr, *other); });
}
SourmashStr signature_save_json(Signature* ptr) {
  return pad<SourmashStr>([&] {
    require(ptr, "ptr");
    std::string out;
    smh::signature_to_json(out, *ptr);
    return str_from_string(out);
  });
}
KmerMinHash** signature_get_mhs(Signature* ptr, uintptr_t* size) {
  return pad<KmerMinHash**>([&] {
    require(ptr, "ptr");
    require(size, "size");
    std::vector<KmerMinHash*> v;
    for (auto& mh : ptr->signatures) v.push_back(new KmerMinHash(mh));
    return leak_array(v, size);
  });
}
SourmashStr signatures_save_buffer(Signature** ptr, uintptr_t size) {
  return pad<SourmashStr>([&] {
    require(ptr, "ptr");
    std::vector<const smh::Signature*> v;
    for (uintptr_t i = 0; i < size; i++) {
      if (!ptr[i]) smh::throw_panic("called `Option::unwrap()` on a `None` value");
      v.push_back(ptr[i]);
    }
    return str_from_string(smh::signatures_to_json(v));
  });
}

static Signature** load_common(const std::string& data, uintptr_t ksize, const char* select_moltype, uintptr_t* size) {
  require(size, "size");
  if (select_moltype && !utf8_cstr_ok(select_moltype)) throw Error(smh::kUtf8Error, "invalid utf-8 sequence");
  std::vector<smh::Signature> sigs = smh::load_signatures(data.data(), data.size(), ksize, select_moltype);
  std::vector<Signature*> v;
  for (auto& s : sigs) v.push_back(new Signature(s));
  return leak_array(v, size);
}

Signature** signatures_load_path(const char* ptr, bool ignore_md5sum, uintptr_t ksize, const char* select_moltype,
                                 uintptr_t* size) {
  (void)ignore_md5sum;  // ffi.rs:555 "TODO: implement ignore_md5sum"
  return pad<Signature**>([&] {
    require(ptr, "ptr");
    if (!utf8_cstr_ok(ptr)) throw Error(smh::kUtf8Error, "invalid utf-8 sequence");
    return load_common(smh::read_file(ptr), ksize, select_moltype, size);
  });
}

Signature** signatures_load_buffer(const char* ptr, uintptr_t insize, bool ignore_md5sum, uintptr_t ksize,
                                   const char* select_moltype, uintptr_t* size) {
  (void)ignore_md5sum;
  return pad<Signature**>([&] {
    require(ptr, "ptr");
    return load_common(std::string(ptr, insize), ksize, select_moltype, size);
  });
}

// ------------------------------------------------------------------ errors and strings

void sourmash_err_clear(void) {
  auto& s = smh::last_error();
  s.set = false; s.code = 0; s.message.clear();
}

SourmashStr sourmash_err_get_backtrace(void) {
  SourmashStr r = {nullptr, 0, false};  // no backtrace is captured: the reference returns Default then
  return r;
}

SourmashErrorCode sourmash_err_get_last_code(void) {
  auto& s = smh::last_error();
  return s.set ? s.code : SOURMASH_ERROR_CODE_NO_ERROR;
}

SourmashStr sourmash_err_get_last_message(void) {
  auto& s = smh::last_error();
  if (!s.set) { SourmashStr r = {nullptr, 0, false}; return r; }
  return str_from_string(s.message);
}

void sourmash_init(void) { g_panic_hook = true; }

void sourmash_str_free(SourmashStr* s) {
  if (s && s->owned) {
    free(s->data);
    s->data = nullptr; s->len = 0; s->owned = false;
  }
}

SourmashStr sourmash_str_from_cstr(const char* s) {
  // utils.rs:220-234: borrows the bytes but marks them owned; the caller clears `owned` if it
  // keeps the memory
  return pad<SourmashStr>([&] {
    if (!s || !utf8_cstr_ok(s)) throw Error(smh::kUtf8Error, "invalid utf-8 sequence");
    SourmashStr r;
    r.data = (char*)s; r.len = strlen(s); r.owned = true;
    return r;
  });
}

// ================================================================== additive MI355X ABI

int smh_device_available(void) { return smh::Device::available() ? 1 : 0; }

int smh_device_info(int* device, int* compute_units) {
  return pad_code([&] {
    auto& d = smh::Device::get();
    if (device) *device = d.id();
    if (compute_units) *compute_units = d.cu_count();
  });
}

int smh_add_sequence_len(KmerMinHash* ptr, const char* seq, uint64_t len, bool force) {
  return pad_code([&] { require(ptr, "ptr"); require(seq, "seq"); ptr->add_sequence((const uint8_t*)seq, len, force); });
}

int smh_add_sequences(KmerMinHash* ptr, const char* seq, const uint64_t* offsets, uint32_t n_records, bool force) {
  return pad_code([&] {
    require(ptr, "ptr"); require(seq, "seq"); require(offsets, "offsets");
    if (n_records == 0) return;
    const uint64_t base = offsets[0], total = offsets[n_records] - base;
    std::vector<uint64_t> rel(n_records + 1);
    for (uint32_t i = 0; i <= n_records; i++) rel[i] = offsets[i] - base;
    ptr->add_sequences_host((const uint8_t*)seq + base, total, rel.data(), n_records, force);
  });
}

int smh_add_sequences_dev(KmerMinHash* ptr, const void* seq_dev, uint64_t total_len, const uint64_t* offsets,
                          uint32_t n_records, bool force, void* stream) {
  return pad_code([&] {
    require(ptr, "ptr"); require(seq_dev, "seq_dev"); require(offsets, "offsets");
    ptr->add_sequences_device((const uint8_t*)seq_dev, total_len, offsets, n_records, force,
                              smh::Device::get().user_stream(stream), nullptr);
  });
}

int smh_add_sequences_grouped(KmerMinHash* const* sketches, uint32_t n_sketches, const char* seq, const uint64_t* offsets,
                              const uint32_t* groups, uint32_t n_records, bool force) {
  return pad_code([&] {
    if (n_records == 0) return;
    require(sketches, "sketches"); require(seq, "seq"); require(offsets, "offsets"); require(groups, "groups");
    for (uint32_t g = 0; g < n_sketches; g++) require(sketches[g], "sketches[g]");
    auto& dev = smh::Device::get();
    auto& E = smh::Engine::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    const uint64_t base = offsets[0], total = offsets[n_records] - base;
    std::vector<uint64_t> rel(n_records + 1);
    for (uint32_t i = 0; i <= n_records; i++) rel[i] = offsets[i] - base;
    E.seqbuf.ensure(total + 64);
    if (total) HIP_CHECK(hipMemcpyAsync(E.seqbuf.ptr, seq + base, total, hipMemcpyHostToDevice, dev.stream()));
    std::vector<smh::KmerMinHash*> mhs(sketches, sketches + n_sketches);
    smh::add_sequences_grouped(mhs.data(), n_sketches, E.seqbuf.as<uint8_t>(), total, rel.data(), groups, n_records, force,
                               dev.stream(), nullptr);
  });
}

int smh_add_sequences_grouped_dev(KmerMinHash* const* sketches, uint32_t n_sketches, const void* seq_dev, uint64_t total_len,
                                  const uint64_t* offsets, const uint32_t* groups, uint32_t n_records, bool force,
                                  void* stream) {
  return pad_code([&] {
    if (n_records == 0) return;
    require(sketches, "sketches"); require(seq_dev, "seq_dev"); require(offsets, "offsets"); require(groups, "groups");
    for (uint32_t g = 0; g < n_sketches; g++) require(sketches[g], "sketches[g]");
    std::vector<smh::KmerMinHash*> mhs(sketches, sketches + n_sketches);
    smh::add_sequences_grouped(mhs.data(), n_sketches, (const uint8_t*)seq_dev, total_len, offsets, groups, n_records, force,
                               smh::Device::get().user_stream(stream), nullptr);
  });
}

int smh_add_many(KmerMinHash* ptr, const uint64_t* hashes, uint64_t n) {
  return pad_code([&] { require(ptr, "ptr"); if (n) require(hashes, "hashes"); ptr->materialize(); ptr->add_many(hashes, n); });
}

int smh_add_many_with_abund(KmerMinHash* ptr, const uint64_t* hashes, const uint64_t* abunds, uint64_t n) {
  return pad_code([&] {
    require(ptr, "ptr");
    if (n) { require(hashes, "hashes"); require(abunds, "abunds"); }
    ptr->add_many_with_abund(hashes, abunds, n);
  });
}

int smh_check_compatible(const KmerMinHash* ptr, const KmerMinHash* other) {
  return pad_code([&] { require(ptr, "ptr"); require(other, "other"); ptr->check_compatible(*other); });
}

int smh_intersection(const KmerMinHash* ptr, const KmerMinHash* other, uint64_t** common_out, uint64_t* n_common,
                     uint64_t* union_size) {
  return pad_code([&] {
    require(ptr, "ptr"); require(other, "other"); require(common_out, "common_out"); require(n_common, "n_common");
    std::vector<uint64_t> common;
    uint64_t size = 0;
    ptr->intersection(*other, &common, &size);
    uint64_t* out = (uint64_t*)malloc((common.empty() ? 1 : common.size()) * sizeof(uint64_t));
    if (!out) smh::throw_internal("out of memory");
    if (!common.empty()) memcpy(out, common.data(), common.size() * sizeof(uint64_t));
    *common_out = out; *n_common = common.size();
    if (union_size) *union_size = size;
  });
}

int smh_hash_words(const char* bytes, const uint64_t* offsets, uint32_t n, uint64_t seed, uint64_t* out) {
  return pad_code([&] {
    require(offsets, "offsets"); require(out, "out");
    smh::Engine::get().hash_words((const uint8_t*)bytes, offsets, n, seed, out);
  });
}

int smh_compare_block(KmerMinHash* const* rows, uint32_t n_rows, KmerMinHash* const* cols, uint32_t n_cols,
                      double* jaccard, uint64_t* common, uint64_t* size, uint64_t* count_common,
                      double* containment) {
  return pad_code([&] {
    if (n_rows == 0 || n_cols == 0) return;
    require(rows, "rows"); require(cols, "cols");
    std::vector<const smh::KmerMinHash*> R(n_rows), C(n_cols);
    std::vector<uint32_t> nums(n_rows);
    for (uint32_t i = 0; i < n_rows; i++) { require(rows[i], "rows[i]"); R[i] = rows[i]; nums[i] = rows[i]->num; }
    for (uint32_t j = 0; j < n_cols; j++) { require(cols[j], "cols[j]"); C[j] = cols[j]; }
    for (uint32_t i = 0; i < n_rows; i++)
      for (uint32_t j = 0; j < n_cols; j++) R[i]->check_compatible(*C[j]);
    smh::Engine::get().compare_host(R, C, nums.data(), 0, common, size, jaccard, count_common, containment);
  });
}

int smh_compare_block_dev(const uint64_t* row_hashes_dev, const uint64_t* row_offsets, uint32_t n_rows,
                          const uint64_t* col_hashes_dev, const uint64_t* col_offsets, uint32_t n_cols,
                          uint32_t num, double* jaccard_dev, uint64_t* common_dev, uint64_t* size_dev,
                          uint64_t* count_common_dev, double* containment_dev, void* stream) {
  return pad_code([&] {
    if (n_rows == 0 || n_cols == 0) return;
    require(row_offsets, "row_offsets"); require(col_offsets, "col_offsets");
    auto& dev = smh::Device::get();
    auto& E = smh::Engine::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.user_stream(stream);
    uint32_t mr = 0, mc = 0;
    for (uint32_t i = 0; i < n_rows; i++) mr = std::max<uint32_t>(mr, (uint32_t)(row_offsets[i + 1] - row_offsets[i]));
    for (uint32_t j = 0; j < n_cols; j++) mc = std::max<uint32_t>(mc, (uint32_t)(col_offsets[j + 1] - col_offsets[j]));
    E.cmp_oa.ensure((size_t)(n_rows + 1) * 8);
    E.cmp_ob.ensure((size_t)(n_cols + 1) * 8);
    HIP_CHECK(hipMemcpyAsync(E.cmp_oa.ptr, row_offsets, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(E.cmp_ob.ptr, col_offsets, (size_t)(n_cols + 1) * 8, hipMemcpyHostToDevice, s));
    smh::SketchSet R, C;
    R.hashes = row_hashes_dev; R.offsets = E.cmp_oa.as<uint64_t>(); R.n = n_rows; R.h_offsets = row_offsets;
    C.hashes = col_hashes_dev; C.offsets = E.cmp_ob.as<uint64_t>(); C.n = n_cols; C.h_offsets = col_offsets;
    smh::CompareOut o;
    o.jaccard = jaccard_dev; o.common = common_dev; o.size = size_dev; o.count_common = count_common_dev;
    o.containment = containment_dev;
    const bool same_sets = row_hashes_dev == col_hashes_dev && n_rows == n_cols &&
                           std::memcmp(row_offsets, col_offsets, (size_t)(n_rows + 1) * 8) == 0;
    smh::launch_compare_block(R, C, num, nullptr, o, dev, s, mr, mc, row_offsets[n_rows] - row_offsets[0],
                              col_offsets[n_cols] - col_offsets[0], same_sets);
    HIP_CHECK(hipStreamSynchronize(s));  // the offset staging buffers are reused by the next call
  });
}

int smh_find(KmerMinHash* const* nodes, uint32_t n_nodes, const KmerMinHash* query, double threshold,
             bool containment, uint32_t* out_indices, uint32_t* out_count) {
  return pad_code([&] {
    require(out_count, "out_count");
    *out_count = 0;
    if (n_nodes == 0) return;
    require(nodes, "nodes"); require(query, "query"); require(out_indices, "out_indices");
    std::vector<const smh::KmerMinHash*> R(n_nodes), C(1, query);
    std::vector<uint32_t> nums(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
      require(nodes[i], "nodes[i]");
      R[i] = nodes[i]; nums[i] = nodes[i]->num;
      R[i]->check_compatible(*query);   // Leaf::similarity unwraps compare(): an Err is fatal there too
    }
    std::vector<double> val(n_nodes);
    if (containment) smh::Engine::get().compare_host(R, C, nums.data(), 0, nullptr, nullptr, nullptr, nullptr, val.data());
    else smh::Engine::get().compare_host(R, C, nums.data(), 0, nullptr, nullptr, val.data(), nullptr, nullptr);
    uint32_t k = 0;
    for (uint32_t i = 0; i < n_nodes; i++)
      if (val[i] > threshold) out_indices[k++] = i;   // NaN (empty node, containment) is never > threshold
    *out_count = k;
  });
}

int smh_most_common(const KmerMinHash* leaf, KmerMinHash* const* candidates, uint32_t n, uint32_t* best_pos,
                    uint64_t* best_common) {
  return pad_code([&] {
    if (best_pos) *best_pos = 0;
    if (best_common) *best_common = 0;
    if (n == 0) return;
    require(leaf, "leaf"); require(candidates, "candidates");
    std::vector<const smh::KmerMinHash*> R(1, leaf), C(n);
    for (uint32_t j = 0; j < n; j++) { require(candidates[j], "candidates[j]"); C[j] = candidates[j]; leaf->check_compatible(*C[j]); }
    std::vector<uint64_t> cc(n);
    smh::Engine::get().compare_host(R, C, nullptr, leaf->num, nullptr, nullptr, nullptr, cc.data(), nullptr);
    uint32_t pos = 0;
    uint64_t mx = 0;
    for (uint32_t j = 0; j < n; j++)
      if (cc[j] > mx) { mx = cc[j]; pos = j; }
    if (best_pos) *best_pos = pos;
    if (best_common) *best_common = mx;
  });
}

// ------------------------------------------------------------------ resident index

struct SmhIndex {
  smh::DeviceBuffer hashes, offsets, nums;
  std::vector<uint64_t> h_offsets;
  std::vector<uint32_t> h_nums;
  std::vector<smh::KmerMinHash> params;   // parameters only (mins cleared): check_compatible per node
  uint32_t max_len = 0;
  uint32_t n = 0;
  // the dictionary of the resident set (dense ranks, components, frequent hashes), built by the first all-vs-all compare
  // of the index with itself and kept: later ones skip the pre-pass (the nodes of an index never change)
  smh::CollectionDict* dict = nullptr;
  uint32_t dict_split = 0;      // the frequent-hash setting the dictionary was built under
  SmhIndex() { std::lock_guard<std::mutex> g(registry_mu()); registry().insert(this); }
  ~SmhIndex() {
    { std::lock_guard<std::mutex> g(registry_mu()); registry().erase(this); }
    if (dict) smh::collection_free(dict);
  }
  void drop_dict() { if (dict) { smh::collection_free(dict); dict = nullptr; } }
  // the live indexes: smh_release_workspace() drops their cached dictionaries (memory no other allocator can see)
  static std::set<SmhIndex*>& registry() { static auto* r = new std::set<SmhIndex*>(); return *r; }
  static std::mutex& registry_mu() { static auto* m = new std::mutex(); return *m; }
};

SmhIndex* smh_index_new(KmerMinHash* const* nodes, uint32_t n_nodes) {
  return pad<SmhIndex*>([&] {
    if (n_nodes) require(nodes, "nodes");
    auto idx = std::make_unique<SmhIndex>();
    idx->n = n_nodes;
    idx->h_nums.resize(n_nodes);
    std::vector<const smh::KmerMinHash*> v(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
      require(nodes[i], "nodes[i]");
      v[i] = nodes[i];
      idx->h_nums[i] = nodes[i]->num;
      smh::KmerMinHash p(nodes[i]->num, nodes[i]->ksize, nodes[i]->is_protein, nodes[i]->seed, nodes[i]->max_hash, false);
      idx->params.push_back(p);
    }
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.stream();
    smh::SketchSet set;
    smh::Engine::get().pack_sketches(v, idx->hashes, idx->offsets, &set, &idx->max_len, &idx->h_offsets, s);
    idx->nums.ensure((size_t)n_nodes * 4 + 4);
    if (n_nodes) HIP_CHECK(hipMemcpyAsync(idx->nums.ptr, idx->h_nums.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return idx.release();
  });
}

void smh_index_free(SmhIndex* index) { delete index; }
void smh_index_drop_dictionary(SmhIndex* index) {
  if (!index) return;
  (void)pad_code([&] {
    std::lock_guard<std::recursive_mutex> lock(smh::Device::get().mutex());
    index->drop_dict();
  });
}
uint32_t smh_index_len(const SmhIndex* index) { return index ? index->n : 0; }

namespace {
// rows = resident index, cols = one host sketch: N x 1 block on the device, values back on the host
void index_vs_one(SmhIndex* index, const smh::KmerMinHash* q, bool q_is_row, double* jac, double* cont, uint64_t* cc) {
  auto& dev = smh::Device::get();
  auto& E = smh::Engine::get();
  std::lock_guard<std::recursive_mutex> lock(dev.mutex());
  hipStream_t s = dev.stream();
  q->materialize();
  const uint32_t n = index->n;
  const uint64_t qoff[2] = {0, (uint64_t)q->mins.size()};
  E.cmp_b.ensure(q->mins.size() * 8 + 8);
  E.cmp_ob.ensure(16);
  E.cmp_out.ensure((size_t)n * 8 * 3 + 64);
  if (!q->mins.empty())
    HIP_CHECK(hipMemcpyAsync(E.cmp_b.ptr, q->mins.data(), q->mins.size() * 8, hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMemcpyAsync(E.cmp_ob.ptr, qoff, 16, hipMemcpyHostToDevice, s));
  smh::SketchSet I, Q;
  I.hashes = index->hashes.as<uint64_t>(); I.offsets = index->offsets.as<uint64_t>(); I.n = n;
  Q.hashes = E.cmp_b.as<uint64_t>(); Q.offsets = E.cmp_ob.as<uint64_t>(); Q.n = 1;
  double* d_j = E.cmp_out.as<double>();
  double* d_c = d_j + n;
  uint64_t* d_cc = reinterpret_cast<uint64_t*>(d_c + n);
  smh::CompareOut o;
  o.jaccard = jac ? d_j : nullptr; o.containment = cont ? d_c : nullptr; o.count_common = cc ? d_cc : nullptr;
  if (!q_is_row)
    smh::launch_compare_block(I, Q, 0, index->nums.as<uint32_t>(), o, dev, s, index->max_len, (uint32_t)q->mins.size(),
                              index->h_offsets.back(), q->mins.size());
  else
    smh::launch_compare_block(Q, I, q->num, nullptr, o, dev, s, (uint32_t)q->mins.size(), index->max_len, q->mins.size(),
                              index->h_offsets.back());
  if (jac) HIP_CHECK(hipMemcpyAsync(jac, d_j, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  if (cont) HIP_CHECK(hipMemcpyAsync(cont, d_c, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  if (cc) HIP_CHECK(hipMemcpyAsync(cc, d_cc, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
}
}  // namespace

int smh_index_find(SmhIndex* index, const KmerMinHash* query, double threshold, bool containment, uint32_t* out_indices,
                   uint32_t* out_count) {
  return pad_code([&] {
    require(index, "index"); require(query, "query"); require(out_count, "out_count");
    *out_count = 0;
    if (index->n == 0) return;
    require(out_indices, "out_indices");
    for (auto& p : index->params) p.check_compatible(*query);
    std::vector<double> val(index->n);
    index_vs_one(index, query, false, containment ? nullptr : val.data(), containment ? val.data() : nullptr, nullptr);
    uint32_t k = 0;
    for (uint32_t i = 0; i < index->n; i++)
      if (val[i] > threshold) out_indices[k++] = i;
    *out_count = k;
  });
}

int smh_index_most_common(SmhIndex* index, const KmerMinHash* leaf, uint32_t* best_pos, uint64_t* best_common) {
  return pad_code([&] {
    require(index, "index"); require(leaf, "leaf");
    if (best_pos) *best_pos = 0;
    if (best_common) *best_common = 0;
    if (index->n == 0) return;
    for (auto& p : index->params) leaf->check_compatible(p);
    std::vector<uint64_t> cc(index->n);
    index_vs_one(index, leaf, true, nullptr, nullptr, cc.data());
    uint32_t pos = 0; uint64_t mx = 0;
    for (uint32_t j = 0; j < index->n; j++) if (cc[j] > mx) { mx = cc[j]; pos = j; }
    if (best_pos) *best_pos = pos;
    if (best_common) *best_common = mx;
  });
}

int smh_index_compare(SmhIndex* rows, SmhIndex* cols, double* jaccard, uint64_t* common, uint64_t* size,
                      uint64_t* count_common, double* containment) {
  return pad_code([&] {
    require(rows, "rows"); require(cols, "cols");
    const size_t np = (size_t)rows->n * cols->n;
    if (np == 0) return;
    for (auto& r : rows->params) for (auto& c : cols->params) r.check_compatible(c);
    auto& dev = smh::Device::get();
    auto& E = smh::Engine::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.stream();
    E.cmp_out.ensure(np * 8 * 5 + 64);
    uint64_t* d_common = E.cmp_out.as<uint64_t>();
    uint64_t* d_size = d_common + np;
    double* d_jac = reinterpret_cast<double*>(d_size + np);
    uint64_t* d_cc = reinterpret_cast<uint64_t*>(d_jac + np);
    double* d_cont = reinterpret_cast<double*>(d_cc + np);
    smh::SketchSet R, C;
    R.hashes = rows->hashes.as<uint64_t>(); R.offsets = rows->offsets.as<uint64_t>(); R.n = rows->n;
    C.hashes = cols->hashes.as<uint64_t>(); C.offsets = cols->offsets.as<uint64_t>(); C.n = cols->n;
    R.h_offsets = rows->h_offsets.data(); C.h_offsets = cols->h_offsets.data();
    smh::CompareOut o;
    o.common = common ? d_common : nullptr; o.size = size ? d_size : nullptr; o.jaccard = jaccard ? d_jac : nullptr;
    o.count_common = count_common ? d_cc : nullptr; o.containment = containment ? d_cont : nullptr;
    // one num for every row: pass it as the launch-wide value (lets an index against itself use symmetry)
    bool uniform = true;
    for (uint32_t v : rows->h_nums) uniform &= v == rows->h_nums[0];
    const smh::CompareTuning tune = smh::compare_get_tuning();
    const bool block_route = tune.route == smh::kRouteAuto ? (np >= 4096 && rows->n >= 16) : (tune.route == smh::kRouteComponents || tune.route == smh::kRouteTiled);
    if (rows == cols && block_route && rows->h_offsets.back() > 0) {
      // an index against itself: its dictionary is built once and reused (the pre-pass is most of a sparse matrix's time)
      if (rows->dict && rows->dict_split != tune.split_frequent) { smh::collection_free(rows->dict); rows->dict = nullptr; }
      if (!rows->dict) {
        rows->dict = smh::collection_begin(R.hashes, R.offsets, rows->h_offsets.data(), rows->n, 1, 0, dev, s);
        smh::collection_finish(rows->dict, nullptr, dev, s);
        rows->dict_split = tune.split_frequent;
      }
      smh::collection_compare(rows->dict, 0, rows->n, 0, rows->n, uniform ? rows->h_nums[0] : 0,
                              uniform ? nullptr : rows->nums.as<uint32_t>(), 1, o, dev, s);
    } else
    smh::launch_compare_block(R, C, uniform ? rows->h_nums[0] : 0, uniform ? nullptr : rows->nums.as<uint32_t>(), o, dev, s,
                              rows->max_len, cols->max_len, rows->h_offsets.back(), cols->h_offsets.back(), rows == cols);
    if (common) HIP_CHECK(hipMemcpyAsync(common, d_common, np * 8, hipMemcpyDeviceToHost, s));
    if (size) HIP_CHECK(hipMemcpyAsync(size, d_size, np * 8, hipMemcpyDeviceToHost, s));
    if (jaccard) HIP_CHECK(hipMemcpyAsync(jaccard, d_jac, np * 8, hipMemcpyDeviceToHost, s));
    if (count_common) HIP_CHECK(hipMemcpyAsync(count_common, d_cc, np * 8, hipMemcpyDeviceToHost, s));
    if (containment) HIP_CHECK(hipMemcpyAsync(containment, d_cont, np * 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  });
}

// ---- a scaled sketch's state as device arrays: the cross-rank union of partial sketches ----
int smh_sketch_export_dev(KmerMinHash* ptr, uint64_t* mins_dev, uint64_t* abunds_dev, uint64_t capacity, uint64_t* n_out, void* stream) {
  return pad_code([&] {
    require(ptr, "ptr"); require(n_out, "n_out");
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.user_stream(stream);
    if (!(ptr->num == 0 && ptr->max_hash > 0)) smh::throw_internal("smh_sketch_export_dev: only scaled sketches (num == 0, max_hash > 0)");
    ptr->flush_pending();
    if (!ptr->dev && ptr->has_abunds && ptr->abunds.size() != ptr->mins.size())
      smh::throw_internal("smh_sketch_export_dev: the sketch's abundance vector does not match its hashes (a state the reference's merge can "
                          "leave behind, quirks Q5/Q6); it has no device form");
    ptr->to_device_state();
    const uint64_t n = ptr->dev ? ptr->dev->n : 0;
    *n_out = n;
    if (n == 0 || !mins_dev) return;
    // a buffer sized from an earlier query, and hashes added since: say so (n_out holds the size needed) -- success with an
    // unfilled buffer would be taken for an empty part
    if (capacity < n) smh::throw_internal("smh_sketch_export_dev: capacity is smaller than the sketch (*n_out holds the size needed)");
    smh::DeviceSketch& S = *ptr->dev;
    HIP_CHECK(hipMemcpyAsync(mins_dev, S.uniq.ptr, n * 8, hipMemcpyDeviceToDevice, s));
    if (abunds_dev && ptr->has_abunds) {
      if (S.has_counts) HIP_CHECK(hipMemcpyAsync(abunds_dev, S.counts.ptr, n * 8, hipMemcpyDeviceToDevice, s));
      else smh::starts_to_counts(S.starts.as<uint32_t>(), (uint32_t)n, (uint32_t)S.total, abunds_dev, s);
    }
    HIP_CHECK(hipStreamSynchronize(s));
  });
}
int smh_sketch_absorb_dev(KmerMinHash* ptr, const uint64_t* mins_dev, const uint64_t* abunds_dev, const uint64_t* part_starts,
                          const uint64_t* part_lens, uint32_t n_parts, void* stream) {
  return pad_code([&] {
    require(ptr, "ptr");
    if (n_parts == 0) return;
    require(mins_dev, "mins_dev"); require(part_starts, "part_starts"); require(part_lens, "part_lens");
    if (!(ptr->num == 0 && ptr->max_hash > 0)) smh::throw_internal("smh_sketch_absorb_dev: only scaled sketches (num == 0, max_hash > 0)");
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.user_stream(stream);
    ptr->to_device_state();
    for (uint32_t k = 0; k < n_parts; k++)
      smh::Engine::get().union_arrays_into_device_sketch(*ptr, mins_dev + part_starts[k], abunds_dev ? abunds_dev + part_starts[k] : nullptr,
                                                         part_lens[k], s);
    HIP_CHECK(hipStreamSynchronize(s));
  });
}

// ---- SmhCollection: the dictionary of one collection (all-vs-all, shareable among ranks) ----
struct SmhCollection {
  smh::CollectionDict* d = nullptr;
  ~SmhCollection() { if (d) smh::collection_free(d); }
};

SmhCollection* smh_collection_begin(const uint64_t* hashes_dev, const uint64_t* offsets, uint32_t n, uint32_t world, uint32_t rank,
                                    void* stream) {
  SmhCollection* out = nullptr;
  (void)pad_code([&] {
    require(hashes_dev, "hashes_dev"); require(offsets, "offsets");
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.user_stream(stream);
    std::unique_ptr<SmhCollection> c(new SmhCollection());
    c->d = smh::collection_begin(hashes_dev, nullptr, offsets, n, world, rank, dev, s);
    // the share is complete when the call returns (the caller all-gathers it next); a single owner has nobody to hand it
    // to: its work is left open on the stream, and every later entry point -- on any stream -- is ordered behind it
    // (Device::leave_open: the shared scratch buffers it still reads are not rewritten under it)
    if (world > 1) HIP_CHECK(hipStreamSynchronize(s));
    else dev.leave_open(s);
    out = c.release();
  });
  return out;
}
void smh_collection_free(SmhCollection* c) {
  if (!c) return;
  (void)pad_code([&] {
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    delete c;
  });
}
uint64_t smh_collection_share_bytes(const SmhCollection* c) { return c && c->d ? smh::collection_share_bytes(c->d) : 0; }
const void* smh_collection_share(const SmhCollection* c) { return c && c->d ? smh::collection_share(c->d) : nullptr; }
int smh_collection_share_to(const SmhCollection* c, void* dst_dev, void* stream) {
  return pad_code([&] {
    require(c, "collection"); require(dst_dev, "dst_dev");
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.user_stream(stream);
    HIP_CHECK(hipMemcpyAsync(dst_dev, smh::collection_share(c->d), smh::collection_share_bytes(c->d), hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
  });
}
int smh_collection_finish(SmhCollection* c, const void* gathered_dev, void* stream) {
  return pad_code([&] {
    require(c, "collection");
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.user_stream(stream);
    smh::collection_finish(c->d, gathered_dev, dev, s);
    if (gathered_dev) HIP_CHECK(hipStreamSynchronize(s));   // the gathered buffer may be released by the caller now
    else dev.leave_open(s);                                  // (a single owner: ordered, not waited for -- see smh_collection_begin)
  });
}
int smh_collection_compare(SmhCollection* c, uint32_t row_lo, uint32_t row_hi, uint32_t num, uint32_t ownership,
                           double* jaccard_dev, uint64_t* common_dev, uint64_t* size_dev, uint64_t* count_common_dev,
                           double* containment_dev, void* stream) {
  return pad_code([&] {
    require(c, "collection");
    if (ownership > 2) smh::throw_internal("smh_collection_compare: ownership must be 0, 1 or 2");
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.user_stream(stream);
    smh::CompareOut o;
    o.jaccard = jaccard_dev; o.common = common_dev; o.size = size_dev; o.count_common = count_common_dev;
    o.containment = containment_dev;
    smh::collection_compare(c->d, row_lo, row_hi, 0, smh::collection_len(c->d), num, nullptr, ownership, o, dev, s);
    HIP_CHECK(hipStreamSynchronize(s));
  });
}

int smh_mirror_pack(const void* out_dev, uint32_t n_local, uint32_t n_total, const uint32_t* col_lo, const uint32_t* col_hi,
                    uint32_t n_blocks, void* packed_dev, void* stream) {
  return pad_code([&] {
    if (n_blocks == 0) return;
    require(out_dev, "out_dev"); require(col_lo, "col_lo"); require(col_hi, "col_hi"); require(packed_dev, "packed_dev");
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.user_stream(stream);
    uint64_t at = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
      if (col_hi[b] < col_lo[b] || col_hi[b] > n_total) smh::throw_internal("smh_mirror_pack: block outside the matrix");
      smh::launch_mirror_pack(out_dev, n_local, n_total, col_lo[b], col_hi[b], (uint64_t*)packed_dev + at, s);
      at += (uint64_t)(col_hi[b] - col_lo[b]) * n_local;
    }
    HIP_CHECK(hipStreamSynchronize(s));
  });
}
int smh_mirror_apply(void* out_dev, uint32_t row_lo, uint32_t n_local, uint32_t n_total, const uint32_t* peer_lo, const uint32_t* peer_hi,
                     uint32_t n_blocks, const void* recv_dev, void* stream) {
  return pad_code([&] {
    if (n_blocks == 0) return;
    require(out_dev, "out_dev"); require(peer_lo, "peer_lo"); require(peer_hi, "peer_hi"); require(recv_dev, "recv_dev");
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.user_stream(stream);
    uint64_t at = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
      if (peer_hi[b] < peer_lo[b] || peer_hi[b] > n_total) smh::throw_internal("smh_mirror_apply: block outside the matrix");
      smh::launch_mirror_apply(out_dev, row_lo, n_local, n_total, peer_lo[b], peer_hi[b], (const uint64_t*)recv_dev + at, s);
      at += (uint64_t)(peer_hi[b] - peer_lo[b]) * n_local;
    }
    HIP_CHECK(hipStreamSynchronize(s));
  });
}

int smh_synth_dna_dev(void* out_dev, uint64_t start, uint64_t len, uint64_t seed, uint64_t n_every, void* stream) {
  return pad_code([&] {
    require(out_dev, "out_dev");
    auto& dev = smh::Device::get();
    hipStream_t s = dev.user_stream(stream);
    smh::launch_synth_dna((uint8_t*)out_dev, start, len, seed, n_every, s);
    if (!stream) HIP_CHECK(hipStreamSynchronize(s));   // own stream: the buffer is complete on return
  });
}

int smh_sort_u64(uint64_t* keys, uint32_t* payload, uintptr_t n) {
  return pad_code([&] {
    if (n == 0) return;
    require(keys, "keys");
    auto& dev = smh::Device::get();
    std::lock_guard<std::recursive_mutex> lock(dev.mutex());
    hipStream_t s = dev.stream();
    smh::DeviceBuffer k[2], v[2];
    for (int i = 0; i < 2; i++) { k[i].ensure(n * 8); if (payload) v[i].ensure(n * 4); }
    HIP_CHECK(hipMemcpyAsync(k[0].ptr, keys, n * 8, hipMemcpyHostToDevice, s));
    if (payload) HIP_CHECK(hipMemcpyAsync(v[0].ptr, payload, n * 4, hipMemcpyHostToDevice, s));
    int cur;
    if (payload) cur = smh::radix_sort_u64_v32(k[0].as<uint64_t>(), k[1].as<uint64_t>(), v[0].as<uint32_t>(), v[1].as<uint32_t>(), n,
                                               dev.scratch, s);
    else cur = smh::radix_sort_u64(k[0].as<uint64_t>(), k[1].as<uint64_t>(), nullptr, nullptr, n, dev.scratch, s);
    HIP_CHECK(hipMemcpyAsync(keys, k[cur].ptr, n * 8, hipMemcpyDeviceToHost, s));
    if (payload) HIP_CHECK(hipMemcpyAsync(payload, v[cur].ptr, n * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  });
}

void smh_compare_last_stats(SmhCompareStats* out) {
  if (!out) return;
  const smh::CompareStats st = smh::compare_last_stats();
  out->route = st.route; out->rows_per_tile = st.rows_per_tile;
  out->tiles_visited = st.tiles_visited; out->tiles_total = st.tiles_total; out->pairs_per_tile = st.pairs_per_tile;
  out->lds_overflow_steps = st.lds_overflow_steps;
  out->frequent_hashes = st.frequent_hashes;
  out->pipelined = st.pipelined;
  out->span_halvings = st.span_halvings;
  out->prefetched_after_halving = st.prefetched_after_halving;
}
void smh_compare_get_tuning(SmhCompareTuning* out) {
  if (!out) return;
  const smh::CompareTuning t = smh::compare_get_tuning();
  out->route = t.route; out->visit_all_tiles = t.visit_all_tiles; out->use_symmetry = t.use_symmetry;
  out->comp_pairs_limit = t.comp_pairs_limit; out->split_frequent = t.split_frequent; out->dictionary = t.dictionary;
  out->no_range_masks = t.no_range_masks;
}
int smh_compare_set_tuning(const SmhCompareTuning* in) {
  return pad_code([&] {
    smh::CompareTuning t;   // NULL = defaults
    if (in) {
      if (in->route > smh::kRouteTiled) smh::throw_internal("smh_compare_set_tuning: unknown route");
      t.route = in->route; t.visit_all_tiles = in->visit_all_tiles; t.use_symmetry = in->use_symmetry;
      t.comp_pairs_limit = in->comp_pairs_limit; t.split_frequent = in->split_frequent;
      if (in->dictionary > 1) smh::throw_internal("smh_compare_set_tuning: unknown dictionary build");
      t.dictionary = in->dictionary;
      t.no_range_masks = in->no_range_masks ? 1u : 0u;
    }
    smh::compare_set_tuning(t);
  });
}

// test hook (host only, no device): the compare block's tile planning
int smh_release_workspace(void) {
  return pad_code([&] {
    {
      // the dictionaries resident indexes cached for their all-vs-all compares (ranks, roots, the partition table: up to
      // hundreds of MB for 10 000 long sketches); the next smh_index_compare of an index with itself rebuilds its own
      std::lock_guard<std::recursive_mutex> lock(smh::Device::get().mutex());
      std::lock_guard<std::mutex> g(SmhIndex::registry_mu());
      for (SmhIndex* i : SmhIndex::registry()) i->drop_dict();
    }
    smh::Engine::get().release_workspace();
  });
}

void smh_pool_set_limit(uint64_t bytes) { (void)pad_code([&] { smh::device_pool_set_limit((size_t)bytes); }); }
uint64_t smh_pool_bytes(void) { return (uint64_t)smh::device_pool_bytes(); }

void smh_profile_enable(int on) {
  (void)pad_code([&] { smh::Device::get().profile_enable(on != 0); });
}
void smh_profile_reset(void) {
  (void)pad_code([&] { smh::Device::get().prof_reset(); });
}
int smh_profile_get(const char* name, double* total_ms, uint64_t* launches) {
  return pad_code([&] {
    require(name, "name");
    auto t = smh::Device::get().prof_get(name);
    if (total_ms) *total_ms = t.ms;
    if (launches) *launches = t.launches;
  });
}

}  // extern "C"
