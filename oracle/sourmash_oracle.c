/*
 * oracle/sourmash_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement in plain C of the MinHash hot path of luizirber/sourmash-rust
 * (crate `sourmash` v0.1.1).  It is the parity checker for the HIP kernels and
 * the timed single-core CPU baseline ("port").  It is deliberately the SAME
 * algorithm as the reference, including its cost profile (whole-input uppercase
 * copy, one heap allocation per k-mer for the reverse complement, O(k) window
 * validation, sorted vector + memmove insert, two merges + two intersections
 * per compare) and its quirks Q1..Q10 of SURVEY.md 7.
 *
 * Why a restatement: the reference is Rust; this image has no rustc/cargo and
 * none of its crates.io dependencies, so it can be neither built nor imported
 * (SURVEY.md 8c).  The hash arithmetic is not in the reference tree at all: it
 * is the third-party crate `murmurhash3 ~0.0.5` (Cargo.toml:49, no lockfile),
 * function murmurhash3_x64_128(&[u8], u64) -> (u64, u64), called at
 * src/lib.rs:33-35.  Restated here from Austin Appleby's public-domain
 * MurmurHash3_x64_128 definition, with the seed widened to u64 as that crate
 * does (Q10).
 *
 * PINNING (checked by tests/test_oracle_kat.py, all green):
 *   - tests/test.rs:5           hash("ACG", 42) == 1731421407650554201
 *   - tests/minhash.rs:13-16    add_sequence("ATGR") is an error
 *   - tests/minhash.rs:39-51    the 8 exact merged mins for k=10,num=20
 *   - tests/minhash.rs:54-83    compare == 1.0 / >= 0.3
 *   - src/index/sbt.rs:543-589  hit counts 1/2 (similarity) and 2/4
 *                               (containment) on tests/data/.sbt.v5 leaves
 *   - tests/data fixture md5sum fields (pin src/lib.rs:72-77) -- host side.
 * Unpinned by any reference test: the protein arm input->output (the reference
 * holds no sequence->sketch vector for it) and seeds >= 2^32.  For those the
 * bar is agreement of this file with the independent oracle/pyoracle.py.
 */
#include "sourmash_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* MurmurHash3 x64_128 (crate murmurhash3 ~0.0.5; call site src/lib.rs:33-35) */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

static inline uint64_t load_le64(const uint8_t *p) {
  uint64_t v = 0;
  for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
  return v;
}

void omh_murmur3_x64_128(const uint8_t *key, size_t len, uint64_t seed, uint64_t out[2]) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = seed, h2 = seed;
  size_t nblocks = len / 16;
  for (size_t i = 0; i < nblocks; i++) {
    uint64_t k1 = load_le64(key + 16 * i), k2 = load_le64(key + 16 * i + 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t *tail = key + nblocks * 16;
  uint64_t k1 = 0, k2 = 0;
  size_t rem = len & 15;
  for (size_t i = rem; i > 8; i--) k2 |= (uint64_t)tail[i - 1] << (8 * (i - 9));
  if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  for (size_t i = (rem > 8 ? 8 : rem); i > 0; i--) k1 |= (uint64_t)tail[i - 1] << (8 * (i - 1));
  if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2; h2 += h1;
  out[0] = h1; out[1] = h2;
}

/* src/lib.rs:33-35 _hash_murmur: first word of the 128-bit digest */
uint64_t omh_hash_murmur(const uint8_t *key, size_t len, uint64_t seed) {
  uint64_t o[2];
  omh_murmur3_x64_128(key, len, seed, o);
  return o[0];
}

/* ------------------------------------------------------------------ */
/* growable u64 vectors standing in for Vec<u64> */

static void vec_reserve(uint64_t **v, size_t *cap, size_t want) {
  if (want <= *cap) return;
  size_t nc = *cap ? *cap : 16;
  while (nc < want) nc *= 2;
  *v = (uint64_t *)realloc(*v, nc * sizeof(uint64_t));
  *cap = nc;
}
static void vec_push(uint64_t **v, size_t *n, size_t *cap, uint64_t x) {
  vec_reserve(v, cap, *n + 1);
  (*v)[(*n)++] = x;
}
/* Vec::insert: panics (returns -1 here) when pos > len */
static int vec_insert(uint64_t **v, size_t *n, size_t *cap, size_t pos, uint64_t x) {
  if (pos > *n) return -1;
  vec_reserve(v, cap, *n + 1);
  memmove(*v + pos + 1, *v + pos, (*n - pos) * sizeof(uint64_t));
  (*v)[pos] = x;
  (*n)++;
  return 0;
}

/* src/lib.rs:142-174 KmerMinHash::new */
omh_t *omh_new(uint32_t num, uint32_t ksize, int is_protein, uint64_t seed,
               uint64_t max_hash, int track_abundance) {
  omh_t *mh = (omh_t *)calloc(1, sizeof(omh_t));
  mh->num = num; mh->ksize = ksize; mh->is_protein = is_protein ? 1 : 0;
  mh->seed = seed; mh->max_hash = max_hash;
  vec_reserve(&mh->mins, &mh->cap, num > 0 ? num : 1000);
  mh->has_abunds = track_abundance ? 1 : 0;
  if (track_abundance) vec_reserve(&mh->abunds, &mh->acap, mh->cap);
  return mh;
}

omh_t *omh_clone(const omh_t *src) {
  omh_t *mh = (omh_t *)calloc(1, sizeof(omh_t));
  *mh = *src;
  mh->mins = NULL; mh->cap = 0; mh->abunds = NULL; mh->acap = 0;
  vec_reserve(&mh->mins, &mh->cap, src->n ? src->n : 1);
  memcpy(mh->mins, src->mins, src->n * sizeof(uint64_t));
  if (src->has_abunds) {
    vec_reserve(&mh->abunds, &mh->acap, src->an ? src->an : 1);
    memcpy(mh->abunds, src->abunds, src->an * sizeof(uint64_t));
  }
  return mh;
}

void omh_free(omh_t *mh) {
  if (!mh) return;
  free(mh->mins);
  free(mh->abunds);
  free(mh);
}

/* src/lib.rs:176-190 check_compatible: ksize, is_protein, max_hash, seed -- NOT num */
int omh_check_compatible(const omh_t *a, const omh_t *b) {
  if (a->ksize != b->ksize) return OMH_MISMATCH_KSIZES;
  if (a->is_protein != b->is_protein) return OMH_MISMATCH_DNA_PROT;
  if (a->max_hash != b->max_hash) return OMH_MISMATCH_MAX_HASH;
  if (a->seed != b->seed) return OMH_MISMATCH_SEED;
  return OMH_OK;
}

/* slice::binary_search: position of an equal element, else the insertion point */
static size_t lower_bound(const uint64_t *v, size_t n, uint64_t x) {
  size_t lo = 0, hi = n;
  while (lo < hi) {
    size_t mid = lo + (hi - lo) / 2;
    if (v[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* src/lib.rs:192-245 add_hash (quirks Q3, Q4).  Returns OMH_PANIC where the
 * Rust would panic on an out-of-range abunds index (possible only after Q5). */
int omh_add_hash(omh_t *mh, uint64_t hash) {
  uint64_t current_max = mh->n ? mh->mins[mh->n - 1] : UINT64_MAX;
  if (!(hash <= mh->max_hash || mh->max_hash == 0)) return OMH_OK;
  if (mh->n == 0) {
    vec_push(&mh->mins, &mh->n, &mh->cap, hash);
    if (mh->has_abunds) vec_push(&mh->abunds, &mh->an, &mh->acap, 1);
    return OMH_OK;
  }
  if (hash <= mh->max_hash || current_max > hash || (uint32_t)mh->n < mh->num) {
    size_t pos = lower_bound(mh->mins, mh->n, hash);
    if (pos == mh->n) {
      vec_push(&mh->mins, &mh->n, &mh->cap, hash);
      if (mh->has_abunds) vec_push(&mh->abunds, &mh->an, &mh->acap, 1);
    } else if (mh->mins[pos] != hash) {
      vec_insert(&mh->mins, &mh->n, &mh->cap, pos, hash);
      if (mh->has_abunds && vec_insert(&mh->abunds, &mh->an, &mh->acap, pos, 1) != 0)
        return OMH_PANIC;
      if (mh->num != 0 && mh->n > (size_t)mh->num) {
        mh->n--;
        if (mh->has_abunds && mh->an > 0) mh->an--;
      }
    } else if (mh->has_abunds) {
      if (pos >= mh->an) return OMH_PANIC;
      mh->abunds[pos] += 1;
    }
  }
  return OMH_OK;
}

/* src/lib.rs:247-250 add_word */
int omh_add_word(omh_t *mh, const uint8_t *word, size_t len) {
  return omh_add_hash(mh, omh_hash_murmur(word, len, mh->seed));
}

/* src/lib.rs:677-689 revcomp: fresh heap buffer per call, as the reference's Vec */
static uint8_t *revcomp_alloc(const uint8_t *seq, size_t len) {
  uint8_t *rc = (uint8_t *)malloc(len ? len : 1);
  for (size_t i = 0; i < len; i++) {
    uint8_t c = seq[len - 1 - i], o;
    switch (c) {
      case 'A': case 'a': o = 'T'; break;
      case 'T': case 't': o = 'A'; break;
      case 'C': case 'c': o = 'G'; break;
      case 'G': case 'g': o = 'C'; break;
      default: o = c;
    }
    rc[i] = o;
  }
  return rc;
}

/* src/lib.rs:795-804 _checkdna */
static int checkdna(const uint8_t *s, size_t len) {
  for (size_t i = 0; i < len; i++) {
    switch (s[i]) {
      case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': break;
      default: return 0;
    }
  }
  return 1;
}

/* str::from_utf8 on a short chunk (src/lib.rs:270, 787): 1 if well-formed UTF-8 */
static int utf8_ok(const uint8_t *s, size_t n) {
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
    else return 0;
    if (i + need >= n) return 0;            /* truncated sequence */
    if (s[i + 1] < lo || s[i + 1] > hi) return 0;
    for (size_t j = 2; j <= need; j++)
      if (s[i + j] < 0x80 || s[i + j] > 0xBF) return 0;
    i += need + 1;
  }
  return 1;
}

/* src/lib.rs:691-777 CODONTABLE: the standard genetic code, index = 16*b0+4*b1+b2, T=0,C=1,A=2,G=3 */
static const char CODON_AA[65] =
    "FFLLSSSSYY**CC*W" "LLLLPPPPHHQQRRRR" "IIIMTTTTNNKKSSRR" "VVVVAAAADDEEGGGG";

static int base_idx(uint8_t c) {
  switch (c) { case 'T': return 0; case 'C': return 1; case 'A': return 2; case 'G': return 3; default: return -1; }
}

/* src/lib.rs:779-793 to_aa: stop at the first incomplete codon, DROP codons that are not
 * in the table (Q8).  Returns -1 where from_utf8(chunk).unwrap() would panic. */
static long to_aa(const uint8_t *seq, size_t len, uint8_t *out) {
  size_t n = 0;
  for (size_t i = 0; i + 3 <= len; i += 3) {
    if (!utf8_ok(seq + i, 3)) return -1;
    int a = base_idx(seq[i]), b = base_idx(seq[i + 1]), c = base_idx(seq[i + 2]);
    if (a < 0 || b < 0 || c < 0) continue;
    out[n++] = (uint8_t)CODON_AA[16 * a + 4 * b + c];
  }
  return (long)n;
}

/* one translated frame, for tests: frame in 0..2, rc selects the reverse complement */
void omh_translate_frames(const uint8_t *seq, size_t len, int frame, int rc,
                          uint8_t *out, size_t *outlen) {
  uint8_t *up = (uint8_t *)malloc(len ? len : 1);
  for (size_t i = 0; i < len; i++) up[i] = (seq[i] >= 'a' && seq[i] <= 'z') ? seq[i] - 32 : seq[i];
  uint8_t *src = up;
  uint8_t *r = NULL;
  if (rc) { r = revcomp_alloc(up, len); src = r; }
  long n = ((size_t)frame <= len) ? to_aa(src + frame, len - frame, out) : 0;
  *outlen = n < 0 ? 0 : (size_t)n;
  free(r);
  free(up);
}

/* src/lib.rs:252-305 add_sequence (quirks Q1, Q2, Q8) */
int omh_add_sequence(omh_t *mh, const uint8_t *seq, size_t len, int force,
                     char *errbuf, size_t errcap) {
  int status = OMH_OK;
  size_t k = mh->ksize;
  /* 253-256: uppercase copy of the whole input (ASCII a-z only) */
  uint8_t *sequence = (uint8_t *)malloc(len ? len : 1);
  for (size_t i = 0; i < len; i++)
    sequence[i] = (seq[i] >= 'a' && seq[i] <= 'z') ? (uint8_t)(seq[i] - 32) : seq[i];
  if (len < k) { free(sequence); return OMH_OK; }          /* 257 */
  if (!mh->is_protein) {
    if (k == 0) { free(sequence); return OMH_PANIC; }        /* windows(0) panics */
    for (size_t i = 0; i + k <= len; i++) {                  /* 260 windows(ksize) */
      const uint8_t *kmer = sequence + i;
      if (checkdna(kmer, k)) {
        uint8_t *rc = revcomp_alloc(kmer, k);                /* 262 */
        int st = (memcmp(kmer, rc, k) < 0) ? omh_add_word(mh, kmer, k)   /* 263-267 */
                                            : omh_add_word(mh, rc, k);
        free(rc);
        if (st != OMH_OK) { status = st; break; }
      } else if (!force) {                                   /* 268-273 */
        if (!utf8_ok(kmer, k)) { status = OMH_PANIC; break; }
        if (errbuf && errcap) {
          size_t m = k < errcap - 1 ? k : errcap - 1;
          memcpy(errbuf, kmer, m);
          errbuf[m] = 0;
        }
        status = OMH_INVALID_DNA;
        break;
      }
    }
  } else {
    /* 277-301: six-frame translation, every window hashed, no validation */
    uint8_t *rc = revcomp_alloc(sequence, len);
    size_t aa_k = k / 3;
    uint8_t *aa = (uint8_t *)malloc(len / 3 + 1);
    for (size_t i = 0; i < 3 && status == OMH_OK; i++) {
      for (int strand = 0; strand < 2 && status == OMH_OK; strand++) {
        const uint8_t *src = strand ? rc : sequence;
        /* 281-286 / 293-294: fresh copy of the suffix, as the reference collects one */
        uint8_t *substr = (uint8_t *)malloc(len - i ? len - i : 1);
        memcpy(substr, src + i, len - i);
        long n = to_aa(substr, len - i, aa);
        free(substr);
        if (n < 0 || aa_k == 0) { status = OMH_PANIC; break; }
        for (size_t w = 0; w + aa_k <= (size_t)n; w++) {
          int st = omh_add_word(mh, aa + w, aa_k);
          if (st != OMH_OK) { status = st; break; }
        }
      }
    }
    free(aa);
    free(rc);
  }
  free(sequence);
  return status;
}

/* src/lib.rs:412-417 add_many, 405-410 add_from */
int omh_add_many(omh_t *mh, const uint64_t *hashes, size_t n) {
  for (size_t i = 0; i < n; i++) {
    int st = omh_add_hash(mh, hashes[i]);
    if (st != OMH_OK) return st;
  }
  return OMH_OK;
}
/* src/lib.rs:419-426 add_many_with_abund: item.0 added item.1 times, literally */
int omh_add_many_with_abund(omh_t *mh, const uint64_t *hashes, const uint64_t *abunds, size_t n) {
  for (size_t i = 0; i < n; i++)
    for (uint64_t r = 0; r < abunds[i]; r++) {
      int st = omh_add_hash(mh, hashes[i]);
      if (st != OMH_OK) return st;
    }
  return OMH_OK;
}
int omh_add_from(omh_t *mh, const omh_t *other) {
  return omh_add_many(mh, other->mins, other->n);
}

/* src/lib.rs:307-403 merge (quirks Q5, Q6): two fresh vectors, two-pointer union */
int omh_merge(omh_t *mh, const omh_t *other) {
  int st = omh_check_compatible(mh, other);
  if (st != OMH_OK) return st;
  size_t max_size = mh->n + other->n;
  uint64_t *merged = (uint64_t *)malloc((max_size ? max_size : 1) * sizeof(uint64_t));
  /* the abundance vectors may be LONGER than the mins (Q5: a merge never truncates them), and
   * every push below consumes at least one entry of one of them */
  size_t max_ab = mh->an + other->an;
  uint64_t *mab = (uint64_t *)malloc((max_ab ? max_ab : 1) * sizeof(uint64_t));
  size_t mn = 0, man = 0;
  size_t si = 0, oi = 0;       /* positions in the two mins */
  size_t sai = 0, oai = 0;     /* positions in the two abundance iterators */
  int s_has = mh->has_abunds, o_has = other->has_abunds;
  /* 331-381 */
  while (si < mh->n) {
    uint64_t value = mh->mins[si];
    if (oi >= other->n) {
      /* 336-343: push value, extend with the rest of self (si is NOT past value yet) */
      while (si < mh->n) merged[mn++] = mh->mins[si++];
      if (s_has) while (sai < mh->an) mab[man++] = mh->abunds[sai++];
      break;
    }
    uint64_t x = other->mins[oi];
    if (x < value) {
      merged[mn++] = x; oi++;
      if (o_has && oai < other->an) mab[man++] = other->abunds[oai++];
    } else if (x == value) {
      merged[mn++] = x; oi++; si++;
      if (o_has && oai < other->an) {
        uint64_t v = other->abunds[oai++];
        if (s_has && sai < mh->an) mab[man++] = v + mh->abunds[sai++];
      }
    } else {
      merged[mn++] = value; si++;
      if (s_has && sai < mh->an) mab[man++] = mh->abunds[sai++];
    }
  }
  /* 382-388 */
  while (oi < other->n) merged[mn++] = other->mins[oi++];
  if (o_has) while (oai < other->an) mab[man++] = other->abunds[oai++];
  /* 391-401 */
  size_t keep = mn;
  if (!(mn < (size_t)mh->num || mh->num == 0)) keep = mh->num;
  free(mh->mins);
  mh->mins = merged; mh->n = keep; mh->cap = max_size ? max_size : 1;
  free(mh->abunds);
  mh->abunds = mab; mh->an = man; mh->acap = max_ab ? max_ab : 1;
  mh->has_abunds = 1;          /* Q5: always Some(..) afterwards, never truncated */
  return OMH_OK;
}

/* src/lib.rs:515-544 Intersection iterator, counted */
static uint64_t intersect_count(const uint64_t *a, size_t na, const uint64_t *b, size_t nb,
                                uint64_t *collect) {
  size_t i = 0, j = 0; uint64_t c = 0;
  while (i < na && j < nb) {
    if (a[i] < b[j]) i++;
    else if (a[i] > b[j]) j++;
    else { if (collect) collect[c] = a[i]; c++; i++; j++; }
  }
  return c;
}

/* src/lib.rs:428-436 count_common: full mins, no truncation */
int omh_count_common(const omh_t *a, const omh_t *b, uint64_t *out) {
  int st = omh_check_compatible(a, b);
  if (st != OMH_OK) return st;
  *out = intersect_count(a->mins, a->n, b->mins, b->n, NULL);
  return OMH_OK;
}

/* src/lib.rs:470-499 intersection_size: combined = new(self params); merge self; merge other;
 * i1 = self ^ other; result = (|i1 ^ combined|, |combined|) */
int omh_intersection_size(const omh_t *a, const omh_t *b, uint64_t *common, uint64_t *size) {
  int st = omh_check_compatible(a, b);
  if (st != OMH_OK) return st;
  omh_t *comb = omh_new(a->num, a->ksize, a->is_protein, a->seed, a->max_hash, a->has_abunds);
  st = omh_merge(comb, a);
  if (st == OMH_OK) st = omh_merge(comb, b);
  if (st != OMH_OK) { omh_free(comb); return st; }
  size_t m = a->n < b->n ? a->n : b->n;
  uint64_t *i1 = (uint64_t *)malloc((m ? m : 1) * sizeof(uint64_t));
  uint64_t n1 = intersect_count(a->mins, a->n, b->mins, b->n, i1);
  *common = intersect_count(i1, (size_t)n1, comb->mins, comb->n, NULL);
  *size = comb->n;
  free(i1);
  omh_free(comb);
  return OMH_OK;
}

/* src/lib.rs:501-508 compare (Q7) */
int omh_compare(const omh_t *a, const omh_t *b, double *out) {
  int st = omh_check_compatible(a, b);
  if (st != OMH_OK) return st;
  uint64_t common, size;
  if (omh_intersection_size(a, b, &common, &size) == OMH_OK)
    *out = (double)common / (double)(size > 1 ? size : 1);
  else
    *out = 0.0;
  return OMH_OK;
}

/* src/index.rs:146-154 Leaf::containment: count_common / self.mins.len() (NaN when empty) */
int omh_containment(const omh_t *a, const omh_t *b, double *out) {
  uint64_t common;
  int st = omh_count_common(a, b, &common);
  if (st != OMH_OK) return st;
  *out = (double)common / (double)a->n;
  return OMH_OK;
}

size_t omh_size(const omh_t *mh) { return mh->n; }
const uint64_t *omh_mins(const omh_t *mh) { return mh->mins; }
int omh_has_abunds(const omh_t *mh) { return mh->has_abunds; }
size_t omh_abunds_size(const omh_t *mh) { return mh->has_abunds ? mh->an : 0; }
const uint64_t *omh_abunds(const omh_t *mh) { return mh->has_abunds ? mh->abunds : NULL; }
/* src/ffi.rs:143-150, 179-188: raw appends, no ordering check */
void omh_mins_push(omh_t *mh, uint64_t v) { vec_push(&mh->mins, &mh->n, &mh->cap, v); }
void omh_abunds_push(omh_t *mh, uint64_t v) {
  if (mh->has_abunds) vec_push(&mh->abunds, &mh->an, &mh->acap, v);
}

/* N x M block of the reference's compare/intersection_size, one faithful call per ordered
 * pair (rows are `self`).  Sketches are given as concatenated mins + offsets. */
int omh_compare_matrix(const uint64_t *mins, const uint64_t *offsets, size_t n_rows,
                       const uint64_t *cmins, const uint64_t *coffsets, size_t n_cols,
                       uint32_t num, uint32_t ksize, uint64_t max_hash,
                       uint64_t *common, uint64_t *size, double *jaccard) {
  omh_t **cols = (omh_t **)malloc((n_cols ? n_cols : 1) * sizeof(omh_t *));
  for (size_t j = 0; j < n_cols; j++) {
    cols[j] = omh_new(num, ksize, 0, 42, max_hash, 0);
    for (uint64_t t = coffsets[j]; t < coffsets[j + 1]; t++) omh_mins_push(cols[j], cmins[t]);
  }
  for (size_t i = 0; i < n_rows; i++) {
    omh_t *row = omh_new(num, ksize, 0, 42, max_hash, 0);
    for (uint64_t t = offsets[i]; t < offsets[i + 1]; t++) omh_mins_push(row, mins[t]);
    for (size_t j = 0; j < n_cols; j++) {
      uint64_t c = 0, s = 0;
      omh_intersection_size(row, cols[j], &c, &s);
      if (common) common[i * n_cols + j] = c;
      if (size) size[i * n_cols + j] = s;
      if (jaccard) jaccard[i * n_cols + j] = (double)c / (double)(s > 1 ? s : 1);
    }
    omh_free(row);
  }
  for (size_t j = 0; j < n_cols; j++) omh_free(cols[j]);
  free(cols);
  return OMH_OK;
}

/* ------------------------------------------------------------------ */
/* deterministic synthetic DNA: counter-based splitmix64, 32 bases per
 * 64-bit word, 2 bits per base LSB first, alphabet "ACGT"; every n_every-th base is 'N'. */
uint64_t osynth_splitmix64(uint64_t seed, uint64_t index) {
  uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

void osynth_dna(uint8_t *out, uint64_t start, uint64_t len, uint64_t seed, uint64_t n_every) {
  static const char alpha[4] = {'A', 'C', 'G', 'T'};
  uint64_t cur_word = UINT64_MAX, w = 0;
  for (uint64_t i = 0; i < len; i++) {
    uint64_t p = start + i;
    if ((p >> 5) != cur_word) { cur_word = p >> 5; w = osynth_splitmix64(seed, cur_word); }
    uint8_t c = (uint8_t)alpha[(w >> (2 * (p & 31))) & 3];
    if (n_every && (p % n_every) == n_every - 1) c = 'N';
    out[i] = c;
  }
}
