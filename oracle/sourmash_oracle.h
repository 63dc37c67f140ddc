/*
 * oracle/sourmash_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the MinHash hot path of luizirber/sourmash-rust.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or call this.  The product library (sourmash-rust_amd/) never does.
 *
 * Every function cites the reference lines (relative to the reference root)
 * it restates.  Pinned by the reference's own known-answer tests -- see the
 * header of sourmash_oracle.c.
 */
#ifndef SOURMASH_ORACLE_H
#define SOURMASH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes = SourmashErrorCode values, src/errors.rs:28-50 */
enum {
  OMH_OK = 0,
  OMH_PANIC = 1,
  OMH_MISMATCH_KSIZES = 101,
  OMH_MISMATCH_DNA_PROT = 102,
  OMH_MISMATCH_MAX_HASH = 103,
  OMH_MISMATCH_SEED = 104,
  OMH_INVALID_DNA = 1101
};

/* src/lib.rs:37-46 -- `abunds: Option<Vec<u64>>` keeps its own length (quirk Q5) */
typedef struct omh {
  uint32_t num;
  uint32_t ksize;
  int is_protein;
  uint64_t seed;
  uint64_t max_hash;
  uint64_t *mins;
  size_t n, cap;
  int has_abunds;
  uint64_t *abunds;
  size_t an, acap;
} omh_t;

uint64_t omh_hash_murmur(const uint8_t *key, size_t len, uint64_t seed);
void omh_murmur3_x64_128(const uint8_t *key, size_t len, uint64_t seed, uint64_t out[2]);

omh_t *omh_new(uint32_t num, uint32_t ksize, int is_protein, uint64_t seed,
               uint64_t max_hash, int track_abundance);
omh_t *omh_clone(const omh_t *src);
void omh_free(omh_t *mh);

int omh_check_compatible(const omh_t *a, const omh_t *b);
int omh_add_hash(omh_t *mh, uint64_t h);
int omh_add_word(omh_t *mh, const uint8_t *word, size_t len);
int omh_add_sequence(omh_t *mh, const uint8_t *seq, size_t len, int force,
                     char *errbuf, size_t errcap);
int omh_add_many(omh_t *mh, const uint64_t *hashes, size_t n);
int omh_add_many_with_abund(omh_t *mh, const uint64_t *hashes, const uint64_t *abunds, size_t n);
int omh_add_from(omh_t *mh, const omh_t *other);
int omh_merge(omh_t *mh, const omh_t *other);
int omh_count_common(const omh_t *a, const omh_t *b, uint64_t *out);
int omh_intersection_size(const omh_t *a, const omh_t *b, uint64_t *common, uint64_t *size);
int omh_compare(const omh_t *a, const omh_t *b, double *out);
int omh_containment(const omh_t *a, const omh_t *b, double *out);

size_t omh_size(const omh_t *mh);
const uint64_t *omh_mins(const omh_t *mh);
int omh_has_abunds(const omh_t *mh);
size_t omh_abunds_size(const omh_t *mh);
const uint64_t *omh_abunds(const omh_t *mh);
void omh_mins_push(omh_t *mh, uint64_t v);
void omh_abunds_push(omh_t *mh, uint64_t v);

/* helpers over flat arrays (tests and the timed CPU baseline) */
void omh_translate_frames(const uint8_t *seq, size_t len, int frame, int rc,
                          uint8_t *out, size_t *outlen);
int omh_compare_matrix(const uint64_t *mins, const uint64_t *offsets, size_t n_rows,
                       const uint64_t *cmins, const uint64_t *coffsets, size_t n_cols,
                       uint32_t num, uint32_t ksize, uint64_t max_hash,
                       uint64_t *common, uint64_t *size, double *jaccard);

/* deterministic synthetic inputs */
uint64_t osynth_splitmix64(uint64_t seed, uint64_t index);
void osynth_dna(uint8_t *out, uint64_t start, uint64_t len, uint64_t seed, uint64_t n_every);

#ifdef __cplusplus
}
#endif
#endif
