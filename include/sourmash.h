/*
 * include/sourmash.h -- the drop-in C ABI of the MI355X implementation.
 *
 * This is the SAME surface as the reference's cbindgen-generated include/sourmash.h
 * (reference include/sourmash.h:1-185, generated from src/ffi.rs, src/utils.rs,
 * src/errors.rs): same symbol names, argument order, struct layouts and error-code
 * numbering, so a caller of the reference (Python sourmash through cffi, or C) links
 * against libsourmash_amd.so unchanged.  Each declaration cites the reference function it
 * replaces.  Additive, MI355X-specific entry points live in sourmash_amd.h.
 *
 * Error convention (reference src/utils.rs:18-45,154-166): functions marked [pad] store a
 * failure in a thread-local slot and return an all-zero value; poll
 * sourmash_err_get_last_code() and clear with sourmash_err_clear().  The slot is not
 * cleared by successful calls.
 */
#ifndef SOURMASH_H_INCLUDED
#define SOURMASH_H_INCLUDED

#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference src/errors.rs:28-50.  In C the enum tag and the typedef share the name exactly as in
 * the cbindgen header; C++ has one name space for both, so the tag gets a suffix there. */
#ifdef __cplusplus
enum SourmashErrorCodeValues {
#else
enum SourmashErrorCode {
#endif
  SOURMASH_ERROR_CODE_NO_ERROR = 0,
  SOURMASH_ERROR_CODE_PANIC = 1,
  SOURMASH_ERROR_CODE_INTERNAL = 2,
  SOURMASH_ERROR_CODE_MSG = 3,
  SOURMASH_ERROR_CODE_UNKNOWN = 4,
  SOURMASH_ERROR_CODE_MISMATCH_K_SIZES = 101,
  SOURMASH_ERROR_CODE_MISMATCH_D_N_A_PROT = 102,
  SOURMASH_ERROR_CODE_MISMATCH_MAX_HASH = 103,
  SOURMASH_ERROR_CODE_MISMATCH_SEED = 104,
  SOURMASH_ERROR_CODE_INVALID_D_N_A = 1101,
  SOURMASH_ERROR_CODE_INVALID_PROT = 1102,
  SOURMASH_ERROR_CODE_IO = 100001,
  SOURMASH_ERROR_CODE_UTF8_ERROR = 100002,
  SOURMASH_ERROR_CODE_PARSE_INT = 100003,
  SOURMASH_ERROR_CODE_SERDE_ERROR = 100004,
};
typedef uint32_t SourmashErrorCode;

typedef struct KmerMinHash KmerMinHash; /* reference src/lib.rs:37-46 */
typedef struct Signature Signature;     /* reference src/lib.rs:546-565 */

/* reference src/utils.rs:168-174; not NUL-terminated, use len */
typedef struct {
  char *data;
  uintptr_t len;
  bool owned;
} SourmashStr;

/* ---- hot path: hashing and sketching --------------------------------------------- */
uint64_t hash_murmur(const char *kmer, uint64_t seed);                       /* src/ffi.rs:15-24 */
KmerMinHash *kmerminhash_new(uint32_t n, uint32_t k, bool prot, uint64_t seed, uint64_t mx,
                             bool track_abundance);                          /* src/ffi.rs:26-43 */
void kmerminhash_free(KmerMinHash *ptr);                                     /* src/ffi.rs:45-53 */
void kmerminhash_add_sequence(KmerMinHash *ptr, const char *sequence, bool force); /* [pad] src/ffi.rs:55-70 */
void kmerminhash_add_hash(KmerMinHash *ptr, uint64_t h);                     /* src/ffi.rs:72-80 */
void kmerminhash_add_word(KmerMinHash *ptr, const char *word);               /* src/ffi.rs:82-95 */
void kmerminhash_add_from(KmerMinHash *ptr, const KmerMinHash *other);       /* [pad] src/ffi.rs:260-274 */
void kmerminhash_merge(KmerMinHash *ptr, const KmerMinHash *other);          /* [pad] src/ffi.rs:244-258 */

/* ---- hot path: comparing ----------------------------------------------------------- */
double kmerminhash_compare(KmerMinHash *ptr, const KmerMinHash *other);      /* [pad] src/ffi.rs:311-325 */
uint64_t kmerminhash_count_common(KmerMinHash *ptr, const KmerMinHash *other); /* [pad] src/ffi.rs:276-290 */
uint64_t kmerminhash_intersection(KmerMinHash *ptr, const KmerMinHash *other); /* [pad] src/ffi.rs:292-309: returns the union-sketch SIZE */

/* ---- accessors ------------------------------------------------------------------------ */
const uint64_t *kmerminhash_get_mins(KmerMinHash *ptr);      /* [pad] src/ffi.rs:97-107: fresh copy, caller-owned */
uintptr_t kmerminhash_get_mins_size(KmerMinHash *ptr);       /* src/ffi.rs:134-141 */
uint64_t kmerminhash_get_min_idx(KmerMinHash *ptr, uint64_t idx);   /* [pad] src/ffi.rs:124-132 */
void kmerminhash_mins_push(KmerMinHash *ptr, uint64_t val);  /* src/ffi.rs:143-150: raw append */
const uint64_t *kmerminhash_get_abunds(KmerMinHash *ptr);    /* [pad] src/ffi.rs:109-122: NULL when untracked */
uintptr_t kmerminhash_get_abunds_size(KmerMinHash *ptr);     /* src/ffi.rs:166-177 */
uint64_t kmerminhash_get_abund_idx(KmerMinHash *ptr, uint64_t idx); /* [pad] src/ffi.rs:152-164 */
void kmerminhash_abunds_push(KmerMinHash *ptr, uint64_t val);/* src/ffi.rs:179-188 */
bool kmerminhash_is_protein(KmerMinHash *ptr);               /* src/ffi.rs:190-197 */
uint64_t kmerminhash_seed(KmerMinHash *ptr);                 /* src/ffi.rs:199-206 */
bool kmerminhash_track_abundance(KmerMinHash *ptr);          /* src/ffi.rs:208-215 */
uint32_t kmerminhash_num(KmerMinHash *ptr);                  /* src/ffi.rs:217-224 */
uint32_t kmerminhash_ksize(KmerMinHash *ptr);                /* src/ffi.rs:226-233 */
uint64_t kmerminhash_max_hash(KmerMinHash *ptr);             /* src/ffi.rs:235-242 */

/* ---- Signature container (host only) ------------------------------------------------ */
Signature *signature_new(void);                                      /* src/ffi.rs:329-332 */
void signature_free(Signature *ptr);                                 /* src/ffi.rs:334-342 */
void signature_set_name(Signature *ptr, const char *name);           /* [pad] src/ffi.rs:344-362 */
void signature_set_filename(Signature *ptr, const char *name);       /* [pad] src/ffi.rs:364-382 */
void signature_push_mh(Signature *ptr, const KmerMinHash *other);    /* [pad] src/ffi.rs:384-399 */
void signature_set_mh(Signature *ptr, const KmerMinHash *other);     /* [pad] src/ffi.rs:401-416 */
SourmashStr signature_get_name(Signature *ptr);                      /* [pad] src/ffi.rs:418-431 */
SourmashStr signature_get_filename(Signature *ptr);                  /* [pad] src/ffi.rs:433-446 */
SourmashStr signature_get_license(Signature *ptr);                   /* [pad] src/ffi.rs:448-457 */
KmerMinHash *signature_first_mh(Signature *ptr);                     /* [pad] src/ffi.rs:459-473 */
bool signature_eq(Signature *ptr, Signature *other);                 /* [pad] src/ffi.rs:475-489 */
SourmashStr signature_save_json(Signature *ptr);                     /* [pad] src/ffi.rs:491-501 */
KmerMinHash **signature_get_mhs(Signature *ptr, uintptr_t *size);    /* [pad] src/ffi.rs:503-521 */
SourmashStr signatures_save_buffer(Signature **ptr, uintptr_t size); /* [pad] src/ffi.rs:523-534 */
Signature **signatures_load_path(const char *ptr, bool ignore_md5sum, uintptr_t ksize,
                                 const char *select_moltype, uintptr_t *size); /* [pad] src/ffi.rs:536-569 */
Signature **signatures_load_buffer(const char *ptr, uintptr_t insize, bool ignore_md5sum,
                                   uintptr_t ksize, const char *select_moltype,
                                   uintptr_t *size);                 /* [pad] src/ffi.rs:570-604 */

/* ---- errors and strings --------------------------------------------------------------- */
void sourmash_err_clear(void);                          /* src/utils.rs:93-98 */
SourmashStr sourmash_err_get_backtrace(void);           /* src/utils.rs:72-90 */
SourmashErrorCode sourmash_err_get_last_code(void);     /* src/utils.rs:106-118 */
SourmashStr sourmash_err_get_last_message(void);        /* src/utils.rs:52-69 */
void sourmash_init(void);                               /* src/utils.rs:100-104 */
void sourmash_str_free(SourmashStr *s);                 /* src/utils.rs:236-245 */
SourmashStr sourmash_str_from_cstr(const char *s);      /* [pad] src/utils.rs:220-234 */

#ifdef __cplusplus
}
#endif
#endif /* SOURMASH_H_INCLUDED */
