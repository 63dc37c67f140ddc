/* Calls every C symbol that the Rust shim (sourmash-rust_amd/rust/src/lib.rs) binds, in the order its
 * methods use them: hand the state over with the raw pushes, make the call, read the state back.
 * There is no Rust toolchain in the build image, so this program is what link-checks the shim's
 * forwards: same symbols, same argument types (tests/test_c_client.py also checks that the list of
 * symbols here covers the shim's extern block).  Host-only entry points are checked for their
 * results; the ones that need the device return code 2 without a GPU and real results with one. */
#include <stdio.h>
#include <string.h>
#include "sourmash_amd.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

static KmerMinHash *to_handle(uint32_t num, uint32_t k, const uint64_t *mins, size_t n, const uint64_t *ab) {
  KmerMinHash *h = kmerminhash_new(num, k, false, 42, 0, ab != NULL);
  for (size_t i = 0; i < n; i++) kmerminhash_mins_push(h, mins[i]);
  if (ab) for (size_t i = 0; i < n; i++) kmerminhash_abunds_push(h, ab[i]);
  return h;
}

int main(void) {
  sourmash_init();
  const int gpu = smh_device_available();
  const uint64_t m1[] = {2, 5, 9, 30}, a1[] = {1, 2, 1, 4}, m2[] = {5, 9, 11, 40};
  KmerMinHash *a = to_handle(4, 21, m1, 4, a1), *b = to_handle(4, 21, m2, 4, NULL), *c = to_handle(4, 31, m2, 4, NULL);

  /* check_compatible */
  CHECK(smh_check_compatible(a, b) == 0);
  CHECK(smh_check_compatible(a, c) == SOURMASH_ERROR_CODE_MISMATCH_K_SIZES);
  CHECK(sourmash_err_get_last_code() == SOURMASH_ERROR_CODE_MISMATCH_K_SIZES);
  SourmashStr msg = sourmash_err_get_last_message();
  CHECK(msg.len > 0);
  sourmash_str_free(&msg);
  sourmash_err_clear();

  /* add_hash, add_many, add_many_with_abund, add_from, merge + read_back */
  kmerminhash_add_hash(a, 7);
  CHECK(kmerminhash_get_mins_size(a) == 4);                 /* num = 4: 30 fell off */
  const uint64_t more[] = {1, 1, 3};
  CHECK(smh_add_many(a, more, 3) == 0);
  const uint64_t wh[] = {2, 100}, wa[] = {5, 9};
  CHECK(smh_add_many_with_abund(a, wh, wa, 2) == 0);
  const uint64_t *mm = kmerminhash_get_mins(a);
  const uint64_t *ma = kmerminhash_get_abunds(a);
  CHECK(kmerminhash_track_abundance(a) && kmerminhash_get_abunds_size(a) == 4);
  CHECK(mm[0] == 1 && mm[1] == 2 && mm[2] == 3 && mm[3] == 5);
  CHECK(ma[0] == 2 && ma[1] == 6 && ma[2] == 1);
  free((void *)mm); free((void *)ma);
  kmerminhash_add_from(b, a);
  kmerminhash_merge(b, a);
  CHECK(sourmash_err_get_last_code() == 0 && kmerminhash_get_mins_size(b) == 4);

  /* the calls that run on the device */
  uint64_t h = 0;
  const uint64_t off[] = {0, 3};
  int rc = smh_hash_words("ACG", off, 1, 42, &h);
  CHECK(gpu ? (rc == 0 && h == 1731421407650554201ULL) : rc == SOURMASH_ERROR_CODE_INTERNAL);
  const char *seq = "TGCCGCCCAGCACCGGGTGACTAGGTTGAGCCATGATTAACCTGCAATGA";
  rc = smh_add_sequence_len(a, seq, strlen(seq), true);
  CHECK(rc == (gpu ? 0 : SOURMASH_ERROR_CODE_INTERNAL));
  const uint64_t roff[] = {0, 25, 50};
  rc = smh_add_sequences(a, seq, roff, 2, true);
  CHECK(rc == (gpu ? 0 : SOURMASH_ERROR_CODE_INTERNAL));
  sourmash_err_clear();
  double j = kmerminhash_compare(a, b);
  uint64_t cc = kmerminhash_count_common(a, b);
  CHECK(gpu ? (j >= 0.0 && j <= 1.0 && cc <= 4) : sourmash_err_get_last_code() == SOURMASH_ERROR_CODE_INTERNAL);
  sourmash_err_clear();
  uint64_t *common = NULL, ncommon = 0, usize = 0;
  rc = smh_intersection(a, b, &common, &ncommon, &usize);
  if (gpu) { CHECK(rc == 0 && ncommon <= 4 && usize <= 4); free(common); } else CHECK(rc == SOURMASH_ERROR_CODE_INTERNAL);
  KmerMinHash *rows[] = {a}, *cols[] = {b};
  uint64_t bc = 0, bs = 0;
  double bj = 0;
  rc = smh_compare_block(rows, 1, cols, 1, &bj, &bc, &bs, NULL, NULL);
  if (gpu) CHECK(rc == 0 && bc == ncommon && bs == usize && bj == j); else CHECK(rc == SOURMASH_ERROR_CODE_INTERNAL);
  KmerMinHash *gs[] = {a, b};
  const uint32_t groups[] = {1, 0};
  rc = smh_add_sequences_grouped(gs, 2, seq, roff, groups, 2, true);
  CHECK(rc == (gpu ? 0 : SOURMASH_ERROR_CODE_INTERNAL));
  sourmash_err_clear();
  SmhIndex *idx = smh_index_new(gs, 2);
  if (gpu) {
    CHECK(idx != NULL && smh_index_len(idx) == 2);
    uint32_t hits[2], nh = 0;
    CHECK(smh_index_find(idx, a, 0.0, false, hits, &nh) == 0 && nh >= 1);
    smh_index_free(idx);
  } else {
    CHECK(idx == NULL && sourmash_err_get_last_code() == SOURMASH_ERROR_CODE_INTERNAL);
    smh_index_free(NULL);
    (void)smh_index_len; (void)smh_index_find;
  }
  sourmash_err_clear();

  /* the reference's own known answers through the legacy symbols, on the device (reference tests/minhash.rs:18-51: two
   * sketches fed two strings each, merged -> eight exact hashes; tests/minhash.rs:54-83: compare == 1.0 / >= 0.3;
   * tests/minhash.rs:5-16: 'R' is refused) */
  if (gpu) {
    KmerMinHash *x = kmerminhash_new(20, 10, false, 42, 0, false), *y = kmerminhash_new(20, 10, false, 42, 0, false);
    kmerminhash_add_sequence(x, "TGCCGCCCAGCA", false);
    kmerminhash_add_sequence(y, "TGCCGCCCAGCA", false);
    kmerminhash_add_sequence(x, "GTCCGCCCAGTGA", false);
    kmerminhash_add_sequence(y, "GTCCGCCCAGTGG", false);
    kmerminhash_merge(x, y);
    CHECK(sourmash_err_get_last_code() == 0);
    const uint64_t want[8] = {2996412506971915891ULL, 4448613756639084635ULL, 8373222269469409550ULL, 9390240264282449587ULL,
                              11085758717695534616ULL, 11668188995231815419ULL, 11760449009842383350ULL, 14682565545778736889ULL};
    CHECK(kmerminhash_get_mins_size(x) == 8);
    const uint64_t *got = kmerminhash_get_mins(x);
    for (int i = 0; i < 8; i++) CHECK(got[i] == want[i]);
    free((void *)got);
    kmerminhash_free(x); kmerminhash_free(y);
    const char *s1 = "TGCCGCCCAGCACCGGGTGACTAGGTTGAGCCATGATTAACCTGCAATGA", *s2 = "GATTGGTGCACACTTAACTGGGTGCCGCGCTGGTGCTGATCCATGAAGTT";
    x = kmerminhash_new(20, 10, false, 42, 0, false); y = kmerminhash_new(20, 10, false, 42, 0, false);
    kmerminhash_add_sequence(x, s1, false);
    kmerminhash_add_sequence(y, s1, false);
    CHECK(kmerminhash_compare(x, y) == 1.0 && kmerminhash_compare(y, x) == 1.0);
    kmerminhash_add_sequence(y, s1, false);
    CHECK(kmerminhash_compare(x, y) == 1.0 && kmerminhash_compare(y, x) == 1.0);
    kmerminhash_add_sequence(y, s2, false);
    CHECK(kmerminhash_compare(x, y) >= 0.3 && kmerminhash_compare(y, x) >= 0.3 && kmerminhash_compare(x, y) < 1.0);
    CHECK(sourmash_err_get_last_code() == 0);
    kmerminhash_free(x); kmerminhash_free(y);
    x = kmerminhash_new(1, 4, false, 42, 0, false);
    kmerminhash_add_sequence(x, "ATGR", false);
    CHECK(sourmash_err_get_last_code() == SOURMASH_ERROR_CODE_INVALID_D_N_A);
    sourmash_err_clear();
    CHECK(hash_murmur("ACG", 42) == 1731421407650554201ULL);       /* reference tests/test.rs:5 */
    kmerminhash_free(x);
  }
  kmerminhash_free(a); kmerminhash_free(b); kmerminhash_free(c);
  printf("c shim symbols ok (%s)\n", gpu ? "gpu" : "no gpu");
  return 0;
}
