/* Plain-C client of include/sourmash.h + include/sourmash_amd.h: proves the headers are valid C,
 * the library links, and the scalar (no-GPU) part of the ABI behaves like the reference.
 * Built and run by tests/test_c_client.py. */
#include <stdio.h>
#include <string.h>
#include "sourmash_amd.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void) {
  sourmash_init();
  KmerMinHash *a = kmerminhash_new(3, 21, false, 42, 0, true);
  CHECK(a != NULL);
  CHECK(kmerminhash_num(a) == 3 && kmerminhash_ksize(a) == 21 && !kmerminhash_is_protein(a));
  CHECK(kmerminhash_seed(a) == 42 && kmerminhash_max_hash(a) == 0 && kmerminhash_track_abundance(a));
  uint64_t hs[] = {5, 3, 9, 3, 1};
  for (int i = 0; i < 5; i++) kmerminhash_add_hash(a, hs[i]);
  CHECK(kmerminhash_get_mins_size(a) == 3);
  const uint64_t *m = kmerminhash_get_mins(a);
  const uint64_t *ab = kmerminhash_get_abunds(a);
  CHECK(m[0] == 1 && m[1] == 3 && m[2] == 5);
  CHECK(ab[0] == 1 && ab[1] == 2 && ab[2] == 1);
  free((void *)m); free((void *)ab);
  CHECK(kmerminhash_get_min_idx(a, 1) == 3 && kmerminhash_get_abund_idx(a, 1) == 2);

  KmerMinHash *b = kmerminhash_new(3, 31, false, 42, 0, false);
  kmerminhash_merge(a, b);
  CHECK(sourmash_err_get_last_code() == SOURMASH_ERROR_CODE_MISMATCH_K_SIZES);
  SourmashStr msg = sourmash_err_get_last_message();
  CHECK(msg.owned && msg.len == strlen("different ksizes cannot be compared"));
  CHECK(memcmp(msg.data, "different ksizes cannot be compared", msg.len) == 0);
  sourmash_str_free(&msg);
  CHECK(msg.data == NULL && !msg.owned);
  sourmash_err_clear();
  CHECK(sourmash_err_get_last_code() == SOURMASH_ERROR_CODE_NO_ERROR);

  Signature *s = signature_new();
  signature_set_name(s, "demo");
  signature_push_mh(s, a);
  SourmashStr js = signature_save_json(s);
  CHECK(js.len > 0 && memmem(js.data, js.len, "\"mins\":[1,3,5]", 14) != NULL);
  CHECK(memmem(js.data, js.len, "\"abundances\":[1,2,1]", 20) != NULL);
  uintptr_t n = 0;
  char buf[4096];
  snprintf(buf, sizeof buf, "[%.*s]", (int)js.len, js.data);
  Signature **loaded = signatures_load_buffer(buf, strlen(buf), false, 21, "DNA", &n);
  CHECK(loaded != NULL && n == 1 && signature_eq(loaded[0], s));
  sourmash_str_free(&js);

  if (!smh_device_available()) {
    kmerminhash_add_sequence(a, "ACGTACGTACGTACGTACGTACGTACGT", true);
    CHECK(sourmash_err_get_last_code() == SOURMASH_ERROR_CODE_INTERNAL);   /* no CPU fallback */
    sourmash_err_clear();
  } else {
    CHECK(hash_murmur("ACG", 42) == 1731421407650554201ULL);
  }
  signature_free(loaded[0]); free(loaded);
  signature_free(s);
  kmerminhash_free(a); kmerminhash_free(b); kmerminhash_free(NULL);
  printf("c abi client ok\n");
  return 0;
}
