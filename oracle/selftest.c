/* oracle/selftest.c -- TEST INFRASTRUCTURE.  The reference's known-answer tests run natively
 * against the C oracle (built with sanitizers by tests/test_c_client.py). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sourmash_oracle.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void) {
  /* reference tests/test.rs:5 */
  CHECK(omh_hash_murmur((const uint8_t *)"ACG", 3, 42) == 1731421407650554201ULL);
  /* reference tests/minhash.rs:5-17 */
  char err[64];
  omh_t *e = omh_new(1, 4, 0, 42, 0, 0);
  CHECK(omh_add_sequence(e, (const uint8_t *)"ATGR", 4, 0, err, sizeof err) == OMH_INVALID_DNA && !strcmp(err, "ATGR"));
  omh_free(e);
  /* reference tests/minhash.rs:19-52 */
  static const uint64_t expect[8] = {2996412506971915891ULL, 4448613756639084635ULL, 8373222269469409550ULL,
                                     9390240264282449587ULL, 11085758717695534616ULL, 11668188995231815419ULL,
                                     11760449009842383350ULL, 14682565545778736889ULL};
  omh_t *a = omh_new(20, 10, 0, 42, 0, 0), *b = omh_new(20, 10, 0, 42, 0, 0);
  omh_add_sequence(a, (const uint8_t *)"TGCCGCCCAGCA", 12, 0, NULL, 0);
  omh_add_sequence(b, (const uint8_t *)"TGCCGCCCAGCA", 12, 0, NULL, 0);
  omh_add_sequence(a, (const uint8_t *)"GTCCGCCCAGTGA", 13, 0, NULL, 0);
  omh_add_sequence(b, (const uint8_t *)"GTCCGCCCAGTGG", 13, 0, NULL, 0);
  CHECK(omh_merge(a, b) == OMH_OK && omh_size(a) == 8 && !memcmp(omh_mins(a), expect, sizeof expect));
  /* reference tests/minhash.rs:54-83 */
  const char *s1 = "TGCCGCCCAGCACCGGGTGACTAGGTTGAGCCATGATTAACCTGCAATGA", *s2 = "GATTGGTGCACACTTAACTGGGTGCCGCGCTGGTGCTGATCCATGAAGTT";
  omh_t *c = omh_new(20, 10, 0, 42, 0, 0), *d = omh_new(20, 10, 0, 42, 0, 0);
  double j = 0;
  omh_add_sequence(c, (const uint8_t *)s1, 50, 0, NULL, 0);
  omh_add_sequence(d, (const uint8_t *)s1, 50, 0, NULL, 0);
  CHECK(omh_compare(c, d, &j) == OMH_OK && j == 1.0);
  omh_add_sequence(d, (const uint8_t *)s2, 50, 0, NULL, 0);
  CHECK(omh_compare(c, d, &j) == OMH_OK && j >= 0.3 && omh_compare(d, c, &j) == OMH_OK && j >= 0.3);
  /* protein arm, abundance, synthetic generator, scaled mode: exercise under the sanitizers */
  uint8_t *seq = (uint8_t *)malloc(50000);
  osynth_dna(seq, 0, 50000, 7, 997);
  omh_t *p = omh_new(0, 27, 1, 42, (uint64_t)1 << 60, 1), *q = omh_new(100, 31, 0, 42, 0, 1);
  CHECK(omh_add_sequence(p, seq, 50000, 1, NULL, 0) == OMH_OK && omh_size(p) > 0 && omh_abunds_size(p) == omh_size(p));
  CHECK(omh_add_sequence(q, seq, 50000, 1, NULL, 0) == OMH_OK && omh_size(q) == 100);
  uint64_t cc = 0;
  CHECK(omh_count_common(q, q, &cc) == OMH_OK && cc == 100);
  free(seq);
  /* Q5: after a merge of full bottom-num sketches the abundance vector is longer than the mins
   * (never truncated); merging such a sketch again, and compare() (which merges clones), must
   * size their buffers from the abundance lengths */
  {
    omh_t *m1 = omh_new(50, 21, 0, 42, 0, 1), *m2 = omh_new(50, 21, 0, 42, 0, 1);
    uint64_t x = 88172645463325252ULL;
    for (int i = 0; i < 400; i++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      if (i < 300) omh_add_hash(m1, x >> 3);
      if (i >= 100) omh_add_hash(m2, x >> 3);
    }
    CHECK(omh_merge(m2, m1) == OMH_OK && omh_size(m2) == 50 && omh_abunds_size(m2) > 50);
    CHECK(omh_merge(m2, m1) == OMH_OK && omh_merge(m1, m2) == OMH_OK);
    CHECK(omh_compare(m1, m2, &j) == OMH_OK && omh_compare(m2, m1, &j) == OMH_OK);
    omh_free(m1); omh_free(m2);
  }
  omh_free(a); omh_free(b); omh_free(c); omh_free(d); omh_free(p); omh_free(q);
  printf("oracle selftest ok\n");
  return 0;
}
