/*
 * include/sourmash_amd.h -- ADDITIVE entry points of the MI355X implementation.
 *
 * The reference ABI (sourmash.h) hands the library one NUL-terminated C string and one pair of
 * sketches per call.  These symbols add what a GPU needs: explicit lengths, many records per
 * call, device-resident buffers, and an N x M compare block.  None of them changes or shadows
 * a reference symbol.  Plain pointers and sizes only; `stream` is a hipStream_t passed as
 * void*.  NULL = the library's own stream, which is first ordered after everything already queued
 * on the legacy default stream (where a caller without streams of its own produced the inputs);
 * every entry point returns with its device work complete -- with two stated exceptions,
 * smh_collection_begin with world == 1 and smh_collection_finish without a gathered buffer (a single
 * owner's dictionary, which only later calls of this library use): their work is left queued, and
 * EVERY later entry point, on whatever stream it is given, is first ordered behind it (an event
 * recorded behind the open work; the library's shared scratch buffers are never rewritten under it).
 * "dev" pointers are HIP device pointers.
 *
 * Error convention: same thread-local slot as sourmash.h; functions returning int return 0 on
 * success and the SourmashErrorCode otherwise.
 */
#ifndef SOURMASH_AMD_H_INCLUDED
#define SOURMASH_AMD_H_INCLUDED

#include "sourmash.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1 when a HIP device can be used, 0 otherwise (never raises). */
int smh_device_available(void);
/* device ordinal in use and its compute-unit count (0 on failure, error slot set) */
int smh_device_info(int *device, int *compute_units);

/* KmerMinHash::add_sequence (reference src/lib.rs:252-305) with an explicit length: the bytes
 * may contain NUL.  Same semantics and errors as kmerminhash_add_sequence. */
int smh_add_sequence_len(KmerMinHash *ptr, const char *seq, uint64_t len, bool force);

/* Many records in one call, as if add_sequence were called on each in order
 * (offsets: n_records+1 host entries into seq).  Every record is processed; the first record
 * that the reference would have failed on is the reported error.  *_dev: seq is a device
 * pointer (inputs already resident in HBM). */
int smh_add_sequences(KmerMinHash *ptr, const char *seq, const uint64_t *offsets,
                      uint32_t n_records, bool force);
int smh_add_sequences_dev(KmerMinHash *ptr, const void *seq_dev, uint64_t total_len,
                          const uint64_t *offsets, uint32_t n_records, bool force, void *stream);

/* Many sketches from one batch: record r feeds sketches[groups[r]] (one genome = one group of
 * contigs; the loop of reference src/lib.rs:252-305 callers that build one signature per input
 * file).  Per sketch the result is that of smh_add_sequences over its records in order.  Sketches
 * of one molecule type with equal (ksize, seed) share ONE hashing launch and ONE sort when they
 * are all scaled with one max_hash, or (DNA) all bottom-num; other
 * parameter combinations are served sketch by sketch. */
int smh_add_sequences_grouped(KmerMinHash *const *sketches, uint32_t n_sketches, const char *seq,
                              const uint64_t *offsets, const uint32_t *groups, uint32_t n_records, bool force);
int smh_add_sequences_grouped_dev(KmerMinHash *const *sketches, uint32_t n_sketches, const void *seq_dev,
                                  uint64_t total_len, const uint64_t *offsets, const uint32_t *groups,
                                  uint32_t n_records, bool force, void *stream);

/* add_hash over an array (reference src/lib.rs:412-417 add_many) */
int smh_add_many(KmerMinHash *ptr, const uint64_t *hashes, uint64_t n);

/* KmerMinHash::add_many_with_abund (reference src/lib.rs:419-426; Rust API only, the reference header
 * has no symbol for it): item i is the pair (hashes[i], abunds[i]) and is added abunds[i] times. */
int smh_add_many_with_abund(KmerMinHash *ptr, const uint64_t *hashes, const uint64_t *abunds, uint64_t n);

/* KmerMinHash::check_compatible (reference src/lib.rs:176-190; Rust API only): 0, or the mismatch code
 * (101 ksize, 102 DNA/protein, 103 max_hash, 104 seed) with the error slot set. */
int smh_check_compatible(const KmerMinHash *ptr, const KmerMinHash *other);

/* KmerMinHash::intersection (reference src/lib.rs:438-468; the reference header only exports its size):
 * *common_out receives a malloc'ed array (free() it) of the hashes in both sketches that lie inside the
 * bottom-`num` of the union, *n_common their number, *union_size the size of the combined sketch. */
int smh_intersection(const KmerMinHash *ptr, const KmerMinHash *other, uint64_t **common_out, uint64_t *n_common,
                     uint64_t *union_size);

/* The union of partial SCALED sketches without leaving HBM -- what folds the per-GPU partial sketches of one input into one
 * signature (KmerMinHash::merge, reference src/lib.rs:307-403, for scaled sketches: set union, abundances add).
 * smh_sketch_export_dev copies the sketch's ascending hashes (and, when it tracks them and abunds_dev is not NULL, their
 * abundances) into the caller's device buffers of `capacity` entries; *n_out = the number of hashes (call with mins_dev NULL
 * to ask; with a buffer, capacity < *n_out is an error -- Internal -- and nothing is written).  Side effect: a sketch whose
 * state is on the host is MOVED to HBM by the call (its host vectors are emptied; accessors bring it back on demand); a sketch
 * whose abundance vector does not match its hashes (quirks Q5/Q6 after a merge) has no device form and is refused.
 * smh_sketch_absorb_dev unites `ptr` with n_parts sorted, distinct parts lying in ONE device buffer (part k =
 * mins_dev[part_starts[k] .. + part_lens[k]), e.g. the output of an all-gather of padded exports): each part is merged by rank
 * arithmetic and two scatters, no sort, no host copy.  A sketch that tracks abundances needs abunds_dev. */
int smh_sketch_export_dev(KmerMinHash *ptr, uint64_t *mins_dev, uint64_t *abunds_dev, uint64_t capacity, uint64_t *n_out,
                          void *stream);
int smh_sketch_absorb_dev(KmerMinHash *ptr, const uint64_t *mins_dev, const uint64_t *abunds_dev, const uint64_t *part_starts,
                          const uint64_t *part_lens, uint32_t n_parts, void *stream);

/* murmur64 of n byte strings (offsets: n+1 host entries) on the device
 * (reference src/lib.rs:33-35 _hash_murmur) */
int smh_hash_words(const char *bytes, const uint64_t *offsets, uint32_t n, uint64_t seed,
                   uint64_t *out);

/* rows x cols block of compare / intersection_size / count_common / containment between
 * host sketches (reference src/lib.rs:428-436,470-508, src/index.rs:146-154).  Row i is `self`,
 * so its `num` truncates the union walk.  Outputs are row-major n_rows*n_cols, any may be NULL.
 * check_compatible (reference src/lib.rs:176-190) is applied to every pair first. */
int smh_compare_block(KmerMinHash *const *rows, uint32_t n_rows, KmerMinHash *const *cols,
                      uint32_t n_cols, double *jaccard, uint64_t *common, uint64_t *size,
                      uint64_t *count_common, double *containment);

/* The same on device-resident sketches in CSR form: hashes_dev[offsets[i]..offsets[i+1]) is
 * sketch i, ascending and distinct; offsets are HOST arrays.  All sketches share ksize / seed /
 * max_hash / molecule (the caller's index guarantees it); `num` is the rows' num (0 = scaled).
 * Output pointers are device pointers, row-major, any may be NULL. */
int smh_compare_block_dev(const uint64_t *row_hashes_dev, const uint64_t *row_offsets, uint32_t n_rows,
                          const uint64_t *col_hashes_dev, const uint64_t *col_offsets, uint32_t n_cols,
                          uint32_t num, double *jaccard_dev, uint64_t *common_dev, uint64_t *size_dev,
                          uint64_t *count_common_dev, double *containment_dev, void *stream);

/* The all-vs-all matrix of ONE collection resident in HBM, optionally computed by `world` cooperating ranks (one process
 * per GPU) that each hold the whole collection (after an all-gather of the signatures):
 *   1. smh_collection_begin   rank g sorts slice g of hash space (1/world of the pooled hashes: the dictionary pre-pass
 *                             is sharded, not replicated) and leaves its findings in a "share" of smh_collection_share_bytes()
 *                             bytes at smh_collection_share() (device memory, same size on every rank);
 *   2. the caller all-gathers the shares (RCCL; world == 1: nothing to do);
 *   3. smh_collection_finish  assembles the dictionary from the gathered shares (world x share_bytes, rank-major; NULL when
 *                             world == 1): dense ranks of every hash, connected components, frequent hashes, range tables;
 *   4. smh_collection_compare rows [row_lo, row_hi) x ALL columns, outputs row-major (row_hi - row_lo) x n in device memory
 *                             (any may be NULL).  Every pair equals KmerMinHash::compare / count_common of the two sketches
 *                             (reference src/lib.rs:428-436, 470-508) with the one `num` given.  ownership:
 *        0  every pair of the block is computed here;
 *        1  the block is the whole matrix: upper triangle + mirrors (the walk is symmetric when there is one num);
 *        2  the block is this rank's share of a matrix the ranks compute together: row i OWNS the pairs (i, j) with
 *           (j - i) mod n < n/2 (ties: i < j) -- every unordered pair has one owner, every row owns n/2 pairs.  Owned
 *           pairs, pairs whose column is one of the block's own rows, and pairs that share no (non-frequent) hash are
 *           final when the call returns; the others must be taken from their owner's rank, transposed
 *           (sourmash-rust_amd/distributed.py does exactly that with one all-to-all).
 * The dictionary can serve any number of compare calls.  offsets: n+1 HOST entries, sketch i = hashes_dev[offsets[i] ..
 * offsets[i+1]) ascending and distinct; hashes_dev must stay valid until smh_collection_free. */
typedef struct SmhCollection SmhCollection;
SmhCollection *smh_collection_begin(const uint64_t *hashes_dev, const uint64_t *offsets, uint32_t n, uint32_t world,
                                    uint32_t rank, void *stream);
uint64_t smh_collection_share_bytes(const SmhCollection *collection);
const void *smh_collection_share(const SmhCollection *collection);
/* the share copied to dst_dev (e.g. this rank's slot of the all-gather's output buffer) */
int smh_collection_share_to(const SmhCollection *collection, void *dst_dev, void *stream);
int smh_collection_finish(SmhCollection *collection, const void *gathered_dev, void *stream);
int smh_collection_compare(SmhCollection *collection, uint32_t row_lo, uint32_t row_hi, uint32_t num, uint32_t ownership,
                           double *jaccard_dev, uint64_t *common_dev, uint64_t *size_dev, uint64_t *count_common_dev,
                           double *containment_dev, void *stream);
void smh_collection_free(SmhCollection *collection);
/* The exchange that completes ownership 2, device side (8-byte outputs: jaccard, common, size, count_common).
 * smh_mirror_pack: for each of n_blocks peers holding rows [col_lo[b], col_hi[b]), this rank's block out_dev (n_local x
 * n_total) restricted to those columns, TRANSPOSED, packed one block after the other into packed_dev -- the send buffer of
 * an all-to-all.  smh_mirror_apply: from the blocks received (peer b holds rows [peer_lo[b], peer_hi[b]); block b is
 * n_local x (peer_hi[b] - peer_lo[b]) row-major, one after the other in recv_dev) the entries the SENDER's rows own are
 * written into out_dev; row_lo = global index of this rank's first row. */
int smh_mirror_pack(const void *out_dev, uint32_t n_local, uint32_t n_total, const uint32_t *col_lo, const uint32_t *col_hi,
                    uint32_t n_blocks, void *packed_dev, void *stream);
int smh_mirror_apply(void *out_dev, uint32_t row_lo, uint32_t n_local, uint32_t n_total, const uint32_t *peer_lo,
                     const uint32_t *peer_hi, uint32_t n_blocks, const void *recv_dev, void *stream);

/* One query against many nodes: LinearIndex::find (reference src/index/linear.rs:25-45) with
 * search_minhashes / search_minhashes_containment (reference src/index/search.rs:3-9).  Writes the
 * positions of the nodes whose node.similarity(query) -- or node.containment(query) =
 * count_common / |node| (reference src/index.rs:131-161) -- is > threshold, in node order.
 * out_indices must hold n_nodes entries. */
int smh_find(KmerMinHash *const *nodes, uint32_t n_nodes, const KmerMinHash *query, double threshold,
             bool containment, uint32_t *out_indices, uint32_t *out_count);

/* scaffold's nearest leaf (reference src/index/sbt.rs:361-370): position of the candidate with the
 * largest count_common(leaf, candidate) -- the first one on ties, 0 when none is > 0 -- and that
 * count. */
int smh_most_common(const KmerMinHash *leaf, KmerMinHash *const *candidates, uint32_t n,
                    uint32_t *best_pos, uint64_t *best_common);

/* A set of sketches kept resident in HBM (CSR: hashes + offsets + per-node num), so that repeated
 * one-vs-many queries -- LinearIndex::find over a fixed index, reference src/index/linear.rs:25-45 --
 * upload only the query.  Nodes are copied at construction; later changes to them are not seen. */
typedef struct SmhIndex SmhIndex;
SmhIndex *smh_index_new(KmerMinHash *const *nodes, uint32_t n_nodes);
void smh_index_free(SmhIndex *index);
uint32_t smh_index_len(const SmhIndex *index);
/* same contracts as smh_find / smh_most_common, against the resident nodes */
int smh_index_find(SmhIndex *index, const KmerMinHash *query, double threshold, bool containment,
                   uint32_t *out_indices, uint32_t *out_count);
int smh_index_most_common(SmhIndex *index, const KmerMinHash *leaf, uint32_t *best_pos, uint64_t *best_common);
/* rows x cols block between two resident sets (rows' num truncates); host outputs, any may be NULL.  An index compared with
 * ITSELF keeps the dictionary of its collection (dense ranks at 4 B per hash, component roots, the per-sketch partition table
 * of n x (ranges + 1) x 4 B -- ~330 MB for 10 000 long scaled sketches) so that later all-vs-all calls skip the pre-pass;
 * smh_index_drop_dictionary gives that memory back (the next such call rebuilds it), and so does smh_release_workspace()
 * for every live index. */
void smh_index_drop_dictionary(SmhIndex *index);
int smh_index_compare(SmhIndex *rows, SmhIndex *cols, double *jaccard, uint64_t *common, uint64_t *size,
                      uint64_t *count_common, double *containment);

/* deterministic synthetic DNA of SURVEY.md 8d written to device memory (benchmark input) */
int smh_synth_dna_dev(void *out_dev, uint64_t start, uint64_t len, uint64_t seed, uint64_t n_every,
                      void *stream);

/* The fold's sort on its own (diagnostic entry point for the parity tests): sorts `n` keys in host memory in place,
 * stably, carrying `payload` (n 32-bit values, or NULL) along -- on the device, through the same code as the sketch
 * fold and the compare pre-pass. */
int smh_sort_u64(uint64_t *keys, uint32_t *payload, uintptr_t n);

/* Which kernel serves an N x M compare block is chosen from the block's shape (a wavefront per
 * pair, a few-against-many stream, a per-component pair kernel, the tiled matrix kernel).  The
 * choice NEVER changes a result.  It can be pinned: the parity tests run every route over the same
 * inputs, and a caller that knows its collection is one big component can skip the pair route. */
enum SmhCompareRoute {
  SMH_ROUTE_AUTO = 0, SMH_ROUTE_WAVE = 1, SMH_ROUTE_FEW = 2, SMH_ROUTE_COMPONENTS = 3, SMH_ROUTE_TILED = 4
};
typedef struct SmhCompareTuning {
  uint32_t route;             /* SmhCompareRoute; default AUTO */
  uint32_t visit_all_tiles;   /* tiled route: 1 = launch every tile, not only those that can hold pairs sharing a hash */
  uint32_t use_symmetry;      /* default 1: all-vs-all with one num computes the upper triangle and mirrors it */
  uint64_t comp_pairs_limit;  /* AUTO: at most this many sharing pairs -> per-component pair kernel (default 96 Ki; the library
                                 lowers it to 16 Ki for a dictionary that carries range masks) */
  uint32_t split_frequent;    /* default 1: hashes held by more than a quarter of the sketches (at most 64 of them) do not
                                 connect sketches; pairs that share only such hashes are decided from per-sketch records
                                 instead of being walked */
  uint32_t dictionary;        /* how the pooled hashes of the collection dictionary are sorted: 0 = default (four radix passes over the
                                 32 most significant bits that vary, then the few keys that tie there are put in order), 1 = all
                                 eight byte passes (what the default falls back to; A/B and parity tests: same matrix either way) */
  uint32_t no_range_masks;    /* 1 = the tiled kernel walks every pair from the first range of rank space on; default 0: per-range
                                 bit masks of the shared hashes tell it where a pair's union reaches its cut, and it walks only
                                 from there (same matrix either way) */
} SmhCompareTuning;
void smh_compare_get_tuning(SmhCompareTuning *out);
int smh_compare_set_tuning(const SmhCompareTuning *tuning);   /* NULL restores the defaults; process-wide */

/* What the last block compare did.  Measurement aid and test evidence (which route ran; whether the
 * tiled kernel's global-memory merge branch was taken). */
typedef struct SmhCompareStats {
  uint32_t route;               /* SmhCompareRoute that ran */
  uint32_t rows_per_tile;       /* tiled route */
  uint64_t tiles_visited;       /* tiled: tiles launched; components: pairs walked */
  uint64_t tiles_total;         /* tiled: tiles in the block; components: pairs in the block */
  uint64_t pairs_per_tile;
  uint64_t lds_overflow_steps;  /* tiled: (tile, range) steps merged from global memory instead of the LDS stage */
  uint32_t frequent_hashes;     /* hashes set aside as frequent in this block (0 = none, or too many to set aside) */
  uint32_t pipelined;           /* tiled: 1 = the software-pipelined kernel walked the tiles (blocks that do not fill the chip for long) */
  uint32_t span_halvings;       /* pipelined kernel: stretches whose speculatively grown span did not fit LDS and was rebuilt, halved */
  uint32_t prefetched_after_halving; /* ... tiles in which prefetched boundary crossings were used after such a rebuild */
} SmhCompareStats;
void smh_compare_last_stats(SmhCompareStats *out);

/* The library keeps its device workspace (candidate buffers, the six-frame residue buffer, sort
 * scratch) between calls and only ever grows it; a long-running process can hand the memory back
 * after a large batch.  Sketches, resident indexes and their device copies are not touched. */
int smh_release_workspace(void);
/* Device blocks the library gives up (sketch buffers, mirrors, workspace) are parked in a pool inside the library, per
 * device and size class, instead of hipFree'd: another allocator in the process (PyTorch's) cannot see that memory.
 * The pool is capped (default 1 GiB; environment SOURMASH_AMD_POOL_MB=<MiB> at load time, 0 = no pooling); this call
 * changes the cap at run time (and trims down to it), smh_pool_bytes reports what is parked right now, and
 * smh_release_workspace() empties the pool together with the workspace. */
void smh_pool_set_limit(uint64_t bytes);
uint64_t smh_pool_bytes(void);

/* HIP-event timing of the library's kernels, on the stream they run on.
 * name: "dna_rolling", "dna_generic", "protein_fused", "translate", "hash_windows", "compare_wave", "compare_few",
 * "compare_pair", "compare_fill", "compare_comp", "compare_tiled" (the plain and the pipelined tiled kernels of one call together). */
void smh_profile_enable(int on);
void smh_profile_reset(void);
int smh_profile_get(const char *name, double *total_ms, uint64_t *launches);

#ifdef __cplusplus
}
#endif
#endif
