/*
 * umi_oracle.h -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker for the HIP path, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED by the reference: tkob-vh/umi-collapse-rs ships no tests,
 * fixtures or golden outputs and there is no Rust
 * toolchain in this image, so the restatement is pinned only by the
 * known-answer vectors G1..G10 hand-derived from the cited lines
 * (tests/golden/kat.json) and by an independent brute-force reachability
 * check in tests/.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference repository root).
 */
#ifndef UMI_ORACLE_H
#define UMI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXW 4 /* words per key: 4*64/3 = 85 bases, far above any UMI */

/* src/utils/bitset.rs:9-14 -- bits + optional n_bits (has_n) */
typedef struct {
    int nwords;
    int has_n;
    uint64_t bits[ORC_MAXW];
    uint64_t nbits[ORC_MAXW];
} orc_bitset;

/* src/utils/mod.rs:63-83 + src/utils/read.rs:23-31.  Returns 0, or -1 where the
 * reference panics (character outside ATCGN, utils/mod.rs:77-79). */
int orc_to_bitset(const uint8_t *s, int len, orc_bitset *out);

/* src/utils/bitset.rs:130-147 (cached Java-style hash, wrapping arithmetic). */
int32_t orc_bitset_hash(const orc_bitset *b);

/* src/utils/bitset.rs:77-91 */
int32_t orc_bit_count_xor(const orc_bitset *a, const orc_bitset *b);

/* src/utils/mod.rs:24-26 */
int32_t orc_umi_dist(const orc_bitset *a, const orc_bitset *b);

/* src/algo/directional.rs:38: (percentage * (freq + 1) as f32) as i32 */
int32_t orc_threshold(float percentage, int32_t freq);

/* src/utils/read.rs:56-63: (sum(qual as f32) / len as f32) as i32 */
int32_t orc_avg_qual(const uint8_t *qual, int len);

/* ---- Naive neighbour store: src/data/naive.rs:12-44 ------------------- */
typedef struct orc_naive orc_naive;
/* naive.rs:22-24.  umis/freq are borrowed for the lifetime of the store. */
orc_naive *orc_naive_new(const orc_bitset *umis, const int32_t *freq, uint32_t n);
void orc_naive_free(orc_naive *d);
/* naive.rs:26-40.  query is an index into umis; removed indices are written to
 * out (capacity n) in ascending index order (the reference returns a HashSet). */
uint32_t orc_naive_remove_near(orc_naive *d, uint32_t query, int32_t k, int32_t max_freq,
                               uint32_t *out);
/* naive.rs:42-44 */
int orc_naive_contains(const orc_naive *d, uint32_t idx);
/* number of umi_dist evaluations so far (for the cpu_baseline leg) */
uint64_t orc_naive_dist_calls(const orc_naive *d);

/* ---- Collapse: src/algo/directional.rs:57-91, src/algo/adjacency.rs:29-63
 * Input order = first-appearance order of the UMIs in the bucket (the
 * canonical determinisation of SURVEY.md 8c); the functions do the stable
 * freq-descending sort themselves (directional.rs:72 / adjacency.rs:45).
 * out_idx (capacity n) receives the surviving input indices in output order.
 * root_of (capacity n, may be NULL) receives for every input index the input
 * index of the root whose visit removed it.  Returns the number of survivors.
 * dist_calls (may be NULL) accumulates the umi_dist evaluations performed.   */
uint32_t orc_directional_apply(const orc_bitset *umis, const int32_t *freq, uint32_t n, int32_t k,
                               float percentage, uint32_t *out_idx, uint32_t *root_of,
                               uint64_t *dist_calls);
uint32_t orc_adjacency_apply(const orc_bitset *umis, const int32_t *freq, uint32_t n, int32_t k,
                             int32_t max_freq /* reference: 0 (adjacency.rs:56) */,
                             uint32_t *out_idx, uint32_t *root_of, uint64_t *dist_calls);

/* ---- Batched form used by the parity tests -----------------------------
 * Single-word keys (umi_len <= 21), entries already in canonical rank order
 * inside each bucket (freq descending, ties by first appearance), i.e. the
 * same contract as umi_dedup_batch in include/umihip.h.  Restates the bucket
 * loop src/deduplicate_sam.rs:207-233 calling apply once per bucket.
 * kept[i]=1 for survivors; root[i] = global index of the removing root.
 * Returns 0, or -1 if an entry is out of rank order.                        */
int orc_dedup_batch(const uint64_t *keys, const uint64_t *nmask /* or NULL */,
                    const int32_t *freq, const uint64_t *bucket_off, uint64_t n_buckets,
                    int umi_len, int32_t k, float percentage, int algo /* 0 dir, 1 adj */,
                    int32_t adj_max_freq, uint8_t *kept, uint32_t *root, uint64_t *dist_calls);

/* The same for keys of n_words words (umi_len > 21; keys / nmask entry-major,
 * [i * n_words + w] = bits[w] / n_bits[w] of entry i): -2 unless n_words is what
 * BitSet::new_with_len gives for umi_len (bitset.rs:17-18). */
int orc_dedup_batch_wide(const uint64_t *keys, const uint64_t *nmask, int n_words, const int32_t *freq,
                         const uint64_t *bucket_off, uint64_t n_buckets, int umi_len, int32_t k,
                         float percentage, int algo, int32_t adj_max_freq, uint8_t *kept,
                         uint32_t *root, uint64_t *dist_calls);

/* ---- Staging: src/deduplicate_sam.rs:148-176, merge/mod.rs
 * reads arrive in file order; bucket_id[i] is the alignment-key id (first
 * appearance numbering is done by the caller), umi is umi_len ASCII bytes per
 * read, score[i] is avg qual or mapq.  merge: 0 any, 1 avgqual/mapqual (>=).
 * Produces, in canonical order (buckets by first appearance, UMIs by freq
 * desc then first appearance): keys/nmask/freq/rep (index of representative
 * read) and bucket_off.  Arrays are caller-allocated with capacity n_reads
 * (bucket_off: n_reads+1).  Returns the number of unique (bucket,UMI) entries
 * through *n_out and the number of buckets through *b_out; 0 ok, -1 bad char. */
int orc_stage_reads(const uint32_t *bucket_id, const uint8_t *umi, const int32_t *score,
                    uint64_t n_reads, int umi_len, int merge, uint64_t *keys, uint64_t *nmask,
                    int32_t *freq, uint64_t *rep, uint64_t *bucket_off, uint64_t *n_out,
                    uint64_t *b_out);

#ifdef __cplusplus
}
#endif
#endif
