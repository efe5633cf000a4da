/*
 * umihip.h -- C ABI of the MI355X (gfx950) UMI-collapse hot path.
 *
 * This is the drop-in boundary for tkob-vh/umi-collapse-rs.  The reference has
 * no FFI of its own; the seam is its pair of generic traits
 *     trait Algorithm  { fn apply(..) }                    src/algo/mod.rs:13-20
 *     trait DataStruct { new / remove_near / contains }    src/data/mod.rs:11-17
 * called from the bucket loop src/deduplicate_sam.rs:207-233.  Every entry point
 * below names the reference interface it replaces.  Plain pointers and sizes
 * only; nothing here throws or unwinds (the reference builds with panic=abort,
 * Cargo.toml:19): failures come back as a negative status and a thread-local
 * message from umi_last_error().
 *
 * Key format (all entry points): one uint64 per UMI = BitSet.bits[0] of the
 * reference (src/utils/bitset.rs:9-14) for umi_len <= 21, i.e. base i occupies
 * bits 3i..3i+2 with A=000 T=101 C=110 G=011 N=100 (src/utils/read.rs:23-31,
 * src/utils/mod.rs:38-41); nmask = BitSet.n_bits[0] (bits 3i..3i+2 set where
 * base i is N; src/utils/mod.rs:45-50), NULL when no UMI holds an N.
 */
#ifndef UMIHIP_H
#define UMIHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UMI_OK 0
#define UMI_ERR_ARG (-1)   /* invalid argument */
#define UMI_ERR_HIP (-2)   /* HIP runtime failure (message has the hipError) */
#define UMI_ERR_ORDER (-3) /* entries of a bucket are not in rank order / freq < 1 */
#define UMI_ERR_NOMEM (-4)
#define UMI_ERR_NODEV (-5) /* no usable gfx950 device */
#define UMI_ERR_CHAR (-6)  /* character outside ATCGN (reference: panic, utils/mod.rs:77-79) */

#define UMI_ALGO_DIRECTIONAL 0 /* src/algo/directional.rs */
#define UMI_ALGO_ADJACENCY 1   /* src/algo/adjacency.rs  */

#define UMI_MAX_UMI_LEN 21 /* one 64-bit word per key (3 bits per base) */
#define UMI_MAX_WIDE_UMI_LEN 85 /* the _wide entry points: up to 4 words per key */

typedef struct umi_ctx umi_ctx;   /* one per process/GPU; not re-entrant (the reference calls
                                     apply strictly sequentially, deduplicate_sam.rs:207) */
typedef struct umi_data umi_data; /* device-backed DataStruct instance */

/* Counters of one batched call.  n_pairs is the algorithmic unit of work
 * W = sum_b n_b(n_b-1)/2; ms_* are HIP-event times on the call's
 * stream and are filled only when the "profile" option is on. */
typedef struct umi_stats {
    uint64_t n_umis;
    uint64_t n_buckets;
    uint64_t max_bucket;        /* max_umi_count of deduplicate_sam.rs:218 */
    uint64_t n_kept;            /* deduped_count of deduplicate_sam.rs:219 */
    uint64_t n_pairs;           /* W */
    uint64_t n_pairs_evaluated; /* pairs the filter kernels walked (tile padding included; key-sorted
                                 * buckets: only the column tiles their scan kept, see bs_tables) */
    uint64_t n_candidates;      /* filter hits sent to the exact check */
    uint64_t n_edges;           /* directed edges fed to the collapse */
    uint32_t n_rounds;          /* label-propagation rounds */
    uint32_t n_pair_launches;
    float ms_total;
    float ms_prep;
    float ms_pairs;
    float ms_collapse;
    float ms_finalize;
    /* (ABI version 2) with "profile": HIP-event time of the kernel that does the call's pair work, alone
     * -- what a roofline of the call is quoted on -- and which one it was */
    float ms_kernel;
    uint32_t kernel_id; /* UMI_KERNEL_* */
} umi_stats;
#define UMI_KERNEL_NONE 0
#define UMI_KERNEL_FUSED 1     /* small_bucket_kernel: one wave per position of <= 128 UMIs */
#define UMI_KERNEL_SEG_PAIRS 2 /* seg_pair_kernel: all pairs inside the n-gram sub-buckets of deep positions */

/* ---- context ----------------------------------------------------------- */
int umi_ctx_create(int device_id, umi_ctx **out);
/* One context over several GPUs of the node (n_devices in 1..64; an id may repeat: two workers on
 * one GPU).  umi_dedup_batch on it shards the call: independent buckets -- the iterations of
 * src/deduplicate_sam.rs:207-233 share nothing but additive counters -- go to the devices by
 * longest-processing-time on n_b^2, one host thread, stream and workspace per device, results
 * scattered back at the buckets' own offsets (no exchange between devices); a call whose work is
 * one giant bucket (>= "split_min" entries, more than half of the call's n_b^2) has that bucket's
 * pair work split over the devices instead, the edge lists gathered on the first one and collapsed
 * there.  Options set on it apply to every device.  The device-pointer entry points take a
 * single-device context (a device pointer belongs to one device); umi_data_new uses the first
 * device. */
int umi_ctx_create_multi(const int *device_ids, int n_devices, umi_ctx **out);
int umi_ctx_device_count(const umi_ctx *ctx); /* 1 for umi_ctx_create's, n_devices for the above */
void umi_ctx_destroy(umi_ctx *ctx);
/* The bucket -> rank assignment the multi-device context uses, for hosts that run one process per
 * GPU: owner[b] in [0, n_ranks) for every bucket, deterministic (every rank computes the same
 * table from bucket_off alone).  Host code, no GPU. */
int umi_partition_buckets(const uint64_t *bucket_off, uint64_t n_buckets, uint32_t n_ranks,
                          uint32_t *owner);
/* Thread-local text of the last failure on this thread ("" if none). */
const char *umi_last_error(void);
/* Options (none of them changes a result; unknown name -> UMI_ERR_ARG):
 *   "profile"        0/1: record HIP events, fill the ms_* fields of umi_stats
 *   "edge_capacity"  initial capacity of the permitted-pair list, entries (it grows by itself)
 *   "fused_max"      0..128 (default 128): largest bucket the fused one-wave-per-bucket kernel takes
 *   "fused_blocks"   1..64 (default 20): 256-thread blocks per CU of that kernel's persistent grid
 *   "fused_sliced"   0/1 (default 1): that kernel's bit-sliced body for k <= 3 (0: columns one by one)
 *   "small_max"      (default 1024) largest bucket taken as 64-row popcount chunks; above, 2048-row tiles
 *   "seg_index"      0/1 (default 1): buckets of at least "seg_min" entries (default 512) are cut into
 *                    n-gram sub-buckets on the device -- two UMIs within k substitutions agree on one of
 *                    k + 1 base ranges -- and only the pairs inside a sub-bucket are evaluated; same
 *                    result as the all-pairs popcount kernels, which take those buckets when it is 0 or
 *                    when k + 1 parts would be shorter than 3 bases (the parity suite's cross-check);
 *                    n_pairs_evaluated counts the pairs inside the sub-buckets
 *   "seg_min"        2..2^31, see above
 *   "seg_blocks"     one-wave blocks of the segment index's pair kernel (0 = 16 per CU)
 *   "seg_lds"        0/1 (default 1): its counting sort through per-block LDS histograms (0: one atomic
 *                    per entry)
 *   "seg_unite"      0/1 (default 1): symmetric pairs united where the pair kernel finds them (0: through
 *                    the list)
 *   "seg_ckey"       0/1 (default 1): the pair kernel compares 3-bit-per-base compare keys where the bases
 *                    outside a bin fit 32 bits (0: the 2-bit filter keys)
 *   "seg_sliced"     0/1 (default 1): ... 64 columns of a tile at a time, from wave ballots of the columns'
 *                    code bits (k <= 3); 0: one broadcast column at a time
 *   "spin_wait"      0/1 (default 1): the end of a batched call is seen by watching a word in pinned host
 *                    memory that the stream's last kernel writes (0: hipStreamSynchronize)
 *   "table_pieces"   1..64 (default 1): a bucket table of more than 4096 positions is walked, uploaded
 *                    and handed to the fused kernel in this many pieces
 *   "split_min"      multi-device contexts, see umi_ctx_create_multi
 * The round-1 tile kernels (bit-sliced masks, key-sorted scan + item walk, range pruning, hook/jump
 * collapse) and their options ("bitslice", "bs_*", "prune", "two_phase", "ovf_capacity") exist in the
 * development build only (make dev: libumihip_dev.so, -DUMIHIP_DEV), as cross-checks. */
int umi_ctx_set_option(umi_ctx *ctx, const char *name, int64_t value);
/* Version of this header the library was built from (umi_stats grew in 2), for loaders */
#define UMI_ABI_VERSION 2
int umi_abi_version(void);

/* ---- staging helper: src/utils/mod.rs:63-83 (to_bitset) ---------------- */
/* n UMIs of umi_len ASCII bytes each, packed back to back -> keys / nmask
 * (nmask may be NULL).  Host code, no GPU.  UMI_ERR_CHAR where the reference
 * panics. */
int umi_encode_umis(const uint8_t *ascii, uint64_t n, int umi_len, uint64_t *keys,
                    uint64_t *nmask);

/* ---- keys of more than one word (umi_len 22..UMI_MAX_WIDE_UMI_LEN): BitSet.bits as
 *      n_words = ceil(3 * umi_len / 64) words per key, entry-major (keys[i * n_words + w] =
 *      bits[w] of entry i; nmask likewise or NULL), everything else as in umi_dedup_batch.  The
 *      distance is the reference's per-word arithmetic (src/utils/bitset.rs:77-91), a base that
 *      straddles two words included.  Positions of up to 128 UMIs go through the fused kernel (all
 *      words' bases sliced), deep positions through the n-gram partition of the first word's 21 bases
 *      (two UMIs within k overall are within k there) with every candidate pair decided on all words,
 *      the ones in between through an exact all-pairs kernel; a multi-device context shards the
 *      positions as for one-word keys.  Dual 12 + 12 UMIs are the 24-base, two-word case.
 *      umi_encode_umis_wide is to_bitset (src/utils/mod.rs:63-83) for these lengths, host code. */
int umi_encode_umis_wide(const uint8_t *ascii, uint64_t n, int umi_len, int n_words, uint64_t *keys,
                         uint64_t *nmask);
int umi_dedup_batch_wide(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, int n_words,
                         const int32_t *freq, const uint64_t *bucket_off, uint64_t n_buckets,
                         int umi_len, int k, float percentage, int algo, int32_t adj_max_freq,
                         uint8_t *kept, uint32_t *root, umi_stats *stats);
int umi_dedup_batch_wide_device(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask,
                                int n_words, const int32_t *d_freq, const uint64_t *bucket_off,
                                uint64_t n_buckets, int umi_len, int k, float percentage, int algo,
                                int32_t adj_max_freq, uint8_t *d_kept, uint32_t *d_root,
                                void *hip_stream, umi_stats *stats);

/* ---- read staging on the device: the per-read part of
 *      DeduplicateSAM::deduplicate_and_merge, src/deduplicate_sam.rs:148-176
 *      (to_bitset per read, the per-position map UMI -> ReadFreq with freq += 1 and the kept read
 *      chosen by Merge, src/merge/mod.rs:18-51), and the rank order of a position's UMIs
 *      (src/algo/directional.rs:67-72), in the canonical determinisation: positions by first
 *      appearance in the file, UMIs of a position by freq descending, ties by first appearance.
 * in : per read, in file order: align_key (the caller's injective packing of the reference's
 *      Alignment -- strand, unclipped position, reference id, deduplicate_sam.rs:141-145,507-514 --
 *      into 64 bits, or any dense id; only its low align_key_bits bits are looked at),
 *      umi_len ASCII bytes of UMI, score (avg qual or mapq, may be NULL).
 *      merge: 0 = keep the first read of a UMI (merge "any"), 1 = the highest score, the first
 *      on ties (merge/mod.rs:35,49).
 * out: the batched path's input -- keys / nmask (may be NULL) / freq per unique (position, UMI)
 *      in canonical order, rep = file index of the read that represents it, bucket_off
 *      [*n_buckets + 1]; capacity n_reads (bucket_off: n_reads + 1), caller-owned.
 * UMI_ERR_CHAR for a character outside ATCGN (the reference panics, utils/mod.rs:77-79).
 * n_reads < 2^30 per call (UMI_ERR_ARG beyond).  The smaller align_key_bits, the fewer passes the
 * sort of the reads takes: where align_key_bits + 7 bits per 3 bases of UMI fit 64 bits (12-bp UMIs:
 * up to 36 bits of alignment key) the reads are sorted once, on one composed key.
 * The _device form takes and leaves everything in device memory (one synchronisation inside for
 * the two counts); the plain form copies host arrays in and out around it. */
int umi_stage_reads_device(umi_ctx *ctx, const uint64_t *d_align_key, int align_key_bits,
                           const uint8_t *d_umi_ascii, const int32_t *d_score, uint64_t n_reads,
                           int umi_len, int merge, uint64_t *d_keys, uint64_t *d_nmask,
                           int32_t *d_freq, uint64_t *d_rep, uint64_t *d_bucket_off,
                           uint64_t *n_entries, uint64_t *n_buckets, void *hip_stream);
int umi_stage_reads(umi_ctx *ctx, const uint64_t *align_key, int align_key_bits,
                    const uint8_t *umi_ascii, const int32_t *score, uint64_t n_reads, int umi_len,
                    int merge, uint64_t *keys, uint64_t *nmask, int32_t *freq, uint64_t *rep,
                    uint64_t *bucket_off, uint64_t *n_entries, uint64_t *n_buckets);
/* The same for UMIs of up to UMI_MAX_WIDE_UMI_LEN bases: keys / nmask hold n_words =
 * ceil(3 * umi_len / 64) words per entry, entry-major (the input of umi_dedup_batch_wide). */
int umi_stage_reads_wide_device(umi_ctx *ctx, const uint64_t *d_align_key, int align_key_bits,
                                const uint8_t *d_umi_ascii, const int32_t *d_score, uint64_t n_reads, int umi_len,
                                int n_words, int merge, uint64_t *d_keys, uint64_t *d_nmask, int32_t *d_freq,
                                uint64_t *d_rep, uint64_t *d_bucket_off, uint64_t *n_entries, uint64_t *n_buckets,
                                void *hip_stream);
int umi_stage_reads_wide(umi_ctx *ctx, const uint64_t *align_key, int align_key_bits, const uint8_t *umi_ascii,
                         const int32_t *score, uint64_t n_reads, int umi_len, int n_words, int merge, uint64_t *keys,
                         uint64_t *nmask, int32_t *freq, uint64_t *rep, uint64_t *bucket_off, uint64_t *n_entries,
                         uint64_t *n_buckets);

/* ---- batched path: replaces the whole bucket loop
 *      src/deduplicate_sam.rs:207-233 (apply::<UcSAMRead,Naive> per bucket,
 *      counters :217-219) = Directional/Adjacency::apply
 *      (src/algo/directional.rs:57-91, src/algo/adjacency.rs:29-63) over the
 *      Naive store (src/data/naive.rs:26-44). -------------------------------
 * keys/nmask/freq: N = bucket_off[n_buckets] entries; bucket b owns
 * [bucket_off[b], bucket_off[b+1]).  Inside a bucket entries MUST already be in
 * rank order: freq descending (directional.rs:72), ties in first-appearance
 * order (canonical determinisation); freq >= 1.
 * percentage = Cli.percentage (src/cli.rs:25-26), k = Cli.k (src/cli.rs:18-19).
 * adj_max_freq: third argument of remove_near in adjacency.rs:56 (reference: 0).
 * kept[i] = 1 iff entry i survives; survivors in ascending index order are the
 * reference's output order (deduplicate_sam.rs:227-231).  root (may be NULL):
 * global index of the root that removed entry i (ClusterTracker::add_all,
 * directional.rs:42-44).  stats may be NULL.
 * Buffers are caller-owned host memory; the library never frees them. */
int umi_dedup_batch(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask,
                    const int32_t *freq, const uint64_t *bucket_off, uint64_t n_buckets,
                    int umi_len, int k, float percentage, int algo, int32_t adj_max_freq,
                    uint8_t *kept, uint32_t *root, umi_stats *stats);

/* Same contract with keys/nmask/freq/kept/root already resident in this GPU's
 * HBM (d_*), work enqueued on hip_stream (a hipStream_t, NULL = default
 * stream); bucket_off stays a host array.  Returns after the results are
 * complete on the device (the call synchronises the stream to read its
 * counters). */
int umi_dedup_batch_device(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask,
                           const int32_t *d_freq, const uint64_t *bucket_off,
                           uint64_t n_buckets, int umi_len, int k, float percentage, int algo,
                           int32_t adj_max_freq, uint8_t *d_kept, uint32_t *d_root,
                           void *hip_stream, umi_stats *stats);
/* The same with the bucket table resident on the device as well (d_bucket_off: a device copy of
 * bucket_off, or NULL for the call above): nothing of the table is staged or uploaded inside the
 * call -- for a batch of 10^5 small positions that copy is what the first kernel waits for.  The
 * host copy is still the one that is validated and planned from; a device table that differs from
 * it is the caller's error (entries it would lead outside the arrays are skipped and reported as
 * UMI_ERR_ORDER). */
int umi_dedup_batch_device_table(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask,
                                 const int32_t *d_freq, const uint64_t *bucket_off,
                                 const uint64_t *d_bucket_off, uint64_t n_buckets, int umi_len, int k,
                                 float percentage, int algo, int32_t adj_max_freq, uint8_t *d_kept,
                                 uint32_t *d_root, void *hip_stream, umi_stats *stats);

/* The kept mask as one bit per entry (bit i % 8 of byte i / 8; ceil(n / 8) bytes at d_bits), packed
 * on the device and enqueued on hip_stream: what a one-process-per-GPU host all-gathers over RCCL
 * to reassemble the mask of a bucket-sharded job (n / 8 bytes per rank instead of n). */
int umi_pack_mask_device(umi_ctx *ctx, const uint8_t *d_kept, uint64_t n, uint8_t *d_bits,
                         void *hip_stream);

/* The device-pointer call in two halves, for a host that has more to enqueue behind it (the packing
 * and gathering of the kept mask of a multi-GPU step: ~30 us of host time that would otherwise pass
 * with the GPU idle).  umi_dedup_batch_device_begin enqueues the call's work on hip_stream; where every
 * position is the fused kernel's (no host decision is left: BASELINE configs 3, 4, 5) it returns
 * without waiting -- d_kept / d_root are then final in stream order, and work enqueued on the same
 * stream behind the call may read them -- otherwise it runs to its end like umi_dedup_batch_device_table.
 * umi_dedup_batch_end waits for the call's end (if it is still out), reports a contract violation
 * (UMI_ERR_ORDER) and fills stats.  One call may be out per context; any other call on the context
 * that needs its workspace lets it end first (its result keeps waiting for umi_dedup_batch_end; a
 * second begin replaces it). */
int umi_dedup_batch_device_begin(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask,
                                 const int32_t *d_freq, const uint64_t *bucket_off,
                                 const uint64_t *d_bucket_off /* may be NULL */, uint64_t n_buckets,
                                 int umi_len, int k, float percentage, int algo, int32_t adj_max_freq,
                                 uint8_t *d_kept, uint32_t *d_root, void *hip_stream);
int umi_dedup_batch_end(umi_ctx *ctx, umi_stats *stats);

/* ---- one process, several GPUs, resident shards: the batched call on every device of a multi-device
 *      context at once -- device r works on its own arrays d_*[r] (its share of the positions: the
 *      iterations of src/deduplicate_sam.rs:207-233 share nothing but additive counters, so there is no
 *      exchange on the data path), bucket_off[r] / n_buckets[r] are its host tables -- and then the
 *      all-gatherv that reassembles the kept mask on every device: each device's mask packed to bits
 *      (umi_pack_mask_device's layout), padded to slice_bytes, all-gathered over RCCL / xGMI into
 *      d_mask_bits_all[r] (device r's buffer of n_devices * slice_bytes bytes; slice q = device q's
 *      entries in its own order).  d_nmask, d_root, d_mask_bits_all may be NULL (no N anywhere / no roots
 *      wanted / no gather); slice_bytes >= ceil(n_r / 8) for every r.  librccl.so is opened at the
 *      first call that asks for the gather; the device ids of the context must then be distinct. */
int umi_dedup_batch_device_multi(umi_ctx *ctx, const uint64_t *const *d_keys, const uint64_t *const *d_nmask,
                                 const int32_t *const *d_freq, const uint64_t *const *bucket_off,
                                 const uint64_t *n_buckets, int umi_len, int k, float percentage, int algo,
                                 int32_t adj_max_freq, uint8_t *const *d_kept, uint32_t *const *d_root,
                                 uint8_t *const *d_mask_bits_all, uint64_t slice_bytes, umi_stats *stats);

/* ---- multi-GPU split of ONE call's all-pairs work (a single giant bucket
 *      does not shard by buckets).  Each of n_parts ranks holds the same inputs on its own
 *      GPU, evaluates every n_parts-th tile task and gets its share of the permitted-edge
 *      list; the ranks all-gather their lists (RCCL) and every rank -- or one -- collapses the
 *      union.  Entries of d_edges are (src | flag<<31, dst) pairs of uint32 packed in a uint64
 *      and are opaque to the caller.  No counterpart in the reference (it has no second
 *      device); the result equals umi_dedup_batch_device on the same inputs. ------------ */
/* part in [0, n_parts), n_parts >= 2.  d_edges: caller's device buffer of edge_capacity
 * entries; *n_edges_out = entries produced (if it exceeds edge_capacity: UMI_ERR_NOMEM,
 * nothing copied, call again with a larger buffer). */
int umi_pairs_partial_device(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask,
                             const int32_t *d_freq, const uint64_t *bucket_off,
                             uint64_t n_buckets, int umi_len, int k, float percentage, int algo,
                             int32_t adj_max_freq, uint32_t part, uint32_t n_parts,
                             uint64_t *d_edges, uint64_t edge_capacity, uint64_t *n_edges_out,
                             void *hip_stream, umi_stats *stats);
/* Collapse of a gathered edge list over n entries (same index space as the calls that
 * produced it): kept / root as in umi_dedup_batch_device. */
int umi_collapse_edges_device(umi_ctx *ctx, uint64_t n, const uint64_t *d_edges, uint64_t n_edges,
                              int algo, uint8_t *d_kept, uint32_t *d_root, void *hip_stream,
                              umi_stats *stats);

/* ---- per-bucket path: 1:1 with trait DataStruct (src/data/mod.rs:11-17) as
 *      implemented by Naive (src/data/naive.rs).  UMIs are addressed by their
 *      index in the arrays handed to umi_data_new. -------------------------- */
/* DataStruct::new (naive.rs:22-24): takes the {umi -> freq} map of one bucket
 * (n entries, host memory, copied) and builds the all-pairs neighbour lists
 * dist <= max_edits on the GPU once. */
int umi_data_new(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                 uint32_t n, int umi_len, int max_edits, umi_data **out);
/* The same for keys of n_words = ceil(3 * umi_len / 64) words (umi_len up to UMI_MAX_WIDE_UMI_LEN,
 * entry-major as in umi_dedup_batch_wide); remove_near / contains / free as above. */
int umi_data_new_wide(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, int n_words, const int32_t *freq,
                      uint32_t n, int umi_len, int max_edits, umi_data **out);
/* DataStruct::remove_near (naive.rs:26-40): removes and returns every remaining
 * entry o with dist(query,o) <= k && (dist == 0 || freq[o] <= max_freq).
 * k must be <= max_edits.  out_idx has capacity n; ascending index order. */
int umi_data_remove_near(umi_data *d, uint32_t query, int k, int32_t max_freq, uint32_t *out_idx,
                         uint32_t *out_n);
/* DataStruct::contains (naive.rs:42-44): 1 / 0, negative on error. */
int umi_data_contains(const umi_data *d, uint32_t idx);
/* Drop of the store. */
void umi_data_free(umi_data *d);

#ifdef __cplusplus
}
#endif
#endif
