// Internal interface between the host side (umihip_api.cpp) and the gfx950
// kernels (umihip_kernels.hip).  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace umihip {

// One unit of all-pairs work: rows [row0, min(row0 + tile_rows, row_end)) against
// columns [col0, col1), all global entry indices of one bucket.  Only pairs with
// row < col are reported, so a task whose column range starts at row0 covers
// the diagonal tile.
struct PairTask {
    uint32_t row0;
    uint32_t row_end;
    uint32_t col0;
    uint32_t col1;
};

enum PairMode : int {
    MODE_DIRECTIONAL = 0, // directed edges u->v with freq[v] <= thr[u]   (directional.rs:38-39)
    MODE_ADJACENCY = 1,   // forward edges u->v (u<v) with freq[v] <= max_freq (adjacency.rs:56)
    MODE_NEIGHBOURS = 2,  // undirected (i<j, dist) lists for the DataStruct path
};

enum Counter : int {
    CNT_EDGES = 0,      // directed edges appended (may exceed capacity: overflow)
    CNT_CANDIDATES = 1, // filter hits that reached the exact check
    CNT_ERROR = 2,      // prep validation failures
    CNT_KEPT = 3,       // survivors
    CNT_UNKNOWN = 4,    // adjacency collapse: undecided entries
    CNT_RISES = 5,      // entries whose freq is above their predecessor's
    CNT_START_RISES = 6, // ... of which sit at the start of a bucket (the only legal place)
    CNT_OVF = 7,        // filter hits that did not fit a block's LDS queue (global overflow list)
    CNT_ITEMS = 8,      // table kernel: (row tile, column tile) items its scan found to walk ...
    CNT_DIAG_ITEMS = 9, // ... and those on a bucket's diagonal (dense in hits; listed apart, worked off first)
    CNT_GRAB = 10,      // next item to hand out
    CNT_DIAG_GRAB = 11,
    CNT_SEG_TASKS = 12, // segment index: (sub-bucket, 64-row chunk) tasks its scan produced
    CNT_SEG_PAIRS = 13, // ... and the pairs inside its sub-buckets (what the pair kernel compares)
    CNT_KEPT_FUSED = 14, // survivors of the buckets the fused one-wave kernel finished by itself
    CNT_UF_DIRECT = 15, // symmetric pairs the segment index's pair kernel united on the spot (not in the list)
    CNT_EDGES_MOVED = 16, // edges the flatten launch moved from the pair kernel's private slots to the list, behind
                          // the CNT_EDGES it found there (the list's length is the sum of the two)
    CNT_COUNT = 17,
};
// The flags of the collapse rounds ("round r changed a label") sit behind the counters in the
// same device block, so that one memset clears and one copy reads everything the host looks at.
constexpr int MAX_ROUNDS_PER_SYNC = 16; // rounds enqueued between two host checks
constexpr size_t CTRL_FLAGS_OFF = CNT_COUNT * sizeof(unsigned long long);
constexpr size_t CTRL_BYTES = (CTRL_FLAGS_OFF + (MAX_ROUNDS_PER_SYNC + 1) * sizeof(uint32_t) + 255) & ~(size_t)255; // (one fill kernel: a whole number of 16-byte words)

// Bit-sliced kernel: one block works on rows [bucket_start + 32*group0, ...) of one
// bucket (256*G or 64*G groups of 32 rows, see BS_* below) against columns [col0, col1).
struct BsTask {
    uint32_t bucket_start; // global index of the bucket's first entry
    uint32_t bucket_end;
    uint32_t group0;       // first 32-row group of this tile, relative to the bucket
    uint32_t ngroups;      // groups in the bucket = ceil(n / 32)
    uint64_t plane_off;    // word offset of the bucket's planes: planes[plane_off + g*np + b] (group-major)
    uint32_t col0, col1;   // global column range
    uint32_t diag;         // 1 if some column index <= some row index (needs the row<col mask)
    uint32_t pad;
};

// A run of entries for the per-entry kernels (prep, finalize): one block each.
struct RangeTask {
    uint32_t start, end;
    uint32_t seg; // segment (SegDesc index) of the bucket the run lies in, SEG_NONE if it has none
};
constexpr uint32_t SEG_NONE = 0xFFFFFFFFu;

// ---- segment index: the n-gram partition of a large bucket ----------------------------------
// Two UMIs within k substitutions agree exactly on at least one of k+1 disjoint base ranges
// ("parts", pigeonhole).  A large bucket (a segment) is therefore cut k+1 times into
// sub-buckets of entries that share a part's bases, and the all-pairs evaluation runs inside
// the sub-buckets only: every pair within k is in one of them, every pair inside one is decided
// by the same distance arithmetic as before.  The reference family's own n-gram index
// (`--data ngram*` of the CLI, src/cli.rs:43-44, never implemented in the Rust port) prunes the
// same way; the result is Naive's (src/data/naive.rs:26-40), which is all the reference ever runs.
constexpr int SEG_MAX_PARTS = 8; // k + 1
struct SegDesc {
    uint32_t start, end;             // global entry range of the bucket
    uint32_t bin_off[SEG_MAX_PARTS]; // first bin of part j in the call's bin array
    uint8_t b0[SEG_MAX_PARTS];       // first base of part j
    uint8_t nb[SEG_MAX_PARTS];       // leading bases of part j that index its bins (4^nb bins)
    uint64_t mask[SEG_MAX_PARTS];    // filter-key bits those bases occupy: two entries share the bin of
                                     // part j iff (key_a ^ key_b) & mask[j] == 0
};
// An entry in sub-bucket order: everything the pair kernel needs of it, 16 bytes
struct SegRec32 {
    uint32_t key; // filter key
    uint32_t idx; // entry index
    int32_t freq;
    uint32_t ckey; // compare key of the part the record is filed under: the bases outside the bin's own
                   // (those agree inside a sub-bucket), 3 bits each in the reference's code (N folded
                   // onto A), so that bases differing = popcount(a ^ b) / 2; 0 where it does not fit
};
struct SegRec64 {
    uint64_t key;
    uint32_t idx;
    int32_t freq;
};
// One block of the bin scan: bins [bin0, bin0 + nbins) of part `part` of segment `seg`.
struct SegScanChunk {
    uint32_t bin0, nbins; // nbins <= SEG_SCAN_CHUNK
    uint32_t seg, part;
};
constexpr uint32_t SEG_SCAN_CHUNK = 256;
// One unit of sub-bucket work: rows [row0, min(row0 + 64, end)) against columns [col0,
// min(end, col0 + 64 * 2^e)) of the same sub-bucket (which ends at `end`), all positions in the
// sub-bucket arrays; only pairs with row < column count.  A 64-row chunk faces nt - t tiles of 64
// later entries (t = its number, nt = the sub-bucket's chunks).  Up to 16 chunks (1025 entries)
// every tile is a task of its own, e = 0: tasks of one size, which the pair kernel's persistent
// waves can be dealt statically (whole chunks of 1..nt tiles each left the slowest wave of config
// 2 with 1.7 times the average; drawing tasks from counters costs a returning atomic's round trip
// per task, more than a tile).  Beyond that e grows so that a chunk never gives more than 16 tasks.
struct SegTask {
    uint32_t row0, end;
    uint32_t col0;
    uint32_t where; // segment | part << 24 | e << 28
};
#if defined(__HIPCC__)
#define UMIHIP_HD __host__ __device__
#else
#define UMIHIP_HD
#endif
UMIHIP_HD inline uint32_t seg_chunks_of(uint32_t c) { return c >= 2 ? (c - 1 + 63) / 64 : 0u; } // nt
UMIHIP_HD inline uint32_t seg_span_log2(uint32_t nt)
{
    uint32_t e = 0;
    while (((nt + 15) / 16) > (1u << e)) e++;
    return e;
}
// tasks of a sub-bucket of c entries: sum over its chunks t of ceil((nt - t) / span)
UMIHIP_HD inline uint32_t seg_tasks_of_bin(uint32_t c)
{
    const uint32_t nt = seg_chunks_of(c);
    if (nt == 0) return 0;
    const uint32_t span = 1u << seg_span_log2(nt), q = nt / span, r = nt % span;
    return span * (q * (q + 1) / 2) + r * (q + 1);
}
// ... and an upper bound for a part of n entries over `bins` bins (seg_tasks_of_bin(c) <= 1 + c / 7)
inline uint64_t seg_task_bound(uint64_t n, uint64_t bins) { return n / 7 + (bins < n / 2 ? bins : n / 2) + 64; }
// One block of the counting sort's LDS path: entries [start, end) of segment `seg`
// (<= SEG_BLOCK_ENTRIES of them).  The block counts its entries per bin in LDS and touches a bin's
// global word once (count pass: one add; scatter pass: one returning add that reserves the
// block's run inside the bin), so a bin's word takes one atomic per block instead of one per entry
// -- device-scope atomics are served at the memory side of the fabric (~3e10 per second over the
// whole chip, ~90 per microsecond on one word), which is what bounded the per-entry version.
struct SegBlock {
    uint32_t start, end;
    uint32_t seg, pad;
};
constexpr uint32_t SEG_BLOCK_THREADS = 1024;
constexpr uint32_t SEG_BLOCK_ENTRIES = 8 * SEG_BLOCK_THREADS;
constexpr uint32_t SEG_LDS_BINS = 16384; // bins of one part the LDS path takes (64 KB of counters)
constexpr uint32_t RANGE_CHUNK = 2048; // entries per range task (one block of 256 threads)

constexpr int BS_TAB_G = 1;     // row groups of 32 per lane of the table variant (a wave: 64 * 32 * G rows)
constexpr int BS_TAB_TILE = 256;       // columns per work item of the table variant
constexpr int FUSED_MAX = 128; // largest bucket the fused one-wave kernel takes (2 rows per lane)

// One wave transposes 64 rows of one bucket into bit planes.
struct PlaneTask {
    uint32_t row0;       // global index of the first row
    uint32_t bucket_end;
    uint64_t plane_off;
    uint32_t ngroups;
    uint32_t group;      // group index of row0 inside the bucket (even)
};

struct PairArgs {
    const uint64_t *keys;
    const uint64_t *nmask; // may be null
    const int32_t *freq;
    const int32_t *thr;
    const void *fkey; // uint32_t[N] (2 bits/base, umi_len <= 16) or uint64_t[N] (see prep)
    const PairTask *tasks;
    const BsTask *bs_tasks;
    const uint32_t *planes;
    const uint32_t *perm; // prune mode: sorted position -> original entry index (else null)
    uint2 *edges;
    uint8_t *edge_dist; // MODE_NEIGHBOURS only
    unsigned long long *counters;
    uint2 *ovf;        // (row, column) of the filter hits beyond a block's LDS queue, tile indices
    uint32_t ovf_cap;
    uint32_t edge_cap;
    uint32_t n_entries; // entries of the call (keys, fkey, ... have this many)
    int k;
    int mode;
    int32_t adj_max_freq;
};

// tile geometry of the two pair kernels (rows per block)
constexpr int SMALL_THREADS = 64;
constexpr int SMALL_RPT = 1;
constexpr int BIG_THREADS = 256;
constexpr int BIG_RPT = 8;
constexpr int SMALL_ROWS = SMALL_THREADS * SMALL_RPT;
constexpr int BIG_ROWS = BIG_THREADS * BIG_RPT;
constexpr int COL_TILE = 1024; // column keys staged in LDS per step

// segs/bin_cnt (may be null): entries of ranges with a segment also count themselves into the
// bins of their n_seg_parts parts
// skip_seg: the ranges of segments are the count kernel's (SegArgs::prep_keys), which also tells a
// rise at its bucket's start from one inside (seg_from: the smallest segment); entries_too false:
// every range is a segment's, nothing to launch
hipError_t launch_prep(const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                       const uint64_t *bucket_off, uint64_t n_buckets, const RangeTask *ranges,
                       uint32_t n_ranges, uint32_t n, uint32_t fused_max, int umi_len, float percentage,
                       bool key32, void *fkey, int32_t *thr, uint32_t *label, uint32_t *lab,
                       unsigned long long *counters, const SegDesc *segs, int n_seg_parts,
                       uint32_t *bin_cnt, hipStream_t s, bool skip_seg = false, bool entries_too = true,
                       uint32_t seg_from = 0xFFFFFFFFu, int key_words = 1, int full_umi_len = 0);

// ---- segment index (umihip_seg.hip) ----
struct SegArgs {
    const SegDesc *segs;
    const SegScanChunk *chunks;
    uint32_t n_chunks;
    int n_parts;            // k + 1
    uint32_t *bin_cnt;      // [n_bins] entries per bin (prep); reused as the scatter cursor
    uint32_t *bin_start;    // [n_bins] first position of the bin in the sub-bucket arrays
    unsigned long long *scan_state; // [1 + n_chunks] the scan's ticket counter and its chunks' status words
                                    // (zeroed with bin_cnt: they sit behind it in one block)
    SegTask *tasks;         // [task_cap]
    uint32_t task_cap;
    void *sub_rec;          // [n_parts * M] entries in sub-bucket order: SegRec32 / SegRec64, one
                            // 16-byte store per entry and part
    const RangeTask *ranges;
    uint32_t n_ranges;
    const SegBlock *blocks; // LDS path of the counting sort (null: per-entry atomics, histogram by prep)
    uint32_t n_blocks;
    uint32_t lds_bins;      // most bins any part of any segment has
    uint32_t parts_per_pass; // parts whose counters fit the LDS side by side (1..4)
    // the pair kernel's blocks leave what their edge stage still holds at the end in a slot of their
    // own; seg_edge_append moves the slots to the edge list (no storm of atomics on the list's
    // counter when all blocks finish together)
    uint2 *priv_edges;      // [n_blocks * SEG_PRIV_CAP]
    uint8_t *priv_dist;     // ... their distances (DataStruct mode), else null
    uint32_t *priv_cnt;     // [n_blocks] entries in each slot, then (after the scan) their offsets
    // directional batched path: parent array of the union-find (= label[]).  A pair permitted in both
    // directions is united where it is found instead of going through the edge list (null: listed)
    uint32_t *uf_parent;
    // LDS path: the count kernel's blocks also do the entry kernel's work for their entries (filter
    // key, threshold, label, contract check: prep_kernel then skips the segments' ranges) -- one
    // launch and one pass over the keys less.  prep_keys null: prep_kernel has done it.
    const uint64_t *prep_keys, *prep_nmask;
    const int32_t *prep_freq;
    int32_t *prep_thr;
    uint32_t *prep_label;
    float prep_percentage;
    unsigned long long *prep_counters;
    int umi_len;            // (keys of several words: 21, the bases of the first word the filter keys hold)
    int key_words;          // words per key in prep_keys / prep_nmask and in the pair kernel's exact check
    int full_umi_len;       // ... and the whole length of such keys (the N-code check of the entry pass)
    uint32_t use_ckey; // 32-bit keys: the pair kernel compares the records' compare keys (every part of
                       // every segment leaves at most 10 bases outside its bins)
    uint32_t col_sliced; // ... 64 columns at a time, bit-sliced (columns64_sliced), where k <= 3; 0: a broadcast per column
};
// exclusive scan of the bin counts -> bin_start, task list, counters[CNT_SEG_TASKS / _PAIRS];
// then every entry of a segment is copied to its position in each part's sub-bucket order
hipError_t launch_seg_build(const SegArgs &g, void *fkey, const int32_t *freq, bool key32,
                            unsigned long long *counters, hipStream_t s);
constexpr uint32_t SEG_PRIV_CAP = 512; // = the LDS edge stage of a block
// all pairs inside the sub-buckets: filter + exact check + edge emission (percentage: thresholds
// are recomputed from the staged freq).  part/n_parts: a multi-GPU split takes every n_parts-th task.
int seg_pair_blocks_per_cu(bool key32, bool has_n, bool ckey);
hipError_t launch_seg_pairs(const PairArgs &a, const SegArgs &g, bool key32, float percentage,
                            uint32_t part, uint32_t n_parts, uint32_t n_blocks, hipStream_t s);
hipError_t launch_seg_edge_append(const PairArgs &a, const SegArgs &g, uint32_t n_blocks, hipStream_t s);

// ---- multi-word keys (umihip_wide.hip): umi_len 22..85, n_words = 2..4 words per key, entry-major
// a.tasks: rows [row0, row0 + 64) x columns [col0, col1) of one bucket; exact distance from all words
hipError_t launch_wide_pairs(const PairArgs &a, uint32_t n_tasks, int n_words, hipStream_t s);

// ---- read staging on the device (umihip_stage.hip) ----
size_t stage_workspace_bytes(uint32_t n_reads, int n_words);
// reads (alignment key, UMI text, score) -> entries in canonical order + bucket table, all device
// memory (keys / nmask: n_words words per entry); h_pinned4: four pinned 64-bit words for the counts
// the host needs on the way.  0 ok; 1 a character outside ATCGN; negative: -(hipError_t)
int stage_reads_on_device(void *workspace, const uint64_t *d_align, int align_bits, const uint8_t *d_umi,
                          const int32_t *d_score, uint32_t n, int umi_len, int n_words, int merge, uint64_t *d_keys,
                          uint64_t *d_nmask, int32_t *d_freq, uint64_t *d_rep, uint64_t *d_bucket_off,
                          uint64_t *n_entries_out, uint64_t *n_buckets_out, unsigned long long *h_pinned4,
                          hipStream_t s);

// ---- sort and scan primitives of the staging (umihip_radix.hip) ----
constexpr int RADIX_BINS = 256, RADIX_MAX_PASSES = 8; // 8-bit digits; a 64-bit key has at most eight
constexpr int RADIX_HIST_PARTS = 2048;                // blocks of a kernel that counts digits, at most
size_t radix_sort_temp_bytes(uint32_t n);
// stable sort of (key, value) pairs by bits [begin_bit, end_bit) of the key (n < 2^30); the buffers
// are used in turn, *result_in_b says where the result lies.  hist_parts > 0: the digit counts were
// taken by the kernel that wrote the keys -- block b of its hist_parts blocks put its counts of pass
// p (bits [begin_bit + 8p, +8)) and digit d at parts[(b * passes + p) * 256 + d], zeros included,
// parts being what radix_sort_prepare returned; that call zeroes the sort's temporaries and
// therefore comes before that kernel
hipError_t radix_sort_prepare(void *temp, size_t temp_bytes, uint32_t n, int begin_bit, int end_bit, uint32_t **hist_parts,
                              hipStream_t s);
hipError_t radix_sort_pairs_u64(uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, uint32_t n,
                                int begin_bit, int end_bit, void *temp, size_t temp_bytes, bool *result_in_b, hipStream_t s,
                                uint32_t hist_parts = 0);
hipError_t radix_sort_pairs_u32(uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, uint32_t n,
                                int begin_bit, int end_bit, void *temp, size_t temp_bytes, bool *result_in_b, hipStream_t s,
                                uint32_t hist_parts = 0);
size_t scan_temp_bytes(uint32_t n);
hipError_t scan_inclusive_u64(const uint64_t *in, uint64_t *out, uint32_t n, void *temp, size_t temp_bytes, hipStream_t s);
// exclusive scan of a short array in place, by one block (the tile sums of a two-level scan)
hipError_t scan_spine_u64(unsigned long long *sums, uint32_t n, hipStream_t s);

// ---- directional collapse by union-find (umihip_collapse.hip) ----
// comp[] (= label[], identity on entry) becomes the smallest index of each entry's set under the
// symmetric pairs; lab[] is initialised to the identity
hipError_t launch_uf_components(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                                uint32_t *comp, uint32_t *lab, uint32_t n, uint32_t n_edges_hint,
                                hipStream_t s);
// The batched directional path: the sets of the listed symmetric pairs are joined (the segment
// index's pair kernel has united its own), the forest is flattened (launch_uf_flatten), then
// rounds along the listed one-way pairs without pointer jumps.
hipError_t launch_uf_union_list(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                                uint32_t *parent, uint32_t n_edges_hint, hipStream_t s);
// The collapse behind the pair kernels of the batched directional path: what it works on ...
struct CollapseDesc {
    uint32_t *parent, *lab;  // label[] (the union-find forest of the symmetric pairs) and lab[]
    const uint2 *edges;      // the list
    uint32_t edge_cap;
    const RangeTask *ranges; // entries the fused small-bucket kernel left (null: all n)
    uint32_t n_ranges, n;
    uint8_t *kept;
    uint32_t *root;
    unsigned long long *counters;
    uint32_t *changed;       // the rounds' flags in the control block
    // what the segment index's pair kernel left in its blocks' private slots (one-way pairs only: the
    // symmetric ones were united where they were found); the flatten launch appends them to the list
    const uint2 *priv_edges = nullptr; // [priv_blocks * SEG_PRIV_CAP]
    const uint32_t *priv_cnt = nullptr;
    uint32_t priv_blocks = 0;
};
// bytes (a multiple of 8) of the control block to its pinned, device-visible host mirror
// h_seq (pinned, may be null): set to seq once the block has arrived
hipError_t launch_control_to_host(const void *d_ctrl, void *h_ctrl, size_t bytes, hipStream_t s,
                                  unsigned long long *h_seq = nullptr, unsigned long long seq = 0);
// ... phase by phase: comp[v] = root of v in place and lab[v] = v; round `round` along the
// one-way pairs (a no-op once round - 1 was quiet); label = lab[comp[v]], kept, root, survivors
hipError_t launch_collapse_flatten(const CollapseDesc &d, hipStream_t s);
hipError_t launch_collapse_round(const CollapseDesc &d, int round, hipStream_t s);
// check_round >= 0: blocks of their own look, without a store, whether round check_round would still
// move a label (changed[check_round])
hipError_t launch_collapse_finalize(const CollapseDesc &d, hipStream_t s, int check_round = -1);
// bits[i / 8] bit (i % 8) = kept[i] != 0, for i < n (ceil(n / 8) bytes written)
hipError_t launch_pack_mask(const uint8_t *kept, uint64_t n, uint8_t *bits, hipStream_t s);

hipError_t launch_pairs(const PairArgs &a, uint32_t n_tasks, bool big, bool key32, hipStream_t s);

// bit-sliced path (buckets larger than small_max, k <= BS_MAX_K)
constexpr int BS_MAX_K = 3;
constexpr int BS_COL_TILE = 128;   // columns whose masks are staged in LDS per step
constexpr int BS_COL_CHUNK = 1024; // default columns per task (ctx option bs_col_chunk)
constexpr int BS_WIDE_MIN = 32768; // buckets at least this large use 256-thread blocks
// planes per key for a umi length: 2 bits per base, base count rounded up to 8/12/16/22
inline int bs_padded_len(int umi_len) { return umi_len <= 8 ? 8 : umi_len <= 12 ? 12 : umi_len <= 16 ? 16 : 22; }
inline int bs_groups_per_lane(int umi_len) { return umi_len <= 16 ? 2 : 1; }
hipError_t launch_build_planes(const void *fkey2, bool key32, const PlaneTask *tasks,
                               uint32_t n_tasks, uint32_t *planes, int umi_len, hipStream_t s);
// wide: the 4 waves of a block hold different row groups (256*G groups per tile); else they
// hold the same 64*G groups and split the columns
// unit: bases per counted unit of the filter (1 = exact base count, 2 = default)
hipError_t launch_bs_pairs(const PairArgs &a, uint32_t n_tasks, bool wide, bool key32,
                           int umi_len, int unit, int prefix_units, hipStream_t s);
// table variant (key-sorted buckets, 32-bit keys): row tiles of 64 * BS_TAB_G groups, the
// scan's list of (row tile, column tile) items to walk
struct TabRowTile {
    uint32_t bucket_start, bucket_end; // global entry indices
    uint32_t group0, ngroups;          // first group of the tile, groups in the bucket
    uint64_t plane_off;
};
struct TabItem {
    uint32_t row_tile; // index into the row tile list
    uint32_t col0;     // global index of the first column
    uint32_t ncols;    // <= BS_TAB_TILE
    uint32_t diag;     // some column index <= some row index (needs the row < column mask)
};
hipError_t launch_bs_tab(const PairArgs &a, const TabRowTile *rts, uint32_t n_row_tiles, TabItem *items,
                         uint32_t item_cap, int umi_len, uint32_t part, uint32_t n_parts, uint32_t n_waves,
                         bool transposed, hipStream_t s);
// exact check of the n_entries filter hits in a.ovf (the bit-sliced kernels' overflow list)
hipError_t launch_verify_list(const PairArgs &a, bool key32, uint32_t n_entries, hipStream_t s);

// ---- optional prune mode (umihip_sort.hip): sort a large bucket's entries by filter key
size_t sort_temp_bytes(bool key32, uint32_t n, int key_bits);
// true if a bucket of n entries goes through the library's onesweep passes (a restricted bit
// range is only handed to those; its merge path for smaller inputs gets the full key)
bool sort_is_onesweep(uint32_t n);
// fkey_sorted[start..start+n) = sorted keys, perm[start + i] = original global index of the
// i-th smallest; iota_tmp is scratch of the same extent
// (by the bits [begin_bit, key_bits) of the key only)
hipError_t sort_bucket(const void *fkey, bool key32, int begin_bit, int key_bits, uint32_t start, uint32_t n, void *fkey_sorted,
                       uint32_t *perm, uint32_t *iota_tmp, void *tmp, size_t tmp_bytes,
                       hipStream_t s);
// out[i] = keys[pos[i]] widened to 64 bits
hipError_t gather_keys(const void *keys, bool key32, const uint32_t *pos, uint32_t n, uint64_t *out,
                       hipStream_t s);

// Everything for the buckets of 1..fused_max entries, one wave per bucket: contract check,
// thresholds, all pairs, collapse, label[], kept[], root[] (may be NULL), survivors and
// violations into counters[].  Walks bucket_off (device copy) itself.
// sliced: use the bit-sliced body when k <= 3
hipError_t launch_small_buckets(const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                                float percentage, const uint64_t *bucket_off, uint32_t n_buckets,
                                uint32_t fused_max, uint32_t n_entries, uint32_t *label, uint8_t *kept, uint32_t *root,
                                int k, int umi_len, bool sliced, int mode, int32_t adj_max_freq,
                                unsigned long long *counters, uint32_t max_blocks, hipStream_t s);

hipError_t launch_small_buckets_wide(const uint64_t *keys, const uint64_t *nmask, int n_words, const int32_t *freq,
                                     float percentage, const uint64_t *bucket_off, uint32_t n_buckets,
                                     uint32_t fused_max, uint32_t n_entries, uint8_t *kept, uint32_t *root, int k,
                                     int umi_len, unsigned long long *counters, uint32_t max_blocks, hipStream_t s);

// one label-propagation round (hook over edges + pointer jump); round r is a
// no-op on the device when round r-1 changed nothing.
hipError_t launch_prop_round(const uint2 *edges, const unsigned long long *counters,
                             uint32_t edge_cap, uint32_t *label, uint32_t n, uint32_t *changed,
                             int round, uint32_t n_edges_hint, hipStream_t s);

// label[i] = i (collapse of an external edge list)
hipError_t launch_iota(uint32_t *label, uint32_t n, hipStream_t s);
// two-phase directional collapse: components over the symmetric pairs, then the DAG of one-way pairs
hipError_t launch_cc_round(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                           uint32_t *comp, uint32_t n, uint32_t *changed, int round,
                           uint32_t n_edges_hint, hipStream_t s);
hipError_t launch_dag_round(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                            const uint32_t *comp, uint32_t *lab, uint32_t n, uint32_t *changed,
                            int round, uint32_t n_edges_hint, hipStream_t s);
hipError_t launch_map_labels(uint32_t *comp, const uint32_t *lab, uint32_t n, hipStream_t s);

hipError_t launch_finalize(const uint32_t *label, const RangeTask *ranges, uint32_t n_ranges, uint32_t n,
                           uint8_t *kept, uint32_t *root, unsigned long long *counters, hipStream_t s);

// adjacency collapse with max_freq > 0 (greedy in rank order): one iteration
hipError_t launch_adj_iter(const uint2 *edges, const unsigned long long *counters,
                           uint32_t edge_cap, uint8_t *status, uint8_t *blocked, uint32_t *label,
                           uint32_t n, unsigned long long *counters_rw, uint32_t n_edges_hint,
                           hipStream_t s);
hipError_t launch_adj_finalize(const uint8_t *status, const uint32_t *label, const RangeTask *ranges,
                               uint32_t n_ranges, uint32_t n, uint8_t *kept, uint32_t *root,
                               unsigned long long *counters, hipStream_t s);

} // namespace umihip
