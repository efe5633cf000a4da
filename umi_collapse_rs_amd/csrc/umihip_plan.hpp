// Host-side work planning of libumihip.so: which kernel takes which bucket, the tile / range /
// segment tables the kernels read, and the multi-GPU partition of a call.  No HIP calls in here:
// the file also builds with g++ -fsanitize=address,undefined for the CPU-only planner tests
// (tests/cpp/test_plan.cpp).
#pragma once

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "umihip_internal.h"

namespace umihip {

constexpr uint32_t BIG_COL_CHUNK = 16 * COL_TILE; // columns per big task

struct Plan {
    std::vector<PairTask> small_tasks, big_tasks;
    // bit-sliced tasks: [0] column-split tiles, [1] wide tiles, [2]/[3] wide tiles of key-sorted
    // buckets whose kernel keeps the counter state of the 3 / 4 highest units per column run
    std::vector<BsTask> bs_tasks[4];
    // table variant (key-sorted, 32-bit keys, 2 live units): row tiles; the column tiles to walk
    // are found on the device (tab_items_max = all of them)
    std::vector<TabRowTile> tab_rows;
    uint64_t tab_items_max = 0;
    uint64_t n_pairs_eval_tab = 0; // their share of n_pairs_eval if every column tile were walked
    std::vector<PlaneTask> plane_tasks;
    // entries of the buckets the fused kernel does not take, in chunks: what prep and finalize
    // work on (empty for a batch of small positions)
    std::vector<RangeTask> ranges;
    struct BsBucket {
        uint64_t s, e, plane_off;
        uint32_t ngroups;
        bool wide;
        int pu;   // prefix units cached per column run (0: none; needs the bucket sorted by key)
        int live; // table variant: units looked up per column (0: not the table variant)
    };
    std::vector<BsBucket> bs_buckets;
    // segment index (n-gram partition) of the large buckets: descriptors, the blocks of the bin
    // scan, sizes of the device arrays
    std::vector<SegDesc> segs;
    std::vector<SegScanChunk> seg_chunks;
    std::vector<SegBlock> seg_blocks; // entry blocks of the counting sort's LDS path
    uint32_t seg_max_bins = 0;        // most bins any part of any segment has
    int seg_max_rest = 0;             // most bases any part of any segment leaves outside its bins
    int seg_parts = 0;           // k + 1 (0: no segment in this call)
    uint64_t seg_entries = 0;    // entries of all segments (M): each part's sub-bucket order holds M
    uint64_t seg_bins = 0;       // bins of all parts of all segments
    uint64_t seg_task_cap = 0;   // upper bound of the (sub-bucket, 64-row chunk) tasks
    uint64_t n_fused = 0; // buckets left to the fused one-wave kernel
    uint64_t plane_words = 0;
    uint64_t n_pairs = 0, n_pairs_eval = 0, max_bucket = 0, n_tasks_pruned = 0;
    size_t n_bs() const
    {
        size_t t = 0;
        for (auto &v : bs_tasks) t += v.size();
        return t;
    }
    bool any_sorted() const
    {
        for (auto &bb : bs_buckets)
            if (bb.pu || bb.live) return true;
        return false;
    }
};

// Lower bound on the distance between any key of sorted range A and any key of sorted range
// B, from the bases the ranges' own bounds already fix: all keys between lo and hi share
// every base above the highest bit in which lo and hi differ.  bpb = bits per base in the
// filter key (2 for 32-bit keys, 3 for 64-bit ones).
inline int shared_top_bases(uint64_t lo, uint64_t hi, int umi_len, int bpb)
{
    const uint64_t x = lo ^ hi;
    if (!x) return umi_len;
    const int hb = 63 - __builtin_clzll(x);
    return std::max(0, umi_len - 1 - hb / bpb);
}
inline int range_distance_bound(uint64_t a_lo, uint64_t a_hi, uint64_t b_lo, uint64_t b_hi, int umi_len,
                         int bpb)
{
    const int t = std::min(shared_top_bases(a_lo, a_hi, umi_len, bpb),
                           shared_top_bases(b_lo, b_hi, umi_len, bpb));
    int mism = 0;
    const uint64_t mask = (1ull << bpb) - 1;
    for (int i = umi_len - t; i < umi_len; i++)
        mism += ((a_lo >> (bpb * i)) & mask) != ((b_lo >> (bpb * i)) & mask);
    return mism;
}

// Tile tasks of the bit-sliced kernel for every large bucket.  samples (prune mode, else
// null): per bucket the sorted filter keys at positions s, s+128, ..., and e-1.
inline void gen_bs_tasks(Plan &pl, int umi_len, uint32_t col_chunk, int k,
                  const std::vector<std::vector<uint64_t>> *samples, bool key32)
{
    const uint32_t gpl = (uint32_t)bs_groups_per_lane(umi_len);
    const int bpb = key32 ? 2 : 3;
    for (size_t bi = 0; bi < pl.bs_buckets.size(); bi++) {
        const Plan::BsBucket &bb = pl.bs_buckets[bi];
        const uint64_t s = bb.s, e = bb.e;
        if (bb.live) { // table variant: one record per row tile
            const uint32_t tile_groups = 64u * (uint32_t)BS_TAB_G;
            for (uint32_t g0 = 0; g0 < bb.ngroups; g0 += tile_groups) {
                const uint64_t r_lo = s + (uint64_t)g0 * 32;
                pl.tab_rows.push_back(TabRowTile{(uint32_t)s, (uint32_t)e, g0, bb.ngroups, bb.plane_off});
                pl.tab_items_max += (e - r_lo + BS_TAB_TILE - 1) / BS_TAB_TILE;
                pl.n_pairs_eval += (uint64_t)tile_groups * 32 * (e - r_lo);
                pl.n_pairs_eval_tab += (uint64_t)tile_groups * 32 * (e - r_lo);
            }
            continue;
        }
        const uint32_t tile_groups = (bb.wide ? 256u : 64u) * gpl;
        const std::vector<uint64_t> *smp = samples ? &(*samples)[bi] : nullptr;
        std::vector<BsTask> &list = pl.bs_tasks[!bb.wide ? 0 : (bb.pu == 3 ? 2 : (bb.pu == 4 ? 3 : 1))];
        auto key_lo = [&](uint64_t pos) { return (*smp)[(pos - s) / BS_COL_TILE]; };
        auto key_hi = [&](uint64_t end) { // an upper bound of the last key of [.., end)
            if (end >= e) return smp->back();
            const uint64_t idx = (end - s + BS_COL_TILE - 1) / BS_COL_TILE;
            return idx < smp->size() ? (*smp)[idx] : smp->back();
        };
        // two passes: the tasks on the bucket's diagonal first.  They take about three times as
        // long as the others (the filter hits of a key-sorted bucket crowd there), and the launch
        // should end on short tasks.  Within a pass the order is row-tile-major, which shares a
        // row tile's planes in L2.
        for (int pass = 0; pass < 2; pass++)
            for (uint32_t g0 = 0; g0 < bb.ngroups; g0 += tile_groups) {
                const uint64_t r_lo = s + (uint64_t)g0 * 32;
                const uint64_t r_hi = std::min<uint64_t>(e, r_lo + (uint64_t)tile_groups * 32);
                // diagonal chunks of a row tile: those that start before its last row
                const uint64_t c_begin = pass == 0 ? r_lo : r_lo + (r_hi - r_lo + col_chunk - 1) / col_chunk * col_chunk;
                const uint64_t c_end = pass == 0 ? std::min<uint64_t>(e, c_begin + (r_hi - r_lo + col_chunk - 1) / col_chunk * col_chunk) : e;
                for (uint64_t c0 = c_begin; c0 < c_end; c0 += col_chunk) {
                    const uint64_t c1 = std::min<uint64_t>(e, c0 + col_chunk);
                    const bool diag = c0 < r_hi;
                    if (smp && !diag &&
                        range_distance_bound(key_lo(r_lo), key_hi(r_hi), key_lo(c0), key_hi(c1), umi_len,
                                             bpb) > k) {
                        pl.n_tasks_pruned++;
                        continue; // no pair of this tile can be within k
                    }
                    list.push_back(BsTask{(uint32_t)s, (uint32_t)e, g0, bb.ngroups, bb.plane_off,
                                          (uint32_t)c0, (uint32_t)c1, diag ? 1u : 0u, 0u});
                    pl.n_pairs_eval += (uint64_t)tile_groups * 32 * (c1 - c0);
                }
            }
    }
}

// Prefix units worth caching for a key-sorted bucket of n entries: the state of the `pu`
// highest 2-base units is reused along a run of columns that agree in them.  A run of random
// keys is about n / 4^(bases that vary in the prefix) columns long; below ~4 columns the
// bookkeeping costs more than it saves.
inline int choose_prefix_units(uint64_t n, int umi_len)
{
    const int lp = bs_padded_len(umi_len), units = lp / 2, pad = lp - umi_len;
    for (int pu = 4; pu >= 3; pu--) {
        if (pu >= units) continue;
        const int bases = std::max(0, 2 * pu - pad);
        if ((n >> (2 * bases)) >= 4) return pu;
    }
    return 0;
}

// Table variant (32-bit keys): two live units, if the prefix above them (the other units' bases,
// less the padding) still gives runs of ~4 columns; 0 otherwise.
inline int choose_live_units(uint64_t n, int umi_len, uint32_t min_run)
{
    const int lp = bs_padded_len(umi_len), units = lp / 2, pad = lp - umi_len;
    if (units <= 2) return 0;
    const int bases = std::max(0, 2 * (units - 2) - pad);
    return (n >> (2 * bases)) >= min_run ? 2 : 0;
}

// Can the segment index take buckets of this UMI length at this k?  k + 1 parts of at least 3
// bases each (64 sub-buckets per part and more): below that the partition saves less than the
// bit-sliced all-pairs kernels do per pair.
inline bool seg_index_applies(int umi_len, int k)
{
    return k >= 0 && k + 1 <= SEG_MAX_PARTS && umi_len / (k + 1) >= 3;
}

// Segment descriptor of a bucket of n entries: part j covers bases [j L / P, (j+1) L / P); its
// bins are indexed by its leading bases, as many as keep a sub-bucket around 32-64 entries on
// uniform keys (more bins than that only add empty ones to scan, fewer add pairs to evaluate).
inline void plan_segment(Plan &pl, uint64_t s, uint64_t e, int umi_len, int k, bool key32)
{
    const int P = k + 1;
    const uint64_t n = e - s;
    int bits = 0;
    while ((n >> bits) > 1) bits++; // floor(log2 n)
    const int nb_cap = std::max(2, std::min(12, (bits - 4) / 2));
    SegDesc sd;
    sd.start = (uint32_t)s;
    sd.end = (uint32_t)e;
    const uint32_t seg = (uint32_t)pl.segs.size();
    for (int j = 0; j < SEG_MAX_PARTS; j++) {
        sd.bin_off[j] = 0;
        sd.b0[j] = 0;
        sd.nb[j] = 0;
        sd.mask[j] = 0;
    }
    for (int j = 0; j < P; j++) {
        const int b0 = j * umi_len / P, b1 = (j + 1) * umi_len / P;
        const int nb = std::min(b1 - b0, nb_cap);
        const uint64_t bins = 1ull << (2 * nb);
        sd.b0[j] = (uint8_t)b0;
        sd.nb[j] = (uint8_t)nb;
        // 32-bit filter keys: 2 bits per base; 64-bit ones keep the 3-bit layout
        sd.mask[j] = key32 ? (((1ull << (2 * nb)) - 1ull) << (2 * b0)) : (((1ull << (3 * nb)) - 1ull) << (3 * b0));
        sd.bin_off[j] = (uint32_t)pl.seg_bins;
        for (uint64_t b = 0; b < bins; b += SEG_SCAN_CHUNK)
            pl.seg_chunks.push_back({(uint32_t)(pl.seg_bins + b), (uint32_t)std::min<uint64_t>(SEG_SCAN_CHUNK, bins - b),
                                     seg, (uint32_t)j});
        pl.seg_bins += bins;
        pl.seg_max_bins = std::max<uint32_t>(pl.seg_max_bins, (uint32_t)std::min<uint64_t>(bins, 0xFFFFFFFFull));
        pl.seg_max_rest = std::max(pl.seg_max_rest, umi_len - nb);
        pl.seg_task_cap += seg_task_bound(n, bins);
    }
    pl.seg_entries += n;
    for (uint64_t q = s; q < e; q += SEG_BLOCK_ENTRIES)
        pl.seg_blocks.push_back({(uint32_t)q, (uint32_t)std::min<uint64_t>(e, q + SEG_BLOCK_ENTRIES), seg, 0u});
    pl.segs.push_back(sd);
}

// One pass over the bucket table of a call, before anything else looks at it: the copy for the
// upload (pinned staging), the monotonicity check, and everything the plan needs of the buckets
// the fused one-wave kernel takes -- for a batch of 10^5 small positions that is the whole plan, and
// the host's walk over the table is what the step waits for (the kernel is enqueued first), so
// the walk is a branch-free loop the compiler vectorises; a block of buckets that holds a larger
// one (or a step backwards) is walked again the slow way, and those buckets' indices are all
// build_plan visits.
struct TablePass {
    uint64_t n_fused = 0, n_pairs = 0, n_pairs_eval = 0, max_bucket = 0;
    uint64_t bad_at = ~0ull; // first bucket whose end lies before its start
    std::vector<uint64_t> large; // buckets of more than fused_max entries, ascending
};
struct TableBlockSums {
    uint64_t fused, pairs, eval, mx, over;
};
// buckets [b0, b1) of the table, b1 - b0 <= 2048, fused_max <= 1024: sums fit 32 bits
#define UMIHIP_TABLE_BLOCK_BODY                                                                         \
    uint32_t fused = 0, pairs = 0, eval = 0, mx = 0, over = 0;                                          \
    for (uint64_t b = b0; b < b1; b++) {                                                                \
        const uint64_t e = bucket_off[b + 1];                                                           \
        const uint64_t n = e - bucket_off[b]; /* (a step backwards wraps: "large") */                   \
        if (copy) copy[b + 1] = e;                                                                      \
        const uint32_t small = n <= fused_max ? ~0u : 0u;                                               \
        const uint32_t m = (uint32_t)n & small;                                                         \
        fused += m != 0;                                                                                \
        pairs += m * (m - 1u) / 2u; /* (0 for m = 0: the product wraps to 0) */                         \
        eval += m >= 2 ? m * m : 0u;                                                                    \
        mx = m > mx ? m : mx;                                                                           \
        over |= ~small;                                                                                 \
    }                                                                                                   \
    return TableBlockSums{fused, pairs, eval, mx, over};
inline TableBlockSums table_block_generic(const uint64_t *__restrict bucket_off, uint64_t b0, uint64_t b1,
                                          uint32_t fused_max, uint64_t *__restrict copy)
{
    UMIHIP_TABLE_BLOCK_BODY
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline TableBlockSums table_block_avx2(const uint64_t *__restrict bucket_off,
                                                                        uint64_t b0, uint64_t b1, uint32_t fused_max,
                                                                        uint64_t *__restrict copy)
{
    UMIHIP_TABLE_BLOCK_BODY
}
#endif
#undef UMIHIP_TABLE_BLOCK_BODY

inline void scan_table_reset(TablePass &tp)
{
    tp.n_fused = tp.n_pairs = tp.n_pairs_eval = tp.max_bucket = 0;
    tp.bad_at = ~0ull;
    tp.large.clear();
}
// buckets [b_begin, b_end) (copy[b_begin] is the caller's): the table may be walked in pieces, each
// uploaded and handed to the fused kernel while the next is walked
inline void scan_table_range(const uint64_t *__restrict bucket_off, uint64_t b_begin, uint64_t b_end,
                             uint32_t fused_max, uint64_t *__restrict copy, TablePass &tp)
{
#if defined(__x86_64__)
    const bool avx2 = __builtin_cpu_supports("avx2");
#endif
    constexpr uint64_t BLOCK = 2048;
    for (uint64_t b0 = b_begin; b0 < b_end; b0 += BLOCK) {
        const uint64_t b1 = std::min(b_end, b0 + BLOCK);
        TableBlockSums t;
        if (fused_max > 1024) { // (not the fused kernel's range: the 32-bit sums could wrap)
            t = TableBlockSums{0, 0, 0, 0, 0};
            for (uint64_t b = b0; b < b1; b++) {
                const uint64_t e = bucket_off[b + 1], n = e - bucket_off[b];
                if (copy) copy[b + 1] = e;
                if (n <= fused_max) {
                    t.fused += n != 0;
                    t.pairs += n ? n * (n - 1) / 2 : 0;
                    t.eval += n >= 2 ? n * n : 0;
                    t.mx = std::max(t.mx, n);
                } else {
                    t.over = 1;
                }
            }
        } else {
#if defined(__x86_64__)
            t = avx2 ? table_block_avx2(bucket_off, b0, b1, fused_max, copy)
                     : table_block_generic(bucket_off, b0, b1, fused_max, copy);
#else
            t = table_block_generic(bucket_off, b0, b1, fused_max, copy);
#endif
        }
        tp.n_fused += t.fused;
        tp.n_pairs += t.pairs;
        tp.n_pairs_eval += t.eval;
        tp.max_bucket = std::max(tp.max_bucket, t.mx);
        if (t.over)
            for (uint64_t b = b0; b < b1; b++) {
                if (bucket_off[b + 1] < bucket_off[b]) {
                    if (tp.bad_at == ~0ull) tp.bad_at = b;
                } else if (bucket_off[b + 1] - bucket_off[b] > fused_max) {
                    tp.large.push_back(b);
                }
            }
    }
}

inline void scan_table(const uint64_t *__restrict bucket_off, uint64_t n_buckets, uint32_t fused_max,
                       uint64_t *__restrict copy, TablePass &tp)
{
    scan_table_reset(tp);
    if (copy) copy[0] = bucket_off[0];
    scan_table_range(bucket_off, 0, n_buckets, fused_max, copy, tp);
}

// seg_min: buckets of at least this many entries go through the segment index (0: none do)
// pass (may be null): the table has been walked by scan_table with the same fused_max -- only its
// large buckets are visited here, the others' share of the counters comes from it
inline void build_plan(const uint64_t *bucket_off, uint64_t n_buckets, uint32_t small_max, bool use_bs,
                int umi_len, uint32_t fused_max, bool narrow_only, bool cache_prefix, bool tables,
                uint32_t tab_min_run, uint32_t seg_min, int k, bool key32, Plan &pl,
                const TablePass *pass = nullptr)
{
    pl.segs.clear();
    pl.seg_chunks.clear();
    pl.seg_blocks.clear();
    pl.seg_max_bins = 0;
    pl.seg_max_rest = 0;
    pl.seg_parts = 0;
    pl.seg_entries = pl.seg_bins = pl.seg_task_cap = 0;
    const bool seg_ok = seg_min > 0 && seg_index_applies(umi_len, k);
    pl.n_fused = 0;
    pl.ranges.clear();
    uint64_t run_s = 0, run_e = 0; // current run of entries the fused kernel does not take
    uint32_t run_seg = SEG_NONE;
    auto close_run = [&]() {
        for (uint64_t q = run_s; q < run_e; q += RANGE_CHUNK)
            pl.ranges.push_back({(uint32_t)q, (uint32_t)std::min<uint64_t>(run_e, q + RANGE_CHUNK), run_seg});
        run_s = run_e = 0;
        run_seg = SEG_NONE;
    };
    pl.small_tasks.clear();
    pl.big_tasks.clear();
    for (auto &v : pl.bs_tasks) v.clear();
    pl.tab_rows.clear();
    pl.tab_items_max = 0;
    pl.n_pairs_eval_tab = 0;
    pl.plane_tasks.clear();
    pl.bs_buckets.clear();
    pl.plane_words = 0;
    pl.n_pairs = pl.n_pairs_eval = pl.max_bucket = pl.n_tasks_pruned = 0;
    const uint32_t np = 2 * (uint32_t)bs_padded_len(umi_len);
    if (pass) {
        pl.n_fused = pass->n_fused;
        pl.n_pairs = pass->n_pairs;
        pl.n_pairs_eval = pass->n_pairs_eval;
        pl.max_bucket = pass->max_bucket;
    }
    const uint64_t n_visit = pass ? pass->large.size() : n_buckets;
    for (uint64_t vi = 0; vi < n_visit; vi++) {
        const uint64_t b = pass ? pass->large[vi] : vi;
        const uint64_t s = bucket_off[b], e = bucket_off[b + 1];
        const uint64_t n = e - s;
        pl.max_bucket = std::max(pl.max_bucket, n);
        const bool is_seg = seg_ok && n > fused_max && n >= seg_min;
        if (n > fused_max) { // prep and finalize are this bucket's (runs of neighbours merge;
                             // a segment's ranges are its own)
            if (run_e != s || is_seg || run_seg != SEG_NONE) {
                close_run();
                run_s = s;
            }
            run_e = e;
            if (is_seg) run_seg = (uint32_t)pl.segs.size();
        }
        if (n >= 1 && n <= fused_max) pl.n_fused++; // (a bucket of one entry is the fused kernel's too)
        if (n < 2) continue;
        pl.n_pairs += n * (n - 1) / 2;
        if (n <= fused_max) {
            pl.n_pairs_eval += n * n;
        } else if (is_seg) {
            plan_segment(pl, s, e, umi_len, k, key32);
            pl.seg_parts = k + 1;
        } else if (n <= small_max) {
            for (uint64_t r0 = s; r0 < e; r0 += SMALL_ROWS) {
                pl.small_tasks.push_back({(uint32_t)r0, (uint32_t)e, (uint32_t)r0, (uint32_t)e});
                pl.n_pairs_eval += (uint64_t)SMALL_ROWS * (((e - r0) + 31) / 32 * 32);
            }
        } else if (use_bs) {
            const uint32_t ngroups = (uint32_t)((n + 31) / 32);
            // prune mode wants small row tiles: the shorter the key range of a tile, the more
            // leading bases it fixes and the more column chunks it can rule out
            const bool wide = !narrow_only && n >= (uint64_t)BS_WIDE_MIN;
            for (uint32_t g = 0; g < ngroups; g += 2)
                pl.plane_tasks.push_back(
                    {(uint32_t)(s + (uint64_t)g * 32), (uint32_t)e, pl.plane_words, ngroups, g});
            int live = wide && cache_prefix && tables ? choose_live_units(n, umi_len, tab_min_run) : 0;
            // (its item list is sized for the worst case, every column tile of every row tile)
            if (live && (n / (64u * BS_TAB_G * 32u) + 1) * (n / BS_TAB_TILE + 1) / 2 > (1ull << 26)) live = 0;
            pl.bs_buckets.push_back({s, e, pl.plane_words, ngroups, wide,
                                     wide && cache_prefix && !live ? choose_prefix_units(n, umi_len) : 0, live});
            pl.plane_words += (uint64_t)np * ngroups;
        } else {
            for (uint64_t r0 = s; r0 < e; r0 += BIG_ROWS) {
                for (uint64_t c0 = r0; c0 < e; c0 += BIG_COL_CHUNK) {
                    const uint64_t c1 = std::min<uint64_t>(e, c0 + BIG_COL_CHUNK);
                    pl.big_tasks.push_back(
                        {(uint32_t)r0, (uint32_t)e, (uint32_t)c0, (uint32_t)c1});
                    pl.n_pairs_eval += (uint64_t)BIG_ROWS * (((c1 - c0) + 31) / 32 * 32);
                }
            }
        }
    }
    close_run();
}


// ---- multi-GPU: buckets to devices ----------------------------------------------------------
// Longest processing time first on cost n_b^2 + n_b (an all-but-empty bucket still costs a pass;
// ties: lower bucket index first; the device with the least load so far, lowest rank on ties): deterministic, so every rank of a multi-process
// job computes the same partition from the bucket table alone.  owner[b] = rank of bucket b.
inline void partition_buckets_lpt(const uint64_t *bucket_off, uint64_t n_buckets, uint32_t n_ranks,
                                  std::vector<uint32_t> &owner)
{
    owner.assign(n_buckets, 0u);
    if (n_ranks <= 1 || n_buckets == 0) return;
    std::vector<uint64_t> order(n_buckets);
    std::iota(order.begin(), order.end(), 0ull);
    auto size_of = [&](uint64_t b) { return bucket_off[b + 1] - bucket_off[b]; };
    std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return size_of(a) > size_of(b); });
    // (a binary heap keyed by (load, rank); n_b < 2^31, so a cost is below 2^62 and a sum of a
    // few of them can pass 2^64: loads saturate)
    using Item = std::pair<uint64_t, uint32_t>;
    std::vector<Item> heap;
    for (uint32_t r = 0; r < n_ranks; r++) heap.push_back({0ull, r});
    auto cmp = [](const Item &a, const Item &b) { return a > b; }; // min-heap on (load, rank)
    std::make_heap(heap.begin(), heap.end(), cmp);
    for (uint64_t b : order) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        Item &it = heap.back();
        owner[b] = it.second;
        const uint64_t n = size_of(b), c = n * n + n;
        it.first = it.first + c < it.first ? ~0ull : it.first + c;
        std::push_heap(heap.begin(), heap.end(), cmp);
    }
}

} // namespace umihip
