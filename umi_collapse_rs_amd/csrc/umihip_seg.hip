// Segment index of the large buckets (gfx950): the n-gram partition and the all-pairs evaluation
// inside its sub-buckets.
//
// What it replaces in the reference (tkob-vh/umi-collapse-rs): the same rows of the scope table as
// the tile kernels of umihip_kernels.hip -- Naive::remove_near's linear scans
// (src/data/naive.rs:26-40) with BitSet::bit_count_xor / umi_dist (src/utils/bitset.rs:77-91,
// src/utils/mod.rs:24-26) -- for buckets large enough that evaluating every pair is the cost.
// Two UMIs within k substitutions agree exactly on at least one of k+1 disjoint base ranges, so a
// bucket is cut k+1 times into sub-buckets by the value of one range (a counting sort on the
// device: histogram in prep_kernel, scan, scatter) and only the pairs inside a sub-buckets are
// evaluated, with the same arithmetic; a pair that shares several ranges is reported by the first.
// Result = Naive's, pair for pair.  Integer/bitwise work, 64-lane waves, no MFMA.
#include <hip/hip_runtime.h>

#include "umihip_internal.h"
#include "umihip_device.h"

namespace umihip {

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER_THREAD = SEG_SCAN_CHUNK / SCAN_THREADS; // 16 bins per thread

__device__ __forceinline__ uint32_t tasks_of_bin(uint32_t c) { return seg_tasks_of_bin(c); }

// Exclusive scan of the bin counts -> bin_start, the task list, the pair count: ONE launch, the
// chunks chained by a decoupled look-back.  A block draws its chunk from a ticket counter (so that
// every chunk before it has a block that is already running), sums its 256 bins' (entries, tasks),
// publishes the sums, adds up what the chunks before it have published -- stopping at the first
// that already knows its own prefix -- publishes its prefix and writes its bins.
// Status word of a chunk: bit 63 = sums there, bit 31 = they include all chunks before; entries in
// bits 32..62, tasks in bits 0..30 (both stay below 2^31: checked on the host); one 64-bit access.
constexpr unsigned long long SCAN_HAVE = 1ull << 63, SCAN_PREFIX = 1ull << 31;
__device__ __forceinline__ unsigned long long scan_pack(uint32_t e, uint32_t t) { return ((unsigned long long)e << 32) | t; }
__global__ __launch_bounds__(SCAN_THREADS) void seg_scan_kernel(SegArgs g, unsigned long long *counters)
{
    __shared__ uint2 wsum[SCAN_THREADS / 64];
    __shared__ uint32_t my_chunk;
    __shared__ uint2 prefix;
    unsigned long long *ticket = g.scan_state, *status = g.scan_state + 1;
    if (threadIdx.x == 0) my_chunk = (uint32_t)atomicAdd(ticket, 1ull);
    __syncthreads();
    const uint32_t ci = my_chunk;
    const SegScanChunk ch = g.chunks[ci];
    // a thread owns SCAN_PER_THREAD consecutive bins
    uint32_t c[SCAN_PER_THREAD];
    uint2 mine = make_uint2(0u, 0u);
    const uint32_t b_first = threadIdx.x * SCAN_PER_THREAD;
#pragma unroll
    for (int q = 0; q < SCAN_PER_THREAD; q++) {
        const uint32_t b = b_first + q;
        c[q] = b < ch.nbins ? g.bin_cnt[ch.bin0 + b] : 0u;
        mine.x += c[q];
        mine.y += tasks_of_bin(c[q]);
    }
    uint2 incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t ux = __shfl_up(incl.x, d), uy = __shfl_up(incl.y, d);
        if ((int)(threadIdx.x & 63) >= d) {
            incl.x += ux;
            incl.y += uy;
        }
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x < 64) { // the first wave: the block's sums out, the look-back
        uint2 tot = make_uint2(0u, 0u);
        for (int w = 0; w < SCAN_THREADS / 64; w++) {
            tot.x += wsum[w].x;
            tot.y += wsum[w].y;
        }
        const int lane = threadIdx.x;
        if (lane == 0)
            __hip_atomic_store(&status[ci], SCAN_HAVE | (ci == 0 ? SCAN_PREFIX : 0ull) | scan_pack(tot.x, tot.y),
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        uint2 before = make_uint2(0u, 0u);
        int64_t j0 = (int64_t)ci - 1; // lane l looks at chunk j0 - l
        while (j0 >= 0) {
            const int64_t j = j0 - lane;
            unsigned long long st = SCAN_HAVE | SCAN_PREFIX; // (before the first chunk: an empty prefix)
            if (j >= 0)
                do st = __hip_atomic_load(&status[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                while (!(st & SCAN_HAVE)); // (chunk j's block drew its ticket before this one: it is running)
            const unsigned long long stops = __ballot((st & SCAN_PREFIX) != 0);
            const int first = __ffsll(stops) - 1; // nearest chunk whose sums include everything before it
            const bool take = first < 0 || lane <= first;
            uint32_t e = take && j >= 0 ? (uint32_t)(st >> 32) & 0x7FFFFFFFu : 0u;
            uint32_t t = take && j >= 0 ? (uint32_t)st & 0x7FFFFFFFu : 0u;
            for (int off = 32; off > 0; off >>= 1) {
                e += __shfl_xor(e, off);
                t += __shfl_xor(t, off);
            }
            before.x += e;
            before.y += t;
            if (first >= 0) break;
            j0 -= 64;
        }
        if (lane == 0) {
            if (ci != 0)
                __hip_atomic_store(&status[ci], SCAN_HAVE | SCAN_PREFIX | scan_pack(before.x + tot.x, before.y + tot.y),
                                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            prefix = before;
            if (ci + 1 == g.n_chunks) counters[CNT_SEG_TASKS] = before.y + tot.y;
        }
    }
    __syncthreads();
    const uint2 base = prefix;
    uint2 off = make_uint2(base.x + incl.x - mine.x, base.y + incl.y - mine.y);
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) {
        off.x += wsum[w].x;
        off.y += wsum[w].y;
    }
    // bin_start, the bins' tasks; bin_cnt is cleared for its second use as the scatter cursor
    unsigned long long pairs = 0;
#pragma unroll
    for (int q = 0; q < SCAN_PER_THREAD; q++) {
        const uint32_t b = b_first + q;
        if (b < ch.nbins) {
            g.bin_start[ch.bin0 + b] = off.x;
            g.bin_cnt[ch.bin0 + b] = 0;
            const uint32_t nt = seg_chunks_of(c[q]), e = seg_span_log2(nt), span = 1u << e;
            const uint32_t where = ch.seg | (ch.part << 24) | (e << 28);
            uint32_t ti = off.y;
            for (uint32_t t = 0; t < nt; t++) {
                const uint32_t row0 = off.x + 64u * t;
                for (uint32_t sp = 0; sp < nt - t; sp += span, ti++)
                    if (ti < g.task_cap) g.tasks[ti] = SegTask{row0, off.x + c[q], row0 + 1u + 64u * sp, where};
            }
            const uint32_t n_bin_tasks = ti - off.y;
            pairs += (unsigned long long)c[q] * (c[q] ? c[q] - 1 : 0) / 2;
            off.x += c[q];
            off.y += n_bin_tasks;
        }
    }
    // (64-bit sum over the block, one atomic)
    __shared__ unsigned long long psum[SCAN_THREADS / 64];
    for (int o = 32; o > 0; o >>= 1) pairs += __shfl_down(pairs, o);
    if ((threadIdx.x & 63) == 0) psum[threadIdx.x >> 6] = pairs;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; w++) t += psum[w];
        if (t) atomicAdd(&counters[CNT_SEG_PAIRS], t);
    }
}

template <typename KeyT> struct SegRecOf;
template <> struct SegRecOf<uint32_t> { using type = SegRec32; };
template <> struct SegRecOf<uint64_t> { using type = SegRec64; };
// The 3-bit code of every base of a 2-bit filter key (A 000, T 101, C 110, G 011 as in
// src/utils/read.rs:23-31; the filter key holds the low two bits, the third is their xor): any two
// different bases differ in exactly two bits.  umi_len <= 16: 48 bits.
__device__ __forceinline__ uint64_t expand3(uint32_t fkey, int umi_len)
{
    uint64_t e = 0;
    for (int b = 0; b < umi_len; b++) {
        const uint32_t c = (fkey >> (2 * b)) & 3u;
        e |= (uint64_t)(c | (((c >> 1) ^ c) & 1u) << 2) << (3 * b);
    }
    return e;
}
// ... without the nb bases from b0 on (the bin's own); 0 if more than 10 bases are left
__device__ __forceinline__ uint32_t compare_key(uint64_t e3, int b0, int nb, int umi_len)
{
    if (umi_len - nb > 10) return 0u;
    const uint64_t lo = e3 & ((1ull << (3 * b0)) - 1ull);
    const uint64_t hi = e3 >> (3 * (b0 + nb));
    return (uint32_t)(lo | (hi << (3 * b0)));
}
__device__ __forceinline__ SegRec32 make_rec(uint32_t key, uint32_t idx, int32_t freq, uint32_t ckey) { return SegRec32{key, idx, freq, ckey}; }
__device__ __forceinline__ SegRec64 make_rec(uint64_t key, uint32_t idx, int32_t freq, uint32_t) { return SegRec64{key, idx, freq}; }
__device__ __forceinline__ uint64_t expand3(uint64_t, int) { return 0ull; }

// every entry of a segment to its position in each part's sub-bucket order (the order inside a
// sub-bucket is arbitrary: all its pairs are evaluated, and a pair is reported with its entry
// indices in rank order whatever the positions)
template <typename KeyT>
__global__ __launch_bounds__(256) void seg_scatter_kernel(SegArgs g, const KeyT *__restrict__ fkey,
                                                          const int32_t *__restrict__ freq)
{
    using Rec = typename SegRecOf<KeyT>::type;
    const RangeTask r = g.ranges[blockIdx.x];
    if (r.seg == SEG_NONE) return;
    const SegDesc *__restrict__ sd = g.segs + r.seg;
    Rec *__restrict__ sub = (Rec *)g.sub_rec;
    for (uint32_t i = r.start + threadIdx.x; i < r.end; i += blockDim.x) {
        const KeyT key = fkey[i];
        const int32_t f = freq[i];
        const uint64_t e3 = g.use_ckey ? expand3(key, g.umi_len) : 0ull;
        for (int j = 0; j < g.n_parts; j++) {
            const uint32_t b = sd->bin_off[j] + seg_part_bits(key, sd->b0[j], sd->nb[j]);
            const uint32_t pos = g.bin_start[b] + atomicAdd(&g.bin_cnt[b], 1u);
            sub[pos] = make_rec(key, i, f, g.use_ckey ? compare_key(e3, sd->b0[j], sd->nb[j], g.umi_len) : 0u);
        }
    }
}

// ---- the counting sort's LDS path --------------------------------------------------------------
// A block of 1024 threads owns up to SEG_BLOCK_ENTRIES consecutive entries of one segment (their
// filter keys stay in L2 between the passes) and takes the segment's parts one after the other:
// the block's entries are counted per bin in LDS, and a bin's global word is touched once per
// block.  Count pass: one add of the block's count.  Scatter pass (after the scan has turned the
// totals into bin_start and cleared the words): one returning add that reserves the block's run
// inside the bin; the position of an entry is then bin_start + run start + an LDS counter.
// A (block, bin) run is contiguous in the sub-bucket arrays, so the 16-byte records of a block
// fill whole lines instead of landing one by one.
// The parts of one pass: as many as fit the LDS counters side by side (both parts of a 12-base
// UMI at k = 1: the keys are read once per pass).
constexpr int SEG_PASS_PARTS = 4;
struct SegPass {
    int np;                        // parts in this pass
    int b0[SEG_PASS_PARTS], nb[SEG_PASS_PARTS];
    uint32_t nbins[SEG_PASS_PARTS], bin_off[SEG_PASS_PARTS];
    uint32_t lds_off[SEG_PASS_PARTS]; // first counter of the part in hist[]
    uint32_t total;                // counters of the pass
};
__device__ __forceinline__ SegPass seg_pass_of(const SegDesc *__restrict__ sd, int j0, int n_parts, int per_pass)
{
    SegPass ps;
    ps.np = min(per_pass, n_parts - j0);
    ps.total = 0;
#pragma unroll
    for (int q = 0; q < SEG_PASS_PARTS; q++) {
        const bool on = q < ps.np;
        ps.b0[q] = on ? sd->b0[j0 + q] : 0;
        ps.nb[q] = on ? sd->nb[j0 + q] : 0;
        ps.nbins[q] = on ? 1u << (2 * ps.nb[q]) : 0u;
        ps.bin_off[q] = on ? sd->bin_off[j0 + q] : 0u;
        ps.lds_off[q] = ps.total;
        ps.total += ps.nbins[q];
    }
    return ps;
}
// part and bin of counter c of the pass (c < ps.total)
__device__ __forceinline__ uint32_t seg_pass_global_bin(const SegPass &ps, uint32_t c)
{
    uint32_t gb = 0;
#pragma unroll
    for (int q = 0; q < SEG_PASS_PARTS; q++)
        if (q < ps.np && c >= ps.lds_off[q] && c < ps.lds_off[q] + ps.nbins[q]) gb = ps.bin_off[q] + (c - ps.lds_off[q]);
    return gb;
}

// count the block's entries per bin of the pass's parts into hist[] (LDS, cleared first)
template <typename KeyT>
__device__ __forceinline__ void seg_block_histogram(const SegBlock &blk, const KeyT *__restrict__ fkey,
                                                    const SegPass &ps, uint32_t *hist)
{
    for (uint32_t b = threadIdx.x; b < ps.total; b += SEG_BLOCK_THREADS) hist[b] = 0;
    __syncthreads();
    constexpr int B = 8; // loads in flight per thread
    for (uint32_t i0 = blk.start + threadIdx.x; i0 < blk.end; i0 += B * SEG_BLOCK_THREADS) {
        KeyT key[B];
#pragma unroll
        for (int q = 0; q < B; q++) {
            const uint32_t i = i0 + (uint32_t)q * SEG_BLOCK_THREADS;
            key[q] = i < blk.end ? fkey[i] : KeyT(0);
        }
#pragma unroll
        for (int q = 0; q < B; q++)
            if (i0 + (uint32_t)q * SEG_BLOCK_THREADS < blk.end) {
#pragma unroll
                for (int j = 0; j < SEG_PASS_PARTS; j++)
                    if (j < ps.np) atomicAdd(&hist[ps.lds_off[j] + seg_part_bits(key[q], ps.b0[j], ps.nb[j])], 1u);
            }
    }
    __syncthreads();
}

// the filter key of an entry (umihip_kernels.hip's prep_kernel): 2 bits per base in 32 bits, or the
// 3-bit key itself in 64, N (masked by n_bits) folded onto A
__device__ __forceinline__ void store_filter_key(uint32_t *fkey, uint32_t i, uint64_t k3, int umi_len)
{
    uint32_t fk = 0;
    for (int b = 0; b < umi_len; b++) fk |= (uint32_t)((k3 >> (3 * b)) & 3ull) << (2 * b);
    fkey[i] = fk;
}
__device__ __forceinline__ void store_filter_key(uint64_t *fkey, uint32_t i, uint64_t k3, int) { fkey[i] = k3; }

template <typename KeyT>
__global__ __launch_bounds__(SEG_BLOCK_THREADS) void seg_count_lds_kernel(SegArgs g, KeyT *fkey)
{
    extern __shared__ uint32_t hist[];
    const SegBlock blk = g.blocks[blockIdx.x];
    const SegDesc *__restrict__ sd = g.segs + blk.seg;
    if (g.prep_keys) {
        // the entry kernel's work for this block's entries (prep_kernel, umihip_kernels.hip): filter
        // key, threshold (directional.rs:38), label, and the contract check -- freq >= 1, no N code
        // without nmask, no rise of freq inside the bucket (a segment is one bucket: its first entry
        // is the one place where freq may rise; the host wants CNT_RISES == CNT_START_RISES, and
        // this kernel adds to neither for that entry)
        unsigned int bad = 0, rises = 0;
        for (uint32_t i = blk.start + threadIdx.x; i < blk.end; i += SEG_BLOCK_THREADS) {
            const uint64_t key = g.prep_keys[(size_t)i * g.key_words];
            const uint64_t nm = g.prep_nmask ? g.prep_nmask[(size_t)i * g.key_words] : 0ull;
            const int32_t f = g.prep_freq[i];
            g.prep_thr[i] = threshold_of(g.prep_percentage, f);
            g.prep_label[i] = i;
            const uint64_t k3 = key & ~nm & (g.umi_len >= 21 ? 0x7FFFFFFFFFFFFFFFull : ((1ull << (3 * g.umi_len)) - 1ull));
            store_filter_key(fkey, i, k3, g.umi_len);
            bad += f < 1 ? 1u : 0u;
            const uint64_t b2 = k3 & 0x4924924924924924ull;
            bad += (b2 & ~((k3 << 1) | (k3 << 2))) != 0 ? 1u : 0u;
            if (g.key_words > 1 && !g.prep_nmask)
                bad += wide_n_codes(g.prep_keys + (size_t)i * g.key_words, g.key_words, g.full_umi_len);
            rises += (i > sd->start && f > g.prep_freq[i - 1]) ? 1u : 0u;
        }
        if (__any(bad != 0u)) { // (rare: no reduction tree for it)
            if (bad) atomicAdd(&g.prep_counters[CNT_ERROR], (unsigned long long)bad);
        }
        for (int off = 32; off > 0; off >>= 1) rises += __shfl_down(rises, off);
        if ((threadIdx.x & 63) == 0 && rises) atomicAdd(&g.prep_counters[CNT_RISES], (unsigned long long)rises);
        __syncthreads(); // the block's filter keys are read back below
    }
    for (int j0 = 0; j0 < g.n_parts; j0 += (int)g.parts_per_pass) {
        const SegPass ps = seg_pass_of(sd, j0, g.n_parts, (int)g.parts_per_pass);
        seg_block_histogram(blk, fkey, ps, hist);
        for (uint32_t c = threadIdx.x; c < ps.total; c += SEG_BLOCK_THREADS) {
            const uint32_t cnt = hist[c];
            if (cnt) atomicAdd(&g.bin_cnt[seg_pass_global_bin(ps, c)], cnt);
        }
        __syncthreads();
    }
}

template <typename KeyT>
__global__ __launch_bounds__(SEG_BLOCK_THREADS) void seg_scatter_lds_kernel(SegArgs g, const KeyT *__restrict__ fkey,
                                                                            const int32_t *__restrict__ freq)
{
    using Rec = typename SegRecOf<KeyT>::type;
    extern __shared__ uint32_t hist[];
    const SegBlock blk = g.blocks[blockIdx.x];
    const SegDesc *__restrict__ sd = g.segs + blk.seg;
    Rec *__restrict__ sub = (Rec *)g.sub_rec;
    for (int j0 = 0; j0 < g.n_parts; j0 += (int)g.parts_per_pass) {
        const SegPass ps = seg_pass_of(sd, j0, g.n_parts, (int)g.parts_per_pass);
        seg_block_histogram(blk, fkey, ps, hist);
        // the block's run inside every bin it has entries for: the counter becomes its first position
        for (uint32_t c = threadIdx.x; c < ps.total; c += SEG_BLOCK_THREADS) {
            const uint32_t cnt = hist[c];
            if (cnt) {
                const uint32_t gb = seg_pass_global_bin(ps, c);
                hist[c] = g.bin_start[gb] + atomicAdd(&g.bin_cnt[gb], cnt);
            }
        }
        __syncthreads();
        constexpr int B = 4;
        for (uint32_t i0 = blk.start + threadIdx.x; i0 < blk.end; i0 += B * SEG_BLOCK_THREADS) {
            KeyT key[B];
            int32_t f[B];
#pragma unroll
            for (int q = 0; q < B; q++) {
                const uint32_t i = i0 + (uint32_t)q * SEG_BLOCK_THREADS;
                key[q] = i < blk.end ? fkey[i] : KeyT(0);
                f[q] = i < blk.end ? freq[i] : 0;
            }
            uint64_t e3[B];
#pragma unroll
            for (int q = 0; q < B; q++) e3[q] = g.use_ckey ? expand3(key[q], g.umi_len) : 0ull;
#pragma unroll
            for (int j = 0; j < SEG_PASS_PARTS; j++) {
                if (j < ps.np) {
                    uint32_t pos[B];
#pragma unroll
                    for (int q = 0; q < B; q++)
                        pos[q] = i0 + (uint32_t)q * SEG_BLOCK_THREADS < blk.end
                                     ? atomicAdd(&hist[ps.lds_off[j] + seg_part_bits(key[q], ps.b0[j], ps.nb[j])], 1u)
                                     : 0u;
#pragma unroll
                    for (int q = 0; q < B; q++) {
                        const uint32_t i = i0 + (uint32_t)q * SEG_BLOCK_THREADS;
                        if (i < blk.end)
                            sub[pos[q]] = make_rec(key[q], i, f[q],
                                                   g.use_ckey ? compare_key(e3[q], ps.b0[j], ps.nb[j], g.umi_len) : 0u);
                    }
                }
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ uint32_t readlane_key(uint32_t v, int lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint64_t readlane_key(uint64_t v, int lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// base-level filter test of the inner loop: at most k bases of the two filter keys differ
// (exact on N-free keys: the filter key is the key there)
__device__ __forceinline__ bool within_k(uint32_t z, int k)
{
    return __builtin_popcount((z | (z >> 1)) & 0x55555555u) <= k;
}
__device__ __forceinline__ bool within_k(uint64_t z, int k)
{ // 3-bit codes with the third bit clear: two codes differ in exactly two bits or in none
    return __builtin_popcountll(z) <= 2 * k;
}

// All pairs inside the sub-buckets.  One wave per block, persistent over the task list.  A task
// is 64 rows of a sub-bucket (one per lane, the filter key in a register) against the later
// entries of the same sub-bucket, 64 columns at a time, one per lane as well: column j of the tile
// is broadcast with v_readlane (j is a constant of the unrolled loop), and every lane keeps the
// outcome for its own row as bit j of a 64-bit mask -- eight VALU instructions per column and 64
// pairs (readlane, xor, shift, and-or, popcount, compare, select, or), no branch, no scalar
// dependency, no LDS.  Sub-buckets are dense in neighbours by construction (about one pair in 200
// at config 2: some twenty hits per 64 x 64 tile), so a hit must be cheap where it is found: after
// the tile the lanes with a set bit note (row position, column position) in a queue in LDS, one
// hit per lane and round.  When 64 are queued the wave works them off one per lane: both records
// again (L2), the dedupe rule (a pair that shares an earlier part's bin was reported there), the
// exact distance with the reference's arithmetic where keys carry N, the freq predicate of the
// mode, and the edge goes to the block's LDS stage.
// the key the column loop compares: the record's filter key, or (CK) its compare key
template <bool CK> __device__ __forceinline__ uint32_t loop_key(const SegRec32 &r) { return CK ? r.ckey : r.key; }
template <bool CK> __device__ __forceinline__ uint64_t loop_key(const SegRec64 &r) { return r.key; }
// at most k bases differ
template <bool CK> __device__ __forceinline__ bool loop_within_k(uint32_t z, int k)
{
    return CK ? __builtin_popcount(z) <= 2 * k : within_k(z, k);
}
template <bool CK> __device__ __forceinline__ bool loop_within_k(uint64_t z, int k) { return within_k(z, k); }

// Columns J, J-1, J-2, J-3 of the tile (keys in the lanes of ky) against this lane's row x: the
// outcome "at most k bases differ" of each is shifted into h from below, highest column first.
// Compare keys (CK; lim2 = 2 k, wave-uniform): five instructions per column -- broadcast, xor,
// popcount, compare, add-with-carry -- written out, four columns interleaved so that no broadcast
// or compare mask is read within two instructions of the VALU instruction that wrote it (gfx950:
// two wait states between a VALU write of an SGPR or VCC and a VALU read of it -- the compiler
// covers them with s_nop, inline asm has to keep the distance itself; the compiler's own
// select-and-or accumulation takes two to three instructions instead of the add).
template <bool CK, int J> __device__ __forceinline__ uint32_t four_columns(uint32_t h, uint32_t x, uint32_t ky, int k, uint32_t lim2)
{
    if (CK) {
        uint32_t s0, s1, s2, s3, t0, t1, t2, t3;
        unsigned long long p0, p1, p2, p3; // the compares' lane masks, each its own SGPR pair
        asm("v_readlane_b32 %1, %14, %16\n\t"
            "v_readlane_b32 %2, %14, %17\n\t"
            "v_readlane_b32 %3, %14, %18\n\t"
            "v_readlane_b32 %4, %14, %19\n\t"
            "v_xor_b32_e32 %5, %1, %13\n\t"
            "v_xor_b32_e32 %6, %2, %13\n\t"
            "v_xor_b32_e32 %7, %3, %13\n\t"
            "v_xor_b32_e32 %8, %4, %13\n\t"
            "v_bcnt_u32_b32 %5, %5, 0\n\t"
            "v_bcnt_u32_b32 %6, %6, 0\n\t"
            "v_bcnt_u32_b32 %7, %7, 0\n\t"
            "v_bcnt_u32_b32 %8, %8, 0\n\t"
            "v_cmp_ge_u32_e64 %9, %15, %5\n\t"
            "v_cmp_ge_u32_e64 %10, %15, %6\n\t"
            "v_cmp_ge_u32_e64 %11, %15, %7\n\t"
            "v_cmp_ge_u32_e64 %12, %15, %8\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %9\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %10\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %11\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %12"
            : "+v"(h), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),
              "=&s"(p0), "=&s"(p1), "=&s"(p2), "=&s"(p3)
            : "v"(x), "v"(ky), "s"(lim2), "n"(J), "n"(J - 1), "n"(J - 2), "n"(J - 3)
            : "vcc");
        return h;
    }
#pragma unroll
    for (int j = J; j > J - 4; j--) h = h + h + (within_k(x ^ readlane_key(ky, j), k) ? 1u : 0u);
    return h;
}
template <bool CK, int J> __device__ __forceinline__ uint32_t four_columns(uint32_t h, uint64_t x, uint64_t ky, int k, uint32_t)
{
#pragma unroll
    for (int j = J; j > J - 4; j--) h = h + h + (within_k(x ^ readlane_key(ky, j), k) ? 1u : 0u);
    return h;
}
// columns BASE + 31 .. BASE of the tile
template <bool CK, int BASE, typename KeyT>
__device__ __forceinline__ uint32_t columns32(KeyT x, KeyT ky, int k, uint32_t lim2)
{
    uint32_t h = 0;
    h = four_columns<CK, BASE + 31>(h, x, ky, k, lim2);
    h = four_columns<CK, BASE + 27>(h, x, ky, k, lim2);
    h = four_columns<CK, BASE + 23>(h, x, ky, k, lim2);
    h = four_columns<CK, BASE + 19>(h, x, ky, k, lim2);
    h = four_columns<CK, BASE + 15>(h, x, ky, k, lim2);
    h = four_columns<CK, BASE + 11>(h, x, ky, k, lim2);
    h = four_columns<CK, BASE + 7>(h, x, ky, k, lim2);
    h = four_columns<CK, BASE + 3>(h, x, ky, k, lim2);
    return h;
}

// The same 64 columns against this lane's row, bit-sliced (compare keys, k <= 3): per base of the
// compare key the two low bits of the columns' codes become two wave ballots (bit j = column j), the
// lane XORs them with its own row's bits (as 0 / ~0 masks) -- a mismatch of the base where either
// differs (the third code bit is the xor of the two: src/utils/read.rs:23-31) -- and feeds sticky
// counters "more than l bases differ so far" through v_bitop3: fourteen instructions per base and
// 64 pairs at k = 1, 84 for the six compared bases of a 12-bp position, where the broadcast loop
// above takes 320.  Bit j of the result: at most K bases of column j differ from the row.
template <int K, int NB> __device__ __forceinline__ unsigned long long columns64_sliced(uint32_t x, uint32_t ky)
{
    uint32_t cl[K + 1], ch[K + 1];
#pragma unroll
    for (int l = 0; l <= K; l++) cl[l] = ch[l] = 0u;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const unsigned long long p0 = __ballot((ky & (1u << (3 * b))) != 0u), p1 = __ballot((ky & (2u << (3 * b))) != 0u);
        const uint32_t r0 = (uint32_t)((int32_t)(x << (31 - 3 * b)) >> 31), r1 = (uint32_t)((int32_t)(x << (30 - 3 * b)) >> 31);
        const uint32_t dl = BITOP3((uint32_t)p0 ^ r0, (uint32_t)p1, r1, TT_A | (TT_B ^ TT_C));
        const uint32_t dh = BITOP3((uint32_t)(p0 >> 32) ^ r0, (uint32_t)(p1 >> 32), r1, TT_A | (TT_B ^ TT_C));
#pragma unroll
        for (int l = K; l >= 1; l--) {
            cl[l] = BITOP3(cl[l], cl[l - 1], dl, TT_A | (TT_B & TT_C));
            ch[l] = BITOP3(ch[l], ch[l - 1], dh, TT_A | (TT_B & TT_C));
        }
        cl[0] |= dl;
        ch[0] |= dh;
    }
    return ~(((unsigned long long)ch[K] << 32) | cl[K]);
}
template <int K> __device__ __forceinline__ unsigned long long columns64_sliced_nb(uint32_t x, uint32_t ky, int nb)
{
    switch (nb) { // (wave-uniform: the bases a compare key holds)
    case 1: return columns64_sliced<K, 1>(x, ky);
    case 2: return columns64_sliced<K, 2>(x, ky);
    case 3: return columns64_sliced<K, 3>(x, ky);
    case 4: return columns64_sliced<K, 4>(x, ky);
    case 5: return columns64_sliced<K, 5>(x, ky);
    case 6: return columns64_sliced<K, 6>(x, ky);
    case 7: return columns64_sliced<K, 7>(x, ky);
    case 8: return columns64_sliced<K, 8>(x, ky);
    case 9: return columns64_sliced<K, 9>(x, ky);
    default: return columns64_sliced<K, 10>(x, ky);
    }
}
__device__ __forceinline__ unsigned long long columns64_sliced_k(uint32_t x, uint32_t ky, int k, int nb)
{
    switch (k) {
    case 0: return columns64_sliced_nb<0>(x, ky, nb);
    case 1: return columns64_sliced_nb<1>(x, ky, nb);
    case 2: return columns64_sliced_nb<2>(x, ky, nb);
    default: return columns64_sliced_nb<3>(x, ky, nb);
    }
}
__device__ __forceinline__ unsigned long long columns64_sliced_k(uint64_t, uint64_t, int, int) { return 0ull; } // (never: CK is 32-bit)

// ILP unions per lane, their steps interleaved: a union is a chain of dependent scattered accesses
// (a load or a compare-and-swap per step, each a trip to the memory side), and a wave waits for
// the longest chain among its lanes; with ILP chains per lane in flight the trips of the others
// ride along.  Same algorithm as uf_union, step for step.
template <int ILP>
__device__ __forceinline__ void uf_union_multi(uint32_t *parent, uint32_t (&u)[ILP], uint32_t (&v)[ILP], bool (&act)[ILP])
{
    uint32_t pu[ILP], pv[ILP];
    // (u < v.)  First a direct try, as in uf_union: v, if it is still the root it started as, goes
    // under u with one access; otherwise the swap has returned v's parent and the walk starts with it.
    uint32_t seen[ILP];
#pragma unroll
    for (int s = 0; s < ILP; s++) seen[s] = act[s] ? atomicCAS(&parent[v[s]], v[s], u[s]) : 0u;
#pragma unroll
    for (int s = 0; s < ILP; s++) act[s] = act[s] && seen[s] != v[s] && seen[s] != u[s];
#pragma unroll
    for (int s = 0; s < ILP; s++) {
        pu[s] = act[s] ? ld_parent(&parent[u[s]]) : 0u;
        pv[s] = seen[s]; // (v's parent is known now)
    }
    for (;;) {
        bool cas[ILP], any = false;
#pragma unroll
        for (int s = 0; s < ILP; s++) {
            act[s] = act[s] && pu[s] != pv[s];
            if (act[s] && pu[s] < pv[s]) { // u is the side whose parent is the larger
                uint32_t t = u[s]; u[s] = v[s]; v[s] = t;
                t = pu[s]; pu[s] = pv[s]; pv[s] = t;
            }
            cas[s] = act[s] && u[s] == pu[s]; // a root, as far as was seen: under the other side's parent
            any = any || act[s];
        }
        if (!any) break;
        uint32_t r[ILP];
#pragma unroll
        for (int s = 0; s < ILP; s++) // (all of the step's accesses go out before any is waited for)
            r[s] = !act[s] ? 0u : cas[s] ? atomicCAS(&parent[u[s]], u[s], pv[s]) : ld_parent(&parent[pu[s]]);
#pragma unroll
        for (int s = 0; s < ILP; s++) {
            if (!act[s]) continue;
            if (cas[s]) {
                if (r[s] == u[s]) act[s] = false; // hooked
                else pu[s] = r[s];                // hooked by somebody else meanwhile: that is where it points now
            } else { // climbed
                u[s] = pu[s];
                pu[s] = r[s];
            }
        }
    }
}

// W > 1: keys of W words -- the filter keys hold the first word's 21 bases, every filter hit is decided
// by the distance over all words (src/utils/bitset.rs:77-91, word by word)
template <typename KeyT, bool HAS_N, bool CK, int W = 1>
__global__ __launch_bounds__(64) void seg_pair_kernel(PairArgs a, SegArgs g, float percentage,
                                                      uint32_t part, uint32_t n_parts)
{
    using Rec = typename SegRecOf<KeyT>::type;
    constexpr int ILP = 1;                      // queued hits a lane works off together (more: fewer, longer
                                                // drains, and a wave of config 2 queues under 200 hits in all --
                                                // the unions would all wait for the end of the kernel)
    constexpr uint32_t DRAIN_AT = 64 * ILP;     // a drain starts at this many queued hits ...
    constexpr uint32_t HITQ = DRAIN_AT + 64;    // ... and one round adds at most 64
    __shared__ EdgeStage stage;
    __shared__ uint2 hitq[HITQ]; // (row position, column position) in the sub-bucket arrays
    const int lane = threadIdx.x;
    const Rec *__restrict__ sub = (const Rec *)g.sub_rec;
    const uint4 *__restrict__ task_words = (const uint4 *)g.tasks;
    const bool with_dist = a.mode == MODE_NEIGHBOURS;
    if (lane == 0) {
        stage.count = 0;
        stage.candidates = 0;
    }
    __syncthreads();
    const uint32_t n_tasks = (uint32_t)min((unsigned long long)g.task_cap, a.counters[CNT_SEG_TASKS]);
    unsigned int n_cand = 0, n_direct = 0;
    const uint32_t lim2 = (uint32_t)__builtin_amdgcn_readfirstlane(2 * a.k);
    // masks of the bins of the parts before the task's own: a pair that shares one of them was
    // reported there.  Kept while the tasks stay in one (segment, part): nearly always.
    KeyT dup_mask[SEG_MAX_PARTS - 1];
#pragma unroll
    for (int j = 0; j < SEG_MAX_PARTS - 1; j++) dup_mask[j] = KeyT(0);
    uint32_t have_seg = SEG_NONE, have_part = 0;
    int ck_bases = 0;
    const bool sliced = a.k <= 3 && g.col_sliced != 0;
    uint32_t nq = 0; // queued hits (wave-uniform); the queue outlives a task

    // The queued hits, ILP per lane at a time: both records again (L2), the dedupe rule (a pair
    // that shares an earlier part's bin was reported there), the exact distance with the
    // reference's arithmetic where keys carry N, the freq predicate of the mode; a one-way pair
    // goes to the block's LDS stage, a symmetric one is united on the spot (batched directional
    // path) or staged as well.
    auto drain = [&]() {
        __builtin_amdgcn_wave_barrier();
        for (uint32_t q0 = 0; q0 < nq; q0 += 64 * ILP) {
            Rec ra[ILP], cb[ILP];
            bool live[ILP];
#pragma unroll
            for (int s = 0; s < ILP; s++) {
                const uint32_t q = q0 + (uint32_t)s * 64u + (uint32_t)lane;
                live[s] = q < nq;
                const uint2 h = live[s] ? hitq[q] : make_uint2(0u, 0u);
                ra[s] = sub[live[s] ? h.x : 0u]; // (position 0 exists: the call has a segment)
                cb[s] = sub[live[s] ? h.y : 0u];
            }
            uint32_t uu[ILP], uv[ILP];
            bool unite[ILP];
#pragma unroll
            for (int s = 0; s < ILP; s++) {
                unite[s] = false;
                uu[s] = uv[s] = 0;
                if (!live[s]) continue;
                const KeyT z = ra[s].key ^ cb[s].key;
                bool ok = true;
#pragma unroll
                for (int p = 0; p < SEG_MAX_PARTS - 1; p++)
                    ok = ok && (dup_mask[p] == KeyT(0) || (z & dup_mask[p]) != KeyT(0));
                if (!ok) continue;
                // entry indices in rank order (src/algo/directional.rs:67-72)
                const bool sw = cb[s].idx < ra[s].idx;
                const uint32_t gi = sw ? cb[s].idx : ra[s].idx, gj = sw ? ra[s].idx : cb[s].idx;
                const int32_t fi = sw ? cb[s].freq : ra[s].freq, fj = sw ? ra[s].freq : cb[s].freq;
                int dist;
                if (W > 1) { // bitset.rs:77-91, word by word, and utils/mod.rs:25
                    int res = 0;
#pragma unroll
                    for (int w = 0; w < W; w++) {
                        const uint64_t ka = a.keys[(size_t)gi * W + w], kb = a.keys[(size_t)gj * W + w];
                        const uint64_t xn = HAS_N ? a.nmask[(size_t)gi * W + w] ^ a.nmask[(size_t)gj * W + w] : 0ull;
                        res += __builtin_popcountll(xn | (ka ^ kb)) - __builtin_popcountll(xn) / 3;
                    }
                    dist = res / 2;
                } else if (HAS_N) { // bitset.rs:85-87 (one word) and utils/mod.rs:25
                    const uint64_t ka = a.keys[gi], kb = a.keys[gj];
                    const uint64_t xn = a.nmask[gi] ^ a.nmask[gj];
                    dist = (__builtin_popcountll(xn | (ka ^ kb)) - __builtin_popcountll(xn) / 3) / 2;
                } else { // no N in the call: the filter key is the key
                    dist = filter_key_distance(ra[s].key, cb[s].key);
                }
                n_cand++;
                if (dist > a.k) continue;
                if (a.mode == MODE_NEIGHBOURS) {
                    emit_edge(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, gi, gj, dist, true);
                    continue;
                }
                bool fwd, bwd;
                if (a.mode == MODE_DIRECTIONAL) { // naive.rs:31 under directional.rs:38-39
                    fwd = fj <= threshold_of(percentage, fi);
                    bwd = fi <= threshold_of(percentage, fj);
                } else { // adjacency.rs:56: a root only ever sees entries of larger rank
                    fwd = fj <= a.adj_max_freq;
                    bwd = false;
                }
                if (fwd && bwd) {
                    if (g.uf_parent) { // reachability inside such a set is symmetric: one set
                        unite[s] = true;
                        uu[s] = gi;
                        uv[s] = gj;
                        n_direct++;
                    } else {
                        emit_edge(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, gi | SYM_FLAG, gj, dist, false);
                    }
                } else if (fwd) {
                    emit_edge(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, gi, gj, dist, false);
                } else if (bwd) {
                    emit_edge(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, gj, gi, dist, false);
                }
            }
            if (g.uf_parent) uf_union_multi<ILP>(g.uf_parent, uu, uv, unite);
            __builtin_amdgcn_wave_barrier();
            // (the stage is this wave's alone: its fill level is wave-uniform; a pass adds at most 64 * ILP)
            static_assert(64 * ILP <= EDGE_BUF / 4, "a pass must fit the stage's last quarter");
            if ((unsigned int)__builtin_amdgcn_readfirstlane((int)*(volatile unsigned int *)&stage.count) >= (unsigned int)EDGE_BUF * 3u / 4u)
                flush_edges<64>(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, with_dist, false);
        }
        nq = 0;
        __builtin_amdgcn_wave_barrier();
    };

    // Tasks are dealt statically, task t to block t mod the grid, and two stages run ahead of the
    // task in hand: the record of the task after next and the row and column keys of the next one
    // are under way while this one's columns are walked (a task is one 64 x 64 tile: a few hundred
    // instructions, less than one trip to memory).
    const uint32_t stride = gridDim.x;
    const uint4 none = make_uint4(0u, 0u, 0u, 0u);
    auto task_keys = [&](const uint4 &w, bool valid, KeyT &xk, KeyT &ck) {
        const uint32_t row0 = __builtin_amdgcn_readfirstlane(w.x), end = __builtin_amdgcn_readfirstlane(w.y);
        const uint32_t col0 = __builtin_amdgcn_readfirstlane(w.z);
        xk = pad_row<KeyT>();
        ck = pad_col<KeyT>();
        if (valid && row0 + (uint32_t)lane < end && lane < 64) xk = loop_key<CK>(sub[row0 + (uint32_t)lane]);
        if (valid && col0 + (uint32_t)lane < end) ck = loop_key<CK>(sub[col0 + (uint32_t)lane]);
    };
    uint32_t t = blockIdx.x;
    uint4 tw = t < n_tasks ? task_words[t] : none;
    uint4 tw1 = t + stride < n_tasks && t + stride > t ? task_words[t + stride] : none;
    KeyT x, ky;
    task_keys(tw, t < n_tasks, x, ky);
    while (t < n_tasks) {
        const uint32_t t1 = t + stride, t2 = t1 + stride;
        const bool has1 = t1 < n_tasks && t1 > t, has2 = has1 && t2 < n_tasks && t2 > t1;
        const uint4 tw2 = has2 ? task_words[t2] : none;
        KeyT x1, ky1;
        task_keys(tw1, has1, x1, ky1);
        const uint32_t row0 = __builtin_amdgcn_readfirstlane(tw.x);
        const uint32_t end = __builtin_amdgcn_readfirstlane(tw.y);
        const uint32_t col0 = __builtin_amdgcn_readfirstlane(tw.z);
        const uint32_t where = __builtin_amdgcn_readfirstlane(tw.w);
        const uint32_t seg = where & 0xFFFFFFu, my_part = (where >> 24) & 15u;
        const uint32_t col1 = (uint32_t)min((unsigned long long)end, (unsigned long long)col0 + (64ull << (where >> 28)));
        // A multi-GPU split hands out whole sub-buckets: the order inside one differs from rank to
        // rank (the scatter's atomics), its extent does not -- `end` names the sub-bucket.
        const bool mine = !(n_parts > 1 && (((end ^ (end >> 7)) * 0x9E3779B1u) >> 8) % n_parts != part);
        if (mine) {
            const uint32_t n_rows = min(64u, end - row0);
            const uint32_t r = row0 + (uint32_t)lane;
            if (seg != have_seg || my_part != have_part) { // (wave-uniform)
                if (nq) drain(); // the queued hits belong to the masks in hand
                const SegDesc *__restrict__ sd = g.segs + seg;
#pragma unroll
                for (int j = 0; j < SEG_MAX_PARTS - 1; j++)
                    dup_mask[j] = (uint32_t)j < my_part ? (KeyT)sd->mask[j] : KeyT(0);
                have_seg = seg;
                have_part = my_part;
                ck_bases = __builtin_amdgcn_readfirstlane(g.umi_len - (int)sd->nb[my_part]); // bases a compare key holds
            }
            for (uint32_t c0 = col0; c0 < col1; c0 += 64) {
                // (a task of several tiles -- a sub-bucket beyond 1025 entries: the next 64 columns
                // under way while these are walked)
                const uint32_t cn_pos = c0 + 64u + (uint32_t)lane;
                KeyT kn = pad_col<KeyT>();
                if (c0 + 64u < col1 && cn_pos < end) kn = loop_key<CK>(sub[cn_pos]);
                const uint32_t nc = min(64u, end - c0);
                unsigned long long h; // bit j: this lane's row is within k of column j of the tile
                if (CK && sliced) {
                    h = columns64_sliced_k(x, ky, a.k, ck_bases);
                } else {
                    uint32_t hlo = 0, hhi = 0;
                    hlo = columns32<CK, 0>(x, ky, a.k, lim2);
                    if (nc > 32) hhi = columns32<CK, 32>(x, ky, a.k, lim2);
                    h = ((unsigned long long)hhi << 32) | hlo;
                }
                // the columns this lane's row may pair with: inside the tile, and behind the row
                // (position c0 + j > r, i.e. j > lane - (c0 - row0))
                const int j_min = lane + 1 - (int)(c0 - row0);
                if (nc < 64) h &= (1ull << nc) - 1ull;
                if (j_min > 0) h = j_min >= 64 ? 0ull : h & ~((1ull << j_min) - 1ull);
                if ((uint32_t)lane >= n_rows) h = 0ull;
                while (__any(h != 0ull)) { // one hit per lane and round
                    const unsigned long long bal = __ballot(h != 0ull);
                    if (h) {
                        const int j = __builtin_ctzll(h);
                        h &= h - 1ull;
                        hitq[nq + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull))] =
                            make_uint2(r, c0 + (uint32_t)j);
                    }
                    nq += (uint32_t)__builtin_popcountll(bal);
                    if (nq >= DRAIN_AT) drain();
                }
                ky = kn;
            }
        }
        if (!has1) break;
        t = t1;
        tw = tw1;
        tw1 = tw2;
        x = x1;
        ky = ky1;
    }
    if (nq) drain();
    { // what the stage still holds goes to this block's own slot (seg_edge_append_kernel moves it)
        __builtin_amdgcn_wave_barrier();
        const unsigned int left = min(*(volatile unsigned int *)&stage.count, (unsigned int)EDGE_BUF);
        for (unsigned int i = lane; i < left; i += 64) {
            g.priv_edges[(size_t)blockIdx.x * SEG_PRIV_CAP + i] = stage.e[i];
            if (with_dist) g.priv_dist[(size_t)blockIdx.x * SEG_PRIV_CAP + i] = stage.d[i];
        }
        if (lane == 0) g.priv_cnt[blockIdx.x] = left;
    }
    for (int off = 32; off > 0; off >>= 1) n_cand += __shfl_down(n_cand, off);
    if (lane == 0 && n_cand) atomicAdd(&a.counters[CNT_CANDIDATES], (unsigned long long)n_cand);
    for (int off = 32; off > 0; off >>= 1) n_direct += __shfl_down(n_direct, off);
    if (lane == 0 && n_direct) atomicAdd(&a.counters[CNT_UF_DIRECT], (unsigned long long)n_direct);
}

// offsets of the blocks' slots behind what the list holds already (one block), the new total
__global__ __launch_bounds__(1024) void seg_edge_scan_kernel(uint32_t *priv_cnt, uint32_t n_blocks,
                                                             unsigned long long *counters)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const unsigned long long base = counters[CNT_EDGES];
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < n_blocks ? priv_cnt[i] : 0u;
        uint32_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if ((int)(threadIdx.x & 63) >= d) incl += up;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t off = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += wsum[w];
        if (i < n_blocks) priv_cnt[i] = (uint32_t)min(base + off + incl - v, 0xFFFFFFFFull);
        __syncthreads();
        if (threadIdx.x == 1023) carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) counters[CNT_EDGES] = base + carry;
}

__global__ __launch_bounds__(64) void seg_edge_move_kernel(SegArgs g, const uint32_t *__restrict__ priv_off,
                                                           uint32_t n_blocks, uint2 *__restrict__ edges,
                                                           uint8_t *__restrict__ edge_dist, uint32_t edge_cap,
                                                           const unsigned long long *counters)
{
    const uint32_t b = blockIdx.x;
    const unsigned long long first = priv_off[b];
    const unsigned long long next = b + 1 < n_blocks ? (unsigned long long)priv_off[b + 1] : counters[CNT_EDGES];
    const uint32_t cnt = (uint32_t)(next - first);
    for (uint32_t i = threadIdx.x; i < cnt; i += 64) {
        const unsigned long long pos = first + i;
        if (pos < edge_cap) {
            edges[pos] = g.priv_edges[(size_t)b * SEG_PRIV_CAP + i];
            if (edge_dist) edge_dist[pos] = g.priv_dist[(size_t)b * SEG_PRIV_CAP + i];
        }
    }
}

} // namespace

hipError_t launch_seg_build(const SegArgs &g, void *fkey, const int32_t *freq, bool key32,
                            unsigned long long *counters, hipStream_t s)
{
    if (g.n_chunks == 0 || g.n_ranges == 0) return hipSuccess;
    const size_t lds = (size_t)g.lds_bins * g.parts_per_pass * sizeof(uint32_t);
    if (g.blocks) { // the histogram (else prep_kernel has counted the entries, one atomic each)
        if (key32) seg_count_lds_kernel<uint32_t><<<g.n_blocks, SEG_BLOCK_THREADS, lds, s>>>(g, (uint32_t *)fkey);
        else seg_count_lds_kernel<uint64_t><<<g.n_blocks, SEG_BLOCK_THREADS, lds, s>>>(g, (uint64_t *)fkey);
    }
    seg_scan_kernel<<<g.n_chunks, SCAN_THREADS, 0, s>>>(g, counters);
    if (g.blocks) {
        if (key32) seg_scatter_lds_kernel<uint32_t><<<g.n_blocks, SEG_BLOCK_THREADS, lds, s>>>(g, (const uint32_t *)fkey, freq);
        else seg_scatter_lds_kernel<uint64_t><<<g.n_blocks, SEG_BLOCK_THREADS, lds, s>>>(g, (const uint64_t *)fkey, freq);
    } else {
        if (key32) seg_scatter_kernel<uint32_t><<<g.n_ranges, 256, 0, s>>>(g, (const uint32_t *)fkey, freq);
        else seg_scatter_kernel<uint64_t><<<g.n_ranges, 256, 0, s>>>(g, (const uint64_t *)fkey, freq);
    }
    return hipGetLastError();
}

// one-wave blocks of the pair kernel that fit a CU at once (the waves are persistent: a block that
// starts after the others have left finds its work done, or -- dealt statically -- does it alone)
int seg_pair_blocks_per_cu(bool key32, bool has_n, bool ckey)
{
    int nb = 0;
    hipError_t e;
    if (key32 && ckey)
        e = has_n ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, seg_pair_kernel<uint32_t, true, true>, 64, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, seg_pair_kernel<uint32_t, false, true>, 64, 0);
    else if (key32)
        e = has_n ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, seg_pair_kernel<uint32_t, true, false>, 64, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, seg_pair_kernel<uint32_t, false, false>, 64, 0);
    else
        e = has_n ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, seg_pair_kernel<uint64_t, true, false>, 64, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, seg_pair_kernel<uint64_t, false, false>, 64, 0);
    return e == hipSuccess && nb > 0 ? nb : 16;
}

hipError_t launch_seg_pairs(const PairArgs &a, const SegArgs &g, bool key32, float percentage,
                            uint32_t part, uint32_t n_parts, uint32_t n_blocks, hipStream_t s)
{
    if (g.n_chunks == 0 || n_blocks == 0) return hipSuccess;
    const bool has_n = a.nmask != nullptr;
    if (key32 && g.use_ckey) {
        if (has_n) seg_pair_kernel<uint32_t, true, true><<<n_blocks, 64, 0, s>>>(a, g, percentage, part, n_parts);
        else seg_pair_kernel<uint32_t, false, true><<<n_blocks, 64, 0, s>>>(a, g, percentage, part, n_parts);
    } else if (key32) {
        if (has_n) seg_pair_kernel<uint32_t, true, false><<<n_blocks, 64, 0, s>>>(a, g, percentage, part, n_parts);
        else seg_pair_kernel<uint32_t, false, false><<<n_blocks, 64, 0, s>>>(a, g, percentage, part, n_parts);
    } else if (g.key_words <= 1) {
        if (has_n) seg_pair_kernel<uint64_t, true, false><<<n_blocks, 64, 0, s>>>(a, g, percentage, part, n_parts);
        else seg_pair_kernel<uint64_t, false, false><<<n_blocks, 64, 0, s>>>(a, g, percentage, part, n_parts);
    } else {
#define UMI_SEG_WIDE(WN)                                                                                           \
    do {                                                                                                           \
        if (has_n) seg_pair_kernel<uint64_t, true, false, WN><<<n_blocks, 64, 0, s>>>(a, g, percentage, part, n_parts);  \
        else seg_pair_kernel<uint64_t, false, false, WN><<<n_blocks, 64, 0, s>>>(a, g, percentage, part, n_parts);       \
    } while (0)
        switch (g.key_words) {
        case 2: UMI_SEG_WIDE(2); break;
        case 3: UMI_SEG_WIDE(3); break;
        case 4: UMI_SEG_WIDE(4); break;
        default: return hipErrorInvalidValue;
        }
#undef UMI_SEG_WIDE
    }
    return hipGetLastError();
}

// what the pair kernel's blocks still held in their stages when they ended: from their private
// slots to the list
hipError_t launch_seg_edge_append(const PairArgs &a, const SegArgs &g, uint32_t n_blocks, hipStream_t s)
{
    if (g.n_chunks == 0 || n_blocks == 0) return hipSuccess;
    seg_edge_scan_kernel<<<1, 1024, 0, s>>>(g.priv_cnt, n_blocks, a.counters);
    seg_edge_move_kernel<<<n_blocks, 64, 0, s>>>(g, g.priv_cnt, n_blocks, a.edges,
                                                 a.mode == MODE_NEIGHBOURS ? a.edge_dist : nullptr, a.edge_cap,
                                                 a.counters);
    return hipGetLastError();
}

} // namespace umihip
