// Device-side helpers shared by the kernel translation units (umihip_kernels.hip,
// umihip_seg.hip, umihip_collapse.hip): threshold arithmetic of the reference, the per-block
// edge stage and the exact check of a filter hit (verify_pair).  Everything sits in an
// anonymous namespace: each TU gets its own copy, nothing here is part of an interface.
#pragma once

#include <hip/hip_runtime.h>

#include "umihip_internal.h"

namespace umihip {
namespace {

constexpr int CHECK_BLOCK = 32; // columns between two "any hit?" checks

// v_bitop3_b32: any boolean of three words in one instruction; tt = the truth table over
// TT_A / TT_B / TT_C (the columns of the first, second, third operand)
#define BITOP3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
constexpr unsigned TT_A = 0xF0, TT_B = 0xCC, TT_C = 0xAA;

__device__ __forceinline__ int popc(uint32_t x) { return __builtin_popcount(x); }
__device__ __forceinline__ int popc(uint64_t x) { return __builtin_popcountll(x); }

// Padding keys for rows/columns past the end of a range.  They only have to be
// unlikely to pass the filter: every hit is re-checked against the index range.
template <typename KeyT> __device__ __forceinline__ KeyT pad_row();
template <> __device__ __forceinline__ uint32_t pad_row<uint32_t>() { return 0xFFFFFFFFu; }
template <> __device__ __forceinline__ uint64_t pad_row<uint64_t>() { return ~0ull; }
template <typename KeyT> __device__ __forceinline__ KeyT pad_col();
template <> __device__ __forceinline__ uint32_t pad_col<uint32_t>() { return 0x0F000000u; }
template <> __device__ __forceinline__ uint64_t pad_col<uint64_t>() { return 0xF000000000000000ull; }

// Rust `f32 as i32` (saturating, NaN -> 0) of percentage * (freq+1) as f32,
// src/algo/directional.rs:38.
__device__ __forceinline__ int32_t threshold_of(float percentage, int32_t freq)
{
    // freq + 1 wraps in a release build of the reference (Cargo.toml:16-19)
    const float prod = __fmul_rn(percentage, (float)(int32_t)((uint32_t)freq + 1u));
    // v_cvt_i32_f32 is Rust's `as i32`: truncation toward zero, out-of-range values (infinities
    // included) saturate, NaN gives 0 -- one instruction where the C cast needs three range tests
    int32_t t;
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(t) : "v"(prod));
    return t;
}

// Bases 21 .. umi_len - 1 of a key of several words that hold the N code (100): a caller that passes
// no nmask promises there is none (the first word's 21 bases are looked at with the filter key).
__device__ __forceinline__ unsigned int wide_n_codes(const uint64_t *__restrict__ key, int key_words, int umi_len)
{
    unsigned int bad = 0;
    for (int b = 21; b < umi_len; b++) {
        const int bit = 3 * b, w = bit >> 6, sh = bit & 63;
        uint64_t c = key[w] >> sh;
        if (sh > 61 && w + 1 < key_words) c |= key[w + 1] << (64 - sh); // a straddling base
        bad += (c & 7ull) == 4ull ? 1u : 0u;
    }
    return bad;
}

// sum `cnt` over the block (256 threads) and add it to *dst with one atomic
__device__ __forceinline__ void block_count_add(unsigned int cnt, unsigned long long *dst)
{
    __shared__ unsigned int part[4];
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(dst, (unsigned long long)t);
    }
    __syncthreads(); // part[] may be reused by a second call
}

// Edge list entries are (src, dst); SYM_FLAG on src marks a pair permitted in both
// directions, stored once (entry indices stay below 2^31).
constexpr uint32_t SYM_FLAG = 0x80000000u;

// Per-block staging of emitted edges in LDS: one global atomic per flush instead of one
// per edge (a single hot counter word saturates near 90 atomics/us on this chip).
constexpr int EDGE_BUF = 512;
struct EdgeStage {
    uint2 e[EDGE_BUF];
    uint8_t d[EDGE_BUF];
    unsigned int count;      // edges staged (may run past EDGE_BUF: the excess went direct)
    unsigned int candidates; // filter hits seen by this block
    unsigned int base;       // flush: global position of e[0]
};

__device__ __forceinline__ void emit_edge(EdgeStage *st, uint2 *edges, uint8_t *edge_dist,
                                          unsigned long long *counters, uint32_t edge_cap,
                                          uint32_t u, uint32_t v, int dist, bool with_dist)
{
    const unsigned int slot = atomicAdd(&st->count, 1u);
    if (slot < EDGE_BUF) {
        st->e[slot] = make_uint2(u, v);
        st->d[slot] = (uint8_t)dist;
    } else { // stage full (a very dense tile): append directly
        const unsigned long long pos = atomicAdd(&counters[CNT_EDGES], 1ull);
        if (pos < edge_cap) {
            edges[pos] = make_uint2(u, v);
            if (with_dist) edge_dist[pos] = (uint8_t)dist;
        }
    }
}

// Exact check of one filter hit, with the reference's arithmetic, and edge emission.
// Cold path (a few hits per million pairs): kept out of line, arguments by value so
// that the kernel's argument block stays in SGPRs.
__attribute__((unused)) __device__ __noinline__ void verify_pair(const uint64_t *__restrict__ keys,
                                         const uint64_t *__restrict__ nmask,
                                         const int32_t *__restrict__ freq,
                                         const int32_t *__restrict__ thr, uint2 *edges,
                                         uint8_t *edge_dist, unsigned long long *counters,
                                         EdgeStage *st, uint32_t edge_cap, int k, int mode,
                                         int32_t adj_max_freq, uint32_t row_end, uint32_t col1,
                                         uint32_t gi, uint32_t gj, const uint32_t *perm)
{
    if (gi >= row_end || gj >= col1 || gi >= gj) return;
    if (perm) { // prune mode: the tile lives in key-sorted order; back to entry indices
        const uint32_t oi = perm[gi], oj = perm[gj];
        gi = min(oi, oj);
        gj = max(oi, oj);
    }
    atomicAdd(&st->candidates, 1u);
    const uint64_t ka = keys[gi], kb = keys[gj];
    const uint64_t na = nmask ? nmask[gi] : 0ull, nb = nmask ? nmask[gj] : 0ull;
    const uint64_t x = na ^ nb;
    // bitset.rs:85-87 (one word) and utils/mod.rs:25
    const int bcx = __builtin_popcountll(x | (ka ^ kb)) - __builtin_popcountll(x) / 3;
    const int dist = bcx / 2;
    if (dist > k) return;
    if (mode == MODE_NEIGHBOURS) {
        emit_edge(st, edges, edge_dist, counters, edge_cap, gi, gj, dist, true);
        return;
    }
    const int32_t fi = freq[gi], fj = freq[gj];
    bool fwd, bwd;
    if (mode == MODE_DIRECTIONAL) {
        fwd = fj <= thr[gi]; // naive.rs:31 with max_freq = threshold(start) (directional.rs:38-39)
        bwd = fi <= thr[gj];
    } else {
        fwd = fj <= adj_max_freq; // adjacency.rs:56
        bwd = false;              // a root only ever sees entries of larger rank
    }
    if (fwd && bwd) // both directions permitted: one flagged entry, halves the list
        emit_edge(st, edges, edge_dist, counters, edge_cap, gi | SYM_FLAG, gj, dist, false);
    else if (fwd)
        emit_edge(st, edges, edge_dist, counters, edge_cap, gi, gj, dist, false);
    else if (bwd)
        emit_edge(st, edges, edge_dist, counters, edge_cap, gj, gi, dist, false);
}

// Block-wide: move the staged edges to the global list.  Called by every thread.
template <int THREADS>
__device__ __forceinline__ void flush_edges(EdgeStage *st, uint2 *edges, uint8_t *edge_dist,
                                            unsigned long long *counters, uint32_t edge_cap,
                                            bool with_dist, bool final)
{
    __syncthreads();
    const unsigned int n = min(st->count, (unsigned int)EDGE_BUF);
    // Keep staging until the buffer is three quarters full: long runs of edges of one tile
    // task stay together in the list, so the collapse's label gathers hit the same lines.
    // (uniform: count is stable between the two barriers)
    if (n == 0 || (!final && n < (unsigned int)(EDGE_BUF * 3 / 4))) return;
    if (threadIdx.x == 0)
        st->base = (unsigned int)min(atomicAdd(&counters[CNT_EDGES], (unsigned long long)n),
                                     (unsigned long long)0xFFFFFFFFu);
    __syncthreads();
    const unsigned int base = st->base;
    for (unsigned int i = threadIdx.x; i < n; i += THREADS) {
        const unsigned long long pos = (unsigned long long)base + i;
        if (pos < edge_cap) {
            edges[pos] = st->e[i];
            if (with_dist) edge_dist[pos] = st->d[i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) st->count = 0;
}

// Bases in which two filter keys differ (2-bit codes, N folded onto A: never above the exact
// distance).  The unit-level filter lets through pairs that differ in two bases of one unit:
// this count, from two neighbouring reads of the tile's own key array, drops them before the
// exact check gathers keys, freq and thresholds through the permutation.
__device__ __forceinline__ int filter_key_distance(uint32_t a, uint32_t b)
{
    const uint32_t x = a ^ b;
    return __builtin_popcount((x | (x >> 1)) & 0x55555555u);
}
__device__ __forceinline__ int filter_key_distance(uint64_t a, uint64_t b)
{ // 3 bits per base (the third is 0 in a folded key)
    const uint64_t x = a ^ b;
    return __builtin_popcountll((x | (x >> 1) | (x >> 2)) & 0x1249249249249249ull);
}

} // namespace
} // namespace umihip

namespace umihip {
namespace {

// ---- segment index: bin of an entry in one part ------------------------------------------------
// The nb leading bases (from base b0) of a filter key, as 2 bits per base.  32-bit filter keys
// are 2 bits per base already; 64-bit ones keep the reference's 3-bit layout (N folded onto A, so
// the third bit of every base is 0 and two codes differ in exactly two bits or none).
__device__ __forceinline__ uint32_t seg_part_bits(uint32_t fkey, int b0, int nb)
{
    return (fkey >> (2 * b0)) & ((1u << (2 * nb)) - 1u); // nb <= 12
}
__device__ __forceinline__ uint32_t seg_part_bits(uint64_t k3, int b0, int nb)
{
    uint32_t v = 0;
    for (int b = 0; b < nb; b++) v |= (uint32_t)((k3 >> (3 * (b0 + b))) & 3ull) << (2 * b);
    return v;
}
// mask of the key bits seg_part_bits looks at: two keys share the bin iff (a ^ b) & mask == 0
__device__ __forceinline__ uint32_t seg_part_mask(uint32_t, int b0, int nb)
{
    return ((1u << (2 * nb)) - 1u) << (2 * b0);
}
__device__ __forceinline__ uint64_t seg_part_mask(uint64_t, int b0, int nb)
{
    return ((1ull << (3 * nb)) - 1ull) << (3 * b0);
}

} // namespace
} // namespace umihip

// ---- union-find over the symmetric pairs (umihip_collapse.hip; the segment index's pair kernel
// unites the pairs it finds on the spot) ---------------------------------------------------------
namespace umihip {
namespace {

// parent[] is read and written with agent-scope accesses (sc1: past the CU's L1, which other CUs'
// stores never refresh).  parent[v] <= v always (an entry is only ever pointed at a smaller one), so
// every ancestor of v is smaller than v, no cycle can form, and the root is the smallest index
// of the set.
//
// The union is Rem's algorithm without its splicing: both paths are climbed together, always on
// the side whose parent is the larger, and the climb ends when the two parents agree or the larger
// side turns out to be a root, which is then hooked with one compare-and-swap.  What matters on
// this chip: the root of the other -- smaller -- side is never looked at.  With the usual
// find-then-link every find of the giant component (80 % of the 10^6 entries of config 2) ends
// with a load of its root's word to see that it is one; the words next to it in its 64-byte line
// are hooked by atomics all the time, which keeps dropping the line from every XCD's L2, and a
// single line served from the memory side takes ~90 requests per microsecond: 0.23 to 2.1 ms for
// the pass, by how early the edge order lets the giant component form.  Here a pair inside one
// tree ends on "parents agree", and a new entry is hooked under whatever its partner points at:
// 0.15 ms whatever the order, about five scattered agent-scope accesses per pair at the
// ~3e10 per second the chip serves.  Measured and dropped: splicing the climbed side over to the
// other path (atomicMin: +0.03 ms, plain store: +0.07 ms -- the paths are short, the extra writes
// are not free), ordinary loads instead of sc1 ones (no difference).
__device__ __forceinline__ uint32_t ld_parent(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_parent(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (u < v: the pair's entries in rank order)
__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t u, uint32_t v)
{
    // First a direct try: v, if it is still the root it started as, goes under u with one access
    // (the usual case early on, when most entries are alone; u < v keeps parents below children).
    // Otherwise the swap has returned v's parent and the walk below starts with it.
    const uint32_t seen = atomicCAS(&parent[v], v, u);
    if (seen == v || seen == u) return;
    uint32_t pu = ld_parent(&parent[u]), pv = seen; // (v's parent is known now)
    while (pu != pv) {
        if (pu < pv) { // u is the side whose parent is the larger
            uint32_t t = u; u = v; v = t;
            t = pu; pu = pv; pv = t;
        }
        if (u == pu) { // a root, as far as was seen: under the other side's parent (pv < u)
            const uint32_t old = atomicCAS(&parent[u], u, pv);
            if (old == u) return;
            pu = old; // hooked by somebody else meanwhile: that is where it points now
        } else { // climb
            u = pu;
            pu = ld_parent(&parent[u]);
        }
    }
}

} // namespace
} // namespace umihip

// ---- min-label hooks onto hot words (cc_hook / dag_hook rounds) ---------------------------------
namespace umihip {
namespace {

// atomicMin(&arr[idx], val) for the lanes with `todo`, all lanes of the wave calling together.
// A single word takes ~90 atomics per microsecond, and hooks pile up on the roots of the big
// trees: the lanes that share the first pending lane's target send one atomic for their
// minimum, twice; what is left goes one by one.
// A wave keeps the minimum for its hottest target (the leader target of its last trip: the root of
// the giant component for most of them) in registers across the trips of its loop and sends it
// once, when the target changes or the kernel ends.
struct HotMin {
    uint32_t t = 0xFFFFFFFFu, m = 0xFFFFFFFFu; // (wave-uniform)
};
__device__ __forceinline__ void hot_flush(uint32_t *arr, HotMin &h)
{
    // (a fresh look first: while this wave gathered, others have usually lowered the word)
    if (h.t != 0xFFFFFFFFu && (threadIdx.x & 63) == 0 && __atomic_load_n(&arr[h.t], __ATOMIC_RELAXED) > h.m)
        atomicMin(&arr[h.t], h.m);
    h.t = 0xFFFFFFFFu;
    h.m = 0xFFFFFFFFu;
}
__device__ __forceinline__ void wave_atomic_min(uint32_t *arr, uint32_t idx, uint32_t val, bool todo, HotMin &h)
{
    for (int pass = 0; pass < 2 && __any(todo); pass++) {
        const int leader = __ffsll((unsigned long long)__ballot(todo)) - 1;
        const uint32_t tgt = (uint32_t)__shfl((int)idx, leader);
        const bool mine = todo && idx == tgt;
        uint32_t m = mine ? val : 0xFFFFFFFFu;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off));
        if (tgt == h.t) {
            h.m = min(h.m, m);
        } else if (pass == 0) { // the new hot target
            hot_flush(arr, h);
            h.t = tgt;
            h.m = m;
        } else if ((int)(threadIdx.x & 63) == leader && __atomic_load_n(&arr[tgt], __ATOMIC_RELAXED) > m) {
            atomicMin(&arr[tgt], m);
        }
        todo = todo && !mine;
    }
    if (todo) atomicMin(&arr[idx], val);
}

} // namespace
} // namespace umihip
