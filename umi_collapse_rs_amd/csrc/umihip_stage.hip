// Read staging on the device (gfx950): from the reads of a file to the batched hot path's input.
//
// What it replaces in the reference (tkob-vh/umi-collapse-rs): the per-read part of
// DeduplicateSAM::deduplicate_and_merge (src/deduplicate_sam.rs:148-176) -- to_bitset
// (src/utils/mod.rs:63-83, codes src/utils/read.rs:23-31), the per-position map UMI -> ReadFreq
// with its Vacant / Occupied arms (:161-175: freq += 1, the kept read chosen by Merge,
// src/merge/mod.rs:18-51) -- and the order Directional / Adjacency::apply put a position's UMIs in
// (stable by freq descending, src/algo/directional.rs:67-72).  The reference walks the reads one by
// one through two HashMaps; here
//   1. the reads are sorted by (alignment key, UMI), file index as the tie-break: ONE stable radix
//      sort on a composed 64-bit key where alignment bits + 3 bits per base fit a word (every
//      BASELINE config), else a sort per key word, least significant first (umihip_radix.hip).  The
//      kernel that encodes the UMIs also counts the digits of every pass of that sort;
//   2. equal neighbours are the Occupied arm.  Two sweeps over the sorted keys -- head counts per
//      tile, then (after a scan of the tile counts) every head's entry and position number -- give
//      per entry its place, first read, position and key, per position its first read (a minimum
//      per tile in LDS, one global atomic per tile and position);
//   3. what the maps' iteration order leaves open is fixed the canonical way:
//      positions by first appearance in the file, UMIs of a position by freq descending, ties by
//      first appearance.  First appearance needs no sort: one flag byte per read in FILE order
//      (entry's first read / position's first read), counted per 64 reads and scanned, tells every
//      entry its rank; the entries, put in that order, are then radix-sorted by (position rank,
//      max freq - freq) alone -- 32-bit keys where that fits.
// Keys of several words (umi_len 22..85) take the same route with W words per key.
// Integer / byte work, HBM streams; no MFMA.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "umihip_internal.h"

namespace umihip {

namespace {

inline uint32_t grid_for(uint64_t n, int block = 256, uint32_t cap = 4096)
{
    uint64_t g = (n + block - 1) / block;
    return (uint32_t)(g < 1 ? 1 : (g > cap ? cap : g));
}

enum StageCounter : int { SC_BAD = 0, SC_ENTRIES = 1, SC_BUCKETS = 2, SC_FMAX = 3, SC_PMAX = 4, SC_COUNT = 5 };

// ---- 1. encode ------------------------------------------------------------------------------------
// Four bases at a time: the ASCII codes of A C G T N differ in bits 1..3 (A 000, C 001, T 010, G 011,
// N 111), from which the reference's 3-bit codes (src/utils/read.rs:23-31: A 000, T 101, C 110, G 011,
// N 100) are three bitwise expressions per byte lane; the other five bits of each byte are checked
// against what the letter demands (anything else is the reference's panic "Unknown character in UMI
// sequence", src/utils/mod.rs:77-79).  Returns the twelve code bits, base 0 lowest; *bad != 0 if a
// byte is none of the five letters.
__device__ __forceinline__ uint32_t encode4(uint32_t w, uint32_t *bad)
{
    const uint32_t m = 0x01010101u;
    const uint32_t x0 = (w >> 1) & m, x1 = (w >> 2) & m, x2 = (w >> 3) & m;
    const uint32_t c0 = x1 & ~x2, c1 = x0 & ~x2, c2 = (x0 ^ x1) | x2;
    const uint32_t is_t = x1 & ~x0 & ~x2, is_n = x0 & x1 & x2;
    *bad |= ((w ^ ~(is_t | is_n)) & m) | (((w >> 4) ^ is_t) & m) | ((w >> 5) & m) | (~(w >> 6) & m) | ((w >> 7) & m) |
            (x2 & ~(x0 & x1));
    const uint32_t v = c0 | (c1 << 1) | (c2 << 2); // a code in the low bits of every byte
    const uint32_t t = v | (v >> 5);               // bases 0,1 at bits 0..5, bases 2,3 at bits 16..21
    return (t & 0x3Fu) | ((t >> 10) & 0xFC0u);
}

// The sort of the reads only has to bring equal (alignment, UMI) together, so the UMI rides in the
// composed key at seven bits per three bases (5^3 = 125 letters' worth) instead of nine -- 12 bases:
// 28 bits, not 36, a radix pass less -- codes 0 3 4 5 6 -> digits 0 1 2 3 4, a group = d0 + 5 d1 +
// 25 d2, groups in base order from bit 0; a last group of one / two bases takes three / five bits.
struct Pack5 {
    int len, bits;
};
inline Pack5 pack5_of(int umi_len)
{
    Pack5 p;
    p.len = umi_len;
    p.bits = 7 * (umi_len / 3) + (umi_len % 3 == 1 ? 3 : umi_len % 3 == 2 ? 5 : 0);
    return p;
}
__device__ __forceinline__ uint64_t pack5(uint64_t key3, const Pack5 &p)
{
    uint64_t out = 0;
    for (int g = 0; 3 * g < p.len; g++) {
        const uint32_t c = (uint32_t)(key3 >> (9 * g)) & 0x1FFu; // (bases past the end: code 0)
        const uint32_t c0 = c & 7u, c1 = (c >> 3) & 7u, c2 = c >> 6;
        const uint32_t v = (c0 ? c0 - 2u : 0u) + 5u * (c1 ? c1 - 2u : 0u) + 25u * (c2 ? c2 - 2u : 0u);
        out |= (uint64_t)v << (7 * g);
    }
    return out;
}
__device__ __forceinline__ uint64_t unpack5(uint64_t k5, const Pack5 &p)
{
    uint64_t key3 = 0;
    for (int g = 0; 3 * g < p.len; g++) {
        const uint32_t v = (uint32_t)(k5 >> (7 * g)) & 0x7Fu;
        const uint32_t d2 = (v * 41u) >> 10, r = v - 25u * d2; // v / 25 for v < 125
        const uint32_t d1 = (r * 13u) >> 6, d0 = r - 5u * d1;   // r / 5 for r < 25
        const uint32_t c = (d0 ? d0 + 2u : 0u) | ((d1 ? d1 + 2u : 0u) << 3) | ((d2 ? d2 + 2u : 0u) << 6);
        key3 |= (uint64_t)c << (9 * g);
    }
    return key3;
}

// One count per lane into an LDS histogram.  Where the input is all but sorted already (the high
// digits of an alignment key, a position's rank) whole waves carry one digit, and atomics of one
// wave on one LDS word take turns: the lanes that share the first lane's digit add up once if they
// are many.
__device__ __forceinline__ void hist_add(uint32_t *h, uint32_t d)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(1);
    while (todo) {
        const int first = __builtin_ctzll(todo);
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
        const unsigned long long same = __ballot(d == d0) & todo;
        if (__builtin_popcountll(same) < 8) break;
        if (lane == first) atomicAdd(&h[d0], (uint32_t)__builtin_popcountll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&h[d], 1u);
}

// k3[i * W + w] = word w of read i's UMI key (base b at bits 3b .. 3b+2 of the word string,
// utils/mod.rs:38-41, bitset.rs:52-61); idx[i] = i; composed (may be null): the sort key
// (alignment << UMI bits | UMI five-to-a-base) where that fits 64 bits, and hist_parts (with it):
// this block's digit counts of the radix passes over bits [0, hist_bits) of that key.  A block takes
// 256 reads at a time: their text comes in as whole words, lane after lane, and is handed out
// through LDS (a read's umi_len bytes straight from memory are umi_len/4 loads of 4 bytes out of
// every 12: 0.21 ms at 10^7 reads); the words of the chunk after next are asked for before the chunk
// in hand is worked on.
template <int W>
__global__ __launch_bounds__(256) void stage_encode_kernel(const uint8_t *__restrict__ umi, const uint64_t *__restrict__ align,
                                                           uint32_t n, int umi_len, int align_bits,
                                                           uint64_t *__restrict__ k3, uint32_t *__restrict__ idx,
                                                           uint64_t *__restrict__ composed, Pack5 p5,
                                                           unsigned long long *__restrict__ counters,
                                                           uint32_t *__restrict__ hist_parts, int hist_bits)
{
    extern __shared__ uint32_t text[]; // 256 reads' UMIs, then the histograms
    constexpr int PW = W == 1 ? 6 : 22; // words of a chunk per thread, at most (umi_len <= 21 / 85)
    const int words_per_chunk = 64 * umi_len; // (256 * umi_len bytes)
    uint32_t *h = text + words_per_chunk;
    const int n_passes = hist_parts ? (hist_bits + 7) / 8 : 0;
    for (int i = threadIdx.x; i < n_passes * RADIX_BINS; i += 256) h[i] = 0;
    uint32_t bad = 0;
    const bool aligned = ((uintptr_t)umi & 3) == 0;
    const uint64_t total_bytes = (uint64_t)n * umi_len;
    const uint32_t n_chunks = (n + 255u) / 256u;
    // a chunk whose text lies inside the array as whole words comes through registers, a step ahead
    auto whole = [&](uint32_t c) { return aligned && c < n_chunks && ((uint64_t)c + 1) * 256u * umi_len <= total_bytes; };
    uint32_t pre[PW];
    auto fetch = [&](uint32_t c) {
        if (!whole(c)) return;
        const uint32_t *src = (const uint32_t *)(umi + (uint64_t)c * 256u * umi_len);
#pragma unroll
        for (int q = 0; q < PW; q++)
            if (q * 256 + (int)threadIdx.x < words_per_chunk) pre[q] = src[q * 256 + threadIdx.x];
    };
    fetch(blockIdx.x);
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint64_t byte0 = (uint64_t)c * 256u * umi_len;
        __syncthreads(); // (the chunk before has been read; the histograms are zero)
        if (whole(c)) {
#pragma unroll
            for (int q = 0; q < PW; q++)
                if (q * 256 + (int)threadIdx.x < words_per_chunk) text[q * 256 + threadIdx.x] = pre[q];
        } else { // the file's last reads, or text that does not start on a word: byte by byte
            uint8_t *tb = (uint8_t *)text;
            for (int j = threadIdx.x; j < 4 * words_per_chunk; j += 256)
                tb[j] = byte0 + j < total_bytes ? umi[byte0 + j] : (uint8_t)'A';
        }
        fetch(c + gridDim.x);
        __syncthreads();
        const uint32_t i = c * 256u + threadIdx.x;
        if (i >= n) continue;
        const uint8_t *mine = (const uint8_t *)text + (size_t)threadIdx.x * umi_len;
        uint64_t key[W];
#pragma unroll
        for (int w = 0; w < W; w++) key[w] = 0;
        for (int b = 0; b < umi_len; b += 4) {
            uint32_t word;
            if ((umi_len & 3) == 0) {
                word = *(const uint32_t *)(mine + b);
            } else {
                word = 0x41414141u; // ('A' where the UMI has ended: code 0)
                for (int q = 0; q < 4 && b + q < umi_len; q++)
                    word = (word & ~(0xFFu << (8 * q))) | ((uint32_t)mine[b + q] << (8 * q));
            }
            const uint64_t twelve = encode4(word, &bad);
            const int bit = 3 * b, w = bit >> 6, sh = bit & 63;
#pragma unroll
            for (int q = 0; q < W; q++) {
                if (q == w) key[q] |= twelve << sh;
                if (q == w + 1 && sh > 52) key[q] |= twelve >> (64 - sh); // a window across two words
            }
        }
        idx[i] = i;
        if (composed) { // (the UMI key is the low part of it: no array of its own)
            const uint64_t a = align_bits >= 64 ? align[i] : align[i] & ((1ull << align_bits) - 1ull);
            const uint64_t ck = (p5.bits >= 64 ? 0ull : a << p5.bits) | pack5(key[0], p5);
            composed[i] = ck;
            for (int p = 0; p < n_passes; p++) {
                const int sh = 8 * p, bits = min(8, hist_bits - sh);
                hist_add(h + p * RADIX_BINS, (uint32_t)(ck >> sh) & ((1u << bits) - 1u));
            }
        } else {
#pragma unroll
            for (int w = 0; w < W; w++) k3[(size_t)i * W + w] = key[w];
        }
    }
    __syncthreads();
    if (n_passes) { // (every block writes its slice whole: the sort adds the slices up)
        uint32_t *out = hist_parts + (size_t)blockIdx.x * n_passes * RADIX_BINS;
        for (int i = threadIdx.x; i < n_passes * RADIX_BINS; i += 256) out[i] = h[i];
    }
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicAdd(&counters[SC_BAD], 1ull);
}

// out[i] = src[pos[i] * stride + word]
__global__ __launch_bounds__(256) void stage_gather_u64_kernel(const uint64_t *__restrict__ src, int stride, int word,
                                                               const uint32_t *__restrict__ pos, uint32_t n,
                                                               uint64_t *__restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = src[(size_t)pos[i] * stride + word];
}

// ---- 2. entries and positions -----------------------------------------------------------------------
// Reads in (alignment key, UMI, file index) order: does a new UMI entry (bit 0) / a new position
// (bit 1) begin at read i?  composed != null: the sorted composed keys say it all; else the reads'
// keys are looked up through the permutation.
struct SortedReads {
    const uint64_t *composed;
    Pack5 p5;
    int umi_bits; // (of the composed key: p5.bits)
    const uint64_t *align;
    int align_bits;
    const uint64_t *k3;
    const uint32_t *perm;
    uint32_t n;
};
template <int W> __device__ __forceinline__ uint32_t head_bits(const SortedReads &s, uint32_t i)
{
    if (i == 0) return 3u;
    if (s.composed) {
        const uint64_t a = s.composed[i], b = s.composed[i - 1];
        const bool bh = s.umi_bits >= 64 ? false : (a >> s.umi_bits) != (b >> s.umi_bits);
        return (a != b ? 1u : 0u) | (bh ? 2u : 0u);
    }
    const uint32_t r = s.perm[i], q = s.perm[i - 1];
    const uint64_t am = s.align_bits >= 64 ? ~0ull : (1ull << s.align_bits) - 1ull;
    const bool bh = (s.align[r] & am) != (s.align[q] & am);
    bool h = bh;
#pragma unroll
    for (int w = 0; w < W; w++) h = h || s.k3[(size_t)r * W + w] != s.k3[(size_t)q * W + w]; // BitSet equality is on the
                                                                                             // bits alone (bitset.rs:94-101)
    return (h ? 1u : 0u) | (bh ? 2u : 0u);
}

// A tile = 2,048 consecutive sorted reads = 8 rounds of a 256-thread block; a cell = the 64 reads of
// one wave in one round.  Heads are counted with ballots, the two counts ride in one 64-bit word
// (entries low, positions high).
constexpr int HT_ROUNDS = 8, HT_TILE = 256 * HT_ROUNDS, HT_CELLS = 4 * HT_ROUNDS;

template <int W>
__global__ __launch_bounds__(256) void stage_head_sums_kernel(SortedReads s, unsigned long long *__restrict__ sums)
{
    __shared__ unsigned long long wsum[4];
    const uint32_t base = blockIdx.x * HT_TILE;
    unsigned long long mine = 0;
#pragma unroll
    for (int r = 0; r < HT_ROUNDS; r++) {
        const uint32_t i = base + r * 256 + threadIdx.x;
        const uint32_t hb = i < s.n ? head_bits<W>(s, i) : 0u;
        const unsigned long long eb = __ballot(hb & 1u), pb = __ballot(hb & 2u);
        mine += (unsigned long long)__builtin_popcountll(eb) | ((unsigned long long)__builtin_popcountll(pb) << 32);
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine; // (wave-uniform)
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// sums: exclusive over the tiles (scan_spine_u64).  Per head read: its entry's place in the sorted
// order, first read (the sort is stable: the head is the entry's first read in the file), position
// number and key; per position the smallest file index among its entries' first reads; the counts
// for the host.  numbers (may be null): every read's entry number, for the Merge pass.
template <int W>
__global__ __launch_bounds__(256) void stage_head_apply_kernel(SortedReads s, const unsigned long long *__restrict__ sums,
                                                               uint32_t *__restrict__ head_pos,
                                                               uint32_t *__restrict__ ent_first,
                                                               uint32_t *__restrict__ ent_bseq,
                                                               uint64_t *__restrict__ ent_key,
                                                               uint32_t *__restrict__ bfirst, uint32_t *__restrict__ numbers,
                                                               uint32_t *__restrict__ pos_start,
                                                               unsigned long long *__restrict__ counters)
{
    __shared__ unsigned long long cell[HT_CELLS];
    __shared__ uint32_t pos_min[HT_TILE + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t base = blockIdx.x * HT_TILE;
    for (int j = tid; j < HT_TILE + 1; j += 256) pos_min[j] = 0xFFFFFFFFu;
    unsigned long long eb[HT_ROUNDS], pb[HT_ROUNDS];
#pragma unroll
    for (int r = 0; r < HT_ROUNDS; r++) {
        const uint32_t i = base + r * 256 + tid;
        const uint32_t hb = i < s.n ? head_bits<W>(s, i) : 0u;
        eb[r] = __ballot(hb & 1u);
        pb[r] = __ballot(hb & 2u);
        if (lane == 0)
            cell[r * 4 + wave] =
                (unsigned long long)__builtin_popcountll(eb[r]) | ((unsigned long long)__builtin_popcountll(pb[r]) << 32);
    }
    __syncthreads();
    // exclusive prefix over the tile's cells, in the reads' order (round-major), by every wave for itself
    unsigned long long incl = lane < HT_CELLS ? cell[lane] : 0ull;
    const unsigned long long own = incl;
    for (int o = 1; o < HT_CELLS; o <<= 1) {
        const unsigned long long up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const unsigned long long excl = incl - own, before_tile = sums[blockIdx.x];
    // (local position numbers start at the position of the read before the tile: the number of position
    // heads before the tile, less one)
    const uint32_t b_base = (uint32_t)(before_tile >> 32) - 1u;
    const unsigned long long upto = (2ull << lane) - 1ull; // this lane and the ones before it
#pragma unroll
    for (int r = 0; r < HT_ROUNDS; r++) {
        const uint32_t i = base + r * 256 + tid;
        const unsigned long long at_cell = before_tile + __shfl(excl, r * 4 + wave);
        const uint32_t e_incl = (uint32_t)at_cell + (uint32_t)__builtin_popcountll(eb[r] & upto);
        const uint32_t b_incl = (uint32_t)(at_cell >> 32) + (uint32_t)__builtin_popcountll(pb[r] & upto);
        if (i < s.n) {
            const uint32_t e = e_incl - 1u, b = b_incl - 1u;
            if (numbers) numbers[i] = e;
            if ((eb[r] >> lane) & 1ull) {
                const uint32_t rd = s.perm[i];
                head_pos[e] = i;
                ent_first[e] = rd;
                ent_bseq[e] = b;
                if ((pb[r] >> lane) & 1ull) pos_start[b] = e; // (a position's first read is an entry's first)
                if (s.composed) { // (W == 1: the UMI key is the low part of the sorted composed key)
                    ent_key[e] = s.umi_bits >= 64 ? s.composed[i] : s.composed[i] & ((1ull << s.umi_bits) - 1ull); // (still packed)
                } else {
#pragma unroll
                    for (int w = 0; w < W; w++) ent_key[(size_t)e * W + w] = s.k3[(size_t)rd * W + w];
                }
                // (head lanes of one position that meet in a round take turns on its word; reducing them
                // in the wave first, six shuffles a run, was measured slower: 105 against 66 us at 10^7 reads)
                atomicMin(&pos_min[b - b_base], rd);
            }
            if (i == s.n - 1) {
                head_pos[e_incl] = s.n;
                pos_start[b_incl] = e_incl;
                counters[SC_ENTRIES] = e_incl;
                counters[SC_BUCKETS] = b_incl;
            }
        }
    }
    __syncthreads();
    // one look and at most one atomic per position of the tile (a deep position is hundreds of tiles on
    // one word, which takes ~90 accesses per microsecond)
    for (int j = tid; j < HT_TILE + 1; j += 256) {
        const uint32_t m = pos_min[j];
        if (m != 0xFFFFFFFFu) {
            uint32_t *g = &bfirst[b_base + (uint32_t)j];
            if (__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > m) atomicMin(g, m);
        }
    }
}

// block maximum of m to counters[which]: one atomic per block, and a look first
__device__ __forceinline__ void block_max_to(uint32_t m, unsigned long long *word)
{
    __shared__ uint32_t wmax[4];
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, off));
    __syncthreads(); // (wmax may be in use by the call before)
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (m && __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)m)
            atomicMax(word, (unsigned long long)m);
    }
}

// the largest freq of any entry (the width of the freq field of the final sort key) and the most
// entries of any position (whether a wave can order a position by itself)
__global__ __launch_bounds__(256) void stage_fmax_kernel(const uint32_t *__restrict__ head_pos,
                                                         const uint32_t *__restrict__ pos_start,
                                                         unsigned long long *__restrict__ counters)
{
    const uint32_t n_entries = (uint32_t)counters[SC_ENTRIES], n_buckets = (uint32_t)counters[SC_BUCKETS];
    uint32_t pm = 0;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += gridDim.x * blockDim.x)
        pm = max(pm, pos_start[b + 1] - pos_start[b]);
    block_max_to(pm, &counters[SC_PMAX]);
    uint32_t m = 0;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x)
        m = max(m, head_pos[e + 1] - head_pos[e]);
    block_max_to(m, &counters[SC_FMAX]);
}

// The maximum of v over the lanes of a wave that share a run of equal ids, at the last lane of each
// run (ids are non-decreasing across the lanes): a segmented scan by doubling.
__device__ __forceinline__ unsigned long long run_max(unsigned long long v, uint32_t id, bool valid, bool *is_last)
{
    const int lane = threadIdx.x & 63;
    const uint32_t my = valid ? id : 0xFFFFFFFFu;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long ov = __shfl_up(v, off);
        const uint32_t oid = (uint32_t)__shfl_up((int)my, off);
        if (lane >= off && oid == my) v = ov > v ? ov : v;
    }
    const uint32_t nid = (uint32_t)__shfl_down((int)my, 1);
    *is_last = valid && (lane == 63 || nid != my);
    return v;
}

// Merge (merge/mod.rs:35,49): per entry the read with the higher score, the earlier one on a tie; one
// atomic per run of a wave
__global__ __launch_bounds__(256) void stage_best_kernel(const uint32_t *__restrict__ numbers,
                                                         const uint32_t *__restrict__ perm,
                                                         const int32_t *__restrict__ score, uint32_t n,
                                                         unsigned long long *__restrict__ best)
{
    const uint32_t n_round = (n + 63u) & ~63u; // (whole waves: the shuffles need every lane)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        const bool valid = i < n;
        const uint32_t e = valid ? numbers[i] : 0u;
        const uint32_t r = valid ? perm[i] : 0u;
        const unsigned long long packed =
            valid ? ((unsigned long long)((uint32_t)score[r] ^ 0x80000000u) << 32) | (unsigned long long)(0xFFFFFFFFu - r) : 0ull;
        bool last;
        const unsigned long long m = run_max(packed, e, valid, &last);
        if (last && __hip_atomic_load(&best[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < m) atomicMax(&best[e], m);
    }
}

// ---- 3. the canonical order ---------------------------------------------------------------------------
// one byte per read in FILE order: bit 0 at the first read of every entry, bit 1 where that read is
// its position's first as well (one writer per byte: a position's first read is an entry's first)
__global__ __launch_bounds__(256) void stage_mark_kernel(const uint32_t *__restrict__ ent_first,
                                                         const uint32_t *__restrict__ ent_bseq,
                                                         const uint32_t *__restrict__ bfirst, uint32_t n_entries,
                                                         uint8_t *__restrict__ file_flags)
{
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x) {
        const uint32_t r = ent_first[e];
        file_flags[r] = bfirst[ent_bseq[e]] == r ? 3u : 1u;
    }
}

// flags counted per group of 64 reads: rec[g] = the flags of the block's groups before g (entry
// flags low, position flags high) and the group's own flags as two 64-bit maps; block_sums[block] =
// the block's total (a block = 256 groups)
__device__ __forceinline__ unsigned long long flag_counts(uint64_t w)
{
    return (unsigned long long)__builtin_popcountll(w & 0x0101010101010101ull) |
           ((unsigned long long)__builtin_popcountll(w & 0x0202020202020202ull) << 32);
}
// bit 0 of every byte of w, as the eight bits of a byte (byte k -> bit k)
__device__ __forceinline__ uint64_t low_bit_of_bytes(uint64_t w)
{
    return ((w & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56;
}
struct __attribute__((aligned(32))) RankRec {
    unsigned long long pre;
    uint64_t entry_bits, position_bits, pad;
};
__global__ __launch_bounds__(256) void stage_rank_sums_kernel(const uint64_t *__restrict__ flag_words, uint32_t n_groups,
                                                              RankRec *__restrict__ rec,
                                                              unsigned long long *__restrict__ block_sums)
{
    __shared__ unsigned long long wsum[4];
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long mine = 0;
    uint64_t eb = 0, pb = 0;
    if (g < n_groups) {
        const ulonglong2 *p = (const ulonglong2 *)(flag_words + (size_t)g * 8);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const ulonglong2 v = p[q];
            mine += flag_counts(v.x) + flag_counts(v.y);
            eb |= (low_bit_of_bytes(v.x) | (low_bit_of_bytes(v.y) << 8)) << (16 * q);
            pb |= (low_bit_of_bytes(v.x >> 1) | (low_bit_of_bytes(v.y >> 1) << 8)) << (16 * q);
        }
    }
    unsigned long long incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long off = 0;
    for (int w = 0; w < wave; w++) off += wsum[w];
    if (g < n_groups) {
        RankRec o;
        o.pre = off + incl - mine;
        o.entry_bits = eb;
        o.position_bits = pb;
        o.pad = 0;
        rec[g] = o;
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = off + incl;
}

// how many entry firsts (low) / position firsts (high) lie before read r in the file
__device__ __forceinline__ unsigned long long file_rank(uint32_t r, const RankRec *__restrict__ rec,
                                                        const unsigned long long *__restrict__ block_pre)
{
    const uint32_t g = r >> 6;
    const ulonglong4 v = *(const ulonglong4 *)&rec[g];
    const uint64_t before = (1ull << (r & 63u)) - 1ull;
    return block_pre[g >> 8] + v.x + (unsigned long long)__builtin_popcountll(v.y & before) +
           ((unsigned long long)__builtin_popcountll(v.z & before) << 32);
}

// per position: its rank of first appearance
__global__ __launch_bounds__(256) void stage_position_rank_kernel(const uint32_t *__restrict__ bfirst, uint32_t n_buckets,
                                                                  const RankRec *__restrict__ rec,
                                                                  const unsigned long long *__restrict__ block_pre,
                                                                  uint32_t *__restrict__ brank_of)
{
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += gridDim.x * blockDim.x)
        brank_of[b] = (uint32_t)(file_rank(bfirst[b], rec, block_pre) >> 32);
}

// what the final gather needs of an entry, in one 32-byte line
struct __attribute__((aligned(32))) EntryRec {
    uint64_t key0;  // the first key word
    uint32_t first; // its first read
    uint32_t freq;
    uint32_t brank; // its position's rank of first appearance
    uint32_t pad[3];
};

// the entries in order of first appearance, each with the key of the one sort that is left:
// (position's rank of first appearance, max freq - freq); this block's digit counts of that sort
template <typename KeyT>
__global__ __launch_bounds__(256) void stage_order_kernel(const RankRec *__restrict__ rank_rec,
                                                          const unsigned long long *__restrict__ block_pre,
                                                          const uint32_t *__restrict__ ent_first,
                                                          const uint32_t *__restrict__ ent_bseq,
                                                          const uint32_t *__restrict__ brank_of,
                                                          const uint32_t *__restrict__ head_pos,
                                                          const uint64_t *__restrict__ ent_key, int key_words,
                                                          uint32_t n_entries, uint32_t fmax, int freq_bits,
                                                          KeyT *__restrict__ okey, uint32_t *__restrict__ oval,
                                                          EntryRec *__restrict__ rec, uint32_t *__restrict__ hist_parts,
                                                          int hist_bits, Pack5 p5, int packed)
{
    __shared__ uint32_t h[RADIX_MAX_PASSES * RADIX_BINS];
    const int n_passes = (hist_bits + 7) / 8;
    for (int i = threadIdx.x; i < n_passes * RADIX_BINS; i += 256) h[i] = 0;
    __syncthreads();
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x) {
        const uint32_t r = ent_first[e];
        const uint32_t at = (uint32_t)file_rank(r, rank_rec, block_pre); // rank of the entry's first read
        const uint32_t br = brank_of[ent_bseq[e]];
        const uint32_t f = head_pos[e + 1] - head_pos[e];
        const uint64_t k = ((uint64_t)br << freq_bits) | (uint64_t)(fmax - f);
        okey[at] = (KeyT)k;
        oval[at] = e;
        EntryRec o;
        o.key0 = packed ? unpack5(ent_key[e], p5) : ent_key[(size_t)e * key_words];
        o.first = r;
        o.freq = f;
        o.brank = br;
        o.pad[0] = o.pad[1] = o.pad[2] = 0;
        rec[e] = o;
        for (int p = 0; p < n_passes; p++) {
            const int sh = 8 * p, bits = min(8, hist_bits - sh);
            hist_add(h + p * RADIX_BINS, (uint32_t)(k >> sh) & ((1u << bits) - 1u));
        }
    }
    __syncthreads();
    uint32_t *out = hist_parts + (size_t)blockIdx.x * n_passes * RADIX_BINS;
    for (int i = threadIdx.x; i < n_passes * RADIX_BINS; i += 256) out[i] = h[i];
}

// bit 2 set and bits 0, 1 clear: the N code; its three bits go to the mask (utils/mod.rs:45-50,74-76)
__device__ __forceinline__ uint64_t nmask_of(uint64_t key)
{
    const uint64_t b2 = key & 0x4924924924924924ull;
    const uint64_t n = b2 & ~((key << 1) | (key << 2)); // bit 2 of the N bases
    return n | (n >> 1) | (n >> 2);
}

// the entries in their final order, and the table of the positions (a position's entries are
// neighbours, the positions come by rank: its first entry writes its offset)
template <int W>
__global__ __launch_bounds__(256) void stage_emit_kernel(const uint32_t *__restrict__ perm, uint32_t n_entries,
                                                         uint32_t n_buckets, int merge, int umi_len,
                                                         const EntryRec *__restrict__ rec,
                                                         const uint64_t *__restrict__ ent_key,
                                                         const unsigned long long *__restrict__ best,
                                                         uint64_t *__restrict__ keys, uint64_t *__restrict__ nmask,
                                                         int32_t *__restrict__ freq, uint64_t *__restrict__ rep,
                                                         uint64_t *__restrict__ bucket_off)
{
    const uint32_t n_round = (n_entries + 63u) & ~63u; // (whole waves: the neighbour's rank comes by DPP)
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_round; j += gridDim.x * blockDim.x) {
        const bool valid = j < n_entries;
        const uint32_t e = valid ? perm[j] : 0u;
        ulonglong4 raw{0, 0, 0, 0};
        if (valid) raw = *(const ulonglong4 *)&rec[e];
        const uint32_t first = (uint32_t)raw.y, f = (uint32_t)(raw.y >> 32), br = (uint32_t)raw.z;
        // the rank of the entry before: the lane before's, or (lane 0) one more look
        uint32_t br_before = 0xFFFFFFFFu;
        if ((threadIdx.x & 63) == 0 && valid && j > 0) br_before = rec[perm[j - 1]].brank;
        br_before = (uint32_t)__builtin_amdgcn_update_dpp((int)br_before, (int)br, 0x138, 0xF, 0xF, false);
        if (!valid) continue;
        uint64_t key[W];
        key[0] = raw.x;
#pragma unroll
        for (int w = 1; w < W; w++) key[w] = ent_key[(size_t)e * W + w];
#pragma unroll
        for (int w = 0; w < W; w++) keys[(size_t)j * W + w] = key[w];
        if (nmask) {
            if (W == 1) {
                nmask[j] = nmask_of(key[0]);
            } else { // base by base: one may sit across two words (set_n_bit, bitset.rs:63-75)
                uint64_t m[W];
#pragma unroll
                for (int w = 0; w < W; w++) m[w] = 0;
                for (int b = 0; b < umi_len; b++) {
                    const int bit = 3 * b, w = bit >> 6, sh = bit & 63;
                    uint64_t c = 0;
#pragma unroll
                    for (int q = 0; q < W; q++) {
                        if (q == w) c |= key[q] >> sh;
                        if (q == w + 1 && sh > 61) c |= key[q] << (64 - sh);
                    }
                    if ((c & 7ull) == 4ull) {
#pragma unroll
                        for (int q = 0; q < W; q++) {
                            if (q == w) m[q] |= 7ull << sh;
                            if (q == w + 1 && sh > 61) m[q] |= 7ull >> (64 - sh);
                        }
                    }
                }
#pragma unroll
                for (int w = 0; w < W; w++) nmask[(size_t)j * W + w] = m[w];
            }
        }
        freq[j] = (int32_t)f;
        rep[j] = merge ? (uint64_t)(0xFFFFFFFFu - (uint32_t)best[e]) : (uint64_t)first;
        if (j == 0 || br_before != br) bucket_off[br] = j;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bucket_off[n_buckets] = n_entries;
}

// ---- 3'. the canonical order where no position has more than LOCAL_MAX entries: no sort --------------
// A position's entries are neighbours already (the reads were sorted by alignment key); what is left is
// their order among themselves -- freq descending, ties by first appearance -- and where the position
// starts in the output: the sizes of the positions in rank order, scanned.  A wave takes a position:
// its (max freq - freq, first read) keys go to LDS, every lane counts the keys below its own (all
// pairs: a position of 60 entries is 60 broadcast reads), and writes its entry where it belongs.  At
// 10^7 reads in 10^5 positions this replaces the order kernel, three radix passes and the gather.
constexpr uint32_t LOCAL_MAX = 1024;
__global__ __launch_bounds__(256) void stage_rank_sizes_kernel(const uint32_t *__restrict__ pos_start,
                                                               const uint32_t *__restrict__ brank_of, uint32_t n_buckets,
                                                               uint64_t *__restrict__ size_by_rank)
{
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += gridDim.x * blockDim.x)
        size_by_rank[brank_of[b]] = pos_start[b + 1] - pos_start[b];
}

// OrdT: uint32_t where (max freq - freq) and the first read's index fit 32 bits together (first_bits for
// the index), else uint64_t (the index in the low word)
template <int W, typename OrdT>
__global__ __launch_bounds__(256) void stage_emit_local_kernel(const uint32_t *__restrict__ pos_start,
                                                               const uint32_t *__restrict__ brank_of,
                                                               const uint64_t *__restrict__ end_by_rank, // inclusive scan of the sizes
                                                               uint32_t n_entries, uint32_t n_buckets, int merge, int umi_len,
                                                               uint32_t fmax, int first_bits, Pack5 p5, int packed,
                                                               const uint64_t *__restrict__ ent_key,
                                                               const uint32_t *__restrict__ ent_first,
                                                               const uint32_t *__restrict__ head_pos,
                                                               const unsigned long long *__restrict__ best,
                                                               uint64_t *__restrict__ keys, uint64_t *__restrict__ nmask,
                                                               int32_t *__restrict__ freq, uint64_t *__restrict__ rep,
                                                               uint64_t *__restrict__ bucket_off)
{
    constexpr bool NARROW = sizeof(OrdT) == 4;
    constexpr uint32_t PER_READ = 16 / sizeof(OrdT); // keys per 16-byte LDS read
    __shared__ __attribute__((aligned(16))) OrdT order_key[4][LOCAL_MAX + PER_READ];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    OrdT *mine = order_key[wave];
    const uint32_t n_waves = gridDim.x * 4;
    const uint32_t first_mask = NARROW ? (first_bits >= 32 ? 0xFFFFFFFFu : (1u << first_bits) - 1u) : 0xFFFFFFFFu;
    for (uint32_t b = blockIdx.x * 4 + wave; b < n_buckets; b += n_waves) {
        const uint32_t s0 = pos_start[b], cnt = pos_start[b + 1] - s0, br = brank_of[b];
        const uint64_t first_out = end_by_rank[br] - cnt;
        if (lane == 0) bucket_off[br] = first_out;
        const uint32_t padded = (cnt + PER_READ - 1) / PER_READ * PER_READ;
        for (uint32_t i = lane; i < padded; i += 64) {
            const uint32_t e = s0 + i;
            OrdT v = (OrdT)~(OrdT)0; // (padding: below no key)
            if (i < cnt) {
                const uint32_t down = fmax - (head_pos[e + 1] - head_pos[e]);
                v = NARROW ? (OrdT)((down << first_bits) | ent_first[e]) : (OrdT)(((unsigned long long)down << 32) | ent_first[e]);
            }
            mine[i] = v;
        }
        __builtin_amdgcn_wave_barrier(); // (a wave's LDS accesses complete in order)
        for (uint32_t i0 = 0; i0 < cnt; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool valid = i < cnt;
            const OrdT k = valid ? mine[i] : (OrdT)0;
            uint32_t below = 0;
            for (uint32_t j = 0; j < padded; j += PER_READ) { // (one address for the wave: a broadcast of 16 bytes)
                if (NARROW) {
                    const uint4 q = *(const uint4 *)&mine[j];
                    below += (q.x < k ? 1u : 0u) + (q.y < k ? 1u : 0u) + (q.z < k ? 1u : 0u) + (q.w < k ? 1u : 0u);
                } else {
                    const ulonglong2 q = *(const ulonglong2 *)&mine[j];
                    below += (q.x < k ? 1u : 0u) + (q.y < k ? 1u : 0u);
                }
            }
            if (!valid) continue;
            const uint32_t e = s0 + i;
            const uint64_t out = first_out + below;
            uint64_t key[W];
            if (packed) {
                key[0] = unpack5(ent_key[e], p5);
            } else {
#pragma unroll
                for (int w = 0; w < W; w++) key[w] = ent_key[(size_t)e * W + w];
            }
#pragma unroll
            for (int w = 0; w < W; w++) keys[out * W + w] = key[w];
            if (nmask) {
                if (W == 1) {
                    nmask[out] = nmask_of(key[0]);
                } else {
                    uint64_t m[W];
#pragma unroll
                    for (int w = 0; w < W; w++) m[w] = 0;
                    for (int bb = 0; bb < umi_len; bb++) {
                        const int bit = 3 * bb, w = bit >> 6, sh = bit & 63;
                        uint64_t c = 0;
#pragma unroll
                        for (int q = 0; q < W; q++) {
                            if (q == w) c |= key[q] >> sh;
                            if (q == w + 1 && sh > 61) c |= key[q] << (64 - sh);
                        }
                        if ((c & 7ull) == 4ull) {
#pragma unroll
                            for (int q = 0; q < W; q++) {
                                if (q == w) m[q] |= 7ull << sh;
                                if (q == w + 1 && sh > 61) m[q] |= 7ull >> (64 - sh);
                            }
                        }
                    }
#pragma unroll
                    for (int w = 0; w < W; w++) nmask[out * W + w] = m[w];
                }
            }
            freq[out] = (int32_t)(fmax - (uint32_t)(NARROW ? (uint64_t)k >> first_bits : (uint64_t)k >> 32));
            rep[out] = merge ? (uint64_t)(0xFFFFFFFFu - (uint32_t)best[e]) : (uint64_t)((uint32_t)k & first_mask);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bucket_off[n_buckets] = n_entries;
}

inline int bits_for(uint64_t v)
{
    int b = 1;
    while (b < 64 && (v >> b)) b++;
    return b;
}

struct Carver {
    char *p;
    size_t off = 0;
    template <class T> T *take(size_t count)
    {
        T *r = (T *)(p ? p + off : nullptr);
        off = (off + count * sizeof(T) + 255) & ~(size_t)255;
        return r;
    }
};

struct StageBufs {
    uint64_t *k3, *keyA, *keyB, *ent_key, *file_flags;
    unsigned long long *best, *tile_sums;
    uint32_t *idxA, *idxB, *ent_first, *ent_bseq, *head_pos, *bfirst, *numbers, *brank_of, *pos_start;
    EntryRec *rec;
    RankRec *rank_rec;
    unsigned long long *counters;
    void *tmp;
    size_t tmp_bytes, total;
};

StageBufs carve(void *ws, uint32_t n, int n_words)
{
    Carver c{(char *)ws};
    StageBufs b;
    const size_t m = (size_t)n + 2;
    b.k3 = c.take<uint64_t>(m * n_words);
    b.ent_key = c.take<uint64_t>(m * n_words);
    b.keyA = c.take<uint64_t>(m);
    b.keyB = c.take<uint64_t>(m);
    b.best = c.take<unsigned long long>(m);
    b.rec = c.take<EntryRec>(m);
    b.file_flags = c.take<uint64_t>((m + 63) / 64 * 8 + 8);
    b.rank_rec = c.take<RankRec>((m + 63) / 64 + 1);
    b.tile_sums = c.take<unsigned long long>(m / HT_TILE + 2);
    b.idxA = c.take<uint32_t>(m);
    b.idxB = c.take<uint32_t>(m);
    b.ent_first = c.take<uint32_t>(m);
    b.ent_bseq = c.take<uint32_t>(m);
    b.head_pos = c.take<uint32_t>(m);
    b.bfirst = c.take<uint32_t>(m);
    b.numbers = c.take<uint32_t>(m);
    b.brank_of = c.take<uint32_t>(m);
    b.pos_start = c.take<uint32_t>(m);
    b.counters = c.take<unsigned long long>(SC_COUNT);
    b.tmp_bytes = std::max(radix_sort_temp_bytes(n), scan_temp_bytes(n)) + 256;
    b.tmp = c.take<char>(b.tmp_bytes);
    b.total = c.off;
    return b;
}

#define STAGE_TRY(x)                                   \
    do {                                               \
        const hipError_t e_ = (x);                     \
        if (e_ != hipSuccess) return -(int)e_;         \
    } while (0)

template <int W>
int stage_impl(void *workspace, const uint64_t *d_align, int align_bits, const uint8_t *d_umi, const int32_t *d_score,
               uint32_t n, int umi_len, int merge, uint64_t *d_keys, uint64_t *d_nmask, int32_t *d_freq, uint64_t *d_rep,
               uint64_t *d_bucket_off, uint64_t *n_entries_out, uint64_t *n_buckets_out, unsigned long long *h_pinned4,
               hipStream_t s)
{
    StageBufs b = carve(workspace, n, W);
    const bool use_score = merge != 0 && d_score != nullptr;
    const int umi_bits = 3 * umi_len;
    const Pack5 p5 = pack5_of(W == 1 ? umi_len : 1);
    const bool one_key = W == 1 && align_bits + p5.bits <= 64; // the composed sort key fits a word
    const int composed_bits = align_bits + p5.bits;
    STAGE_TRY(hipMemsetAsync(b.counters, 0, SC_COUNT * 8, s));
    STAGE_TRY(hipMemsetAsync(b.bfirst, 0xFF, (size_t)n * 4, s)); // (a position per read at most)
    uint32_t *hist = nullptr;
    if (one_key) STAGE_TRY(radix_sort_prepare(b.tmp, b.tmp_bytes, n, 0, composed_bits, &hist, s));
    const size_t encode_lds = (size_t)256 * umi_len + (size_t)RADIX_MAX_PASSES * RADIX_BINS * 4;
    const uint32_t encode_blocks = grid_for(n, 256, RADIX_HIST_PARTS);
    stage_encode_kernel<W><<<encode_blocks, 256, encode_lds, s>>>(d_umi, d_align, n, umi_len, align_bits, b.k3, b.idxA,
                                                                 one_key ? b.keyA : nullptr, p5, b.counters, hist, composed_bits);
    // ---- 1. reads by (alignment key, UMI, file index)
    uint64_t *ka = b.keyA, *kb = b.keyB;
    uint32_t *va = b.idxA, *vb = b.idxB;
    bool in_b = false;
    auto sort_by = [&](int begin_bit, int end_bit, uint32_t hist_parts) -> hipError_t {
        const hipError_t e = radix_sort_pairs_u64(ka, kb, va, vb, n, begin_bit, end_bit, b.tmp, b.tmp_bytes, &in_b, s, hist_parts);
        if (e == hipSuccess && in_b) {
            std::swap(ka, kb);
            std::swap(va, vb);
        }
        return e;
    };
    if (one_key) {
        STAGE_TRY(sort_by(0, composed_bits, encode_blocks));
    } else { // a stable sort per key word, least significant first: the UMI's words, then the alignment key
        for (int w = 0; w < W; w++) {
            stage_gather_u64_kernel<<<grid_for(n), 256, 0, s>>>(b.k3, W, w, va, n, ka);
            STAGE_TRY(sort_by(0, std::min(64, umi_bits - 64 * w), 0));
        }
        stage_gather_u64_kernel<<<grid_for(n), 256, 0, s>>>(d_align, 1, 0, va, n, ka);
        STAGE_TRY(sort_by(0, align_bits, 0));
    }
    // (va: the reads' file indices in order; ka: the composed keys in order where there is one)
    // ---- 2. entries and positions
    SortedReads sr;
    sr.composed = one_key ? ka : nullptr;
    sr.p5 = p5;
    sr.umi_bits = p5.bits;
    sr.align = d_align;
    sr.align_bits = align_bits;
    sr.k3 = b.k3;
    sr.perm = va;
    sr.n = n;
    const uint32_t tiles = (n + HT_TILE - 1) / HT_TILE;
    stage_head_sums_kernel<W><<<tiles, 256, 0, s>>>(sr, b.tile_sums);
    STAGE_TRY(scan_spine_u64(b.tile_sums, tiles, s));
    stage_head_apply_kernel<W><<<tiles, 256, 0, s>>>(sr, b.tile_sums, b.head_pos, b.ent_first, b.ent_bseq, b.ent_key, b.bfirst,
                                                    use_score ? b.numbers : nullptr, b.pos_start, b.counters);
    stage_fmax_kernel<<<grid_for(n, 256 * 16, 512), 256, 0, s>>>(b.head_pos, b.pos_start, b.counters);
    // the host needs the counts to size what follows (and the verdict on the characters)
    STAGE_TRY(hipMemcpyAsync(h_pinned4, b.counters, SC_COUNT * 8, hipMemcpyDeviceToHost, s));
    STAGE_TRY(hipStreamSynchronize(s));
    if (h_pinned4[SC_BAD]) return 1;
    const uint32_t E = (uint32_t)h_pinned4[SC_ENTRIES], B = (uint32_t)h_pinned4[SC_BUCKETS];
    const uint32_t fmax = (uint32_t)h_pinned4[SC_FMAX], pmax = (uint32_t)h_pinned4[SC_PMAX];
    if (use_score) {
        STAGE_TRY(hipMemsetAsync(b.best, 0, (size_t)E * 8, s));
        stage_best_kernel<<<grid_for(n), 256, 0, s>>>(b.numbers, va, d_score, n, b.best);
    }
    // ---- 3. the canonical order: first appearance by flag bytes in file order, counted and scanned ...
    const uint32_t n_groups = (n + 63u) / 64u, n_blocks = (n_groups + 255u) / 256u;
    STAGE_TRY(hipMemsetAsync(b.file_flags, 0, (size_t)n_groups * 64, s));
    stage_mark_kernel<<<grid_for(E), 256, 0, s>>>(b.ent_first, b.ent_bseq, b.bfirst, E, (uint8_t *)b.file_flags);
    stage_rank_sums_kernel<<<n_blocks, 256, 0, s>>>(b.file_flags, n_groups, b.rank_rec, b.tile_sums);
    STAGE_TRY(scan_spine_u64(b.tile_sums, n_blocks, s));
    stage_position_rank_kernel<<<grid_for(B), 256, 0, s>>>(b.bfirst, B, b.rank_rec, b.tile_sums, b.brank_of);
    // ---- 3'. every position fits a wave's LDS: ordered where it lies, no sort.  (A wave per position pays
    // where positions hold some tens of entries; a file of singletons -- shallow sequencing -- is 10^7
    // positions of one entry each, a trip to memory per wave and position: the sort below does not care.)
    if (pmax <= LOCAL_MAX && (uint64_t)E >= 16ull * B) {
        uint64_t *size_by_rank = b.keyA, *end_by_rank = b.keyB; // (the read sort's buffers are free)
        stage_rank_sizes_kernel<<<grid_for(B), 256, 0, s>>>(b.pos_start, b.brank_of, B, size_by_rank);
        STAGE_TRY(scan_inclusive_u64(size_by_rank, end_by_rank, B, b.tmp, b.tmp_bytes, s));
        const int first_bits = bits_for(n - 1);
        // (UMIHIP_STAGE_WIDE_ORDER: the tests' way to the 64-bit form, which otherwise needs 2^24 reads and a freq to match)
        static const bool force_wide = getenv("UMIHIP_STAGE_WIDE_ORDER") != nullptr;
        if (first_bits + bits_for(fmax) <= 32 && !force_wide)
            stage_emit_local_kernel<W, uint32_t><<<grid_for((uint64_t)B, 4, 8192), 256, 0, s>>>(
                b.pos_start, b.brank_of, end_by_rank, E, B, use_score ? 1 : 0, umi_len, fmax, first_bits, p5, one_key ? 1 : 0,
                b.ent_key, b.ent_first, b.head_pos, b.best, d_keys, d_nmask, d_freq, d_rep, d_bucket_off);
        else
            stage_emit_local_kernel<W, uint64_t><<<grid_for((uint64_t)B, 4, 4096), 256, 0, s>>>(
                b.pos_start, b.brank_of, end_by_rank, E, B, use_score ? 1 : 0, umi_len, fmax, 32, p5, one_key ? 1 : 0,
                b.ent_key, b.ent_first, b.head_pos, b.best, d_keys, d_nmask, d_freq, d_rep, d_bucket_off);
        STAGE_TRY(hipGetLastError());
        STAGE_TRY(hipStreamSynchronize(s));
        *n_entries_out = E;
        *n_buckets_out = B;
        return 0;
    }
    // ... and one stable sort of the entries, taken in that order, by (position rank, max freq - freq).
    // (Everything of the read sort but its order, va, is free by now -- and va's twin and the entry
    // numbers too, in stream order.)
    const int freq_bits = bits_for(fmax), rank_bits = bits_for(B ? B - 1 : 0), order_bits = std::min(64, freq_bits + rank_bits);
    uint32_t *ova = va == b.idxA ? b.idxB : b.idxA, *ovb = b.numbers;
    STAGE_TRY(radix_sort_prepare(b.tmp, b.tmp_bytes, E, 0, order_bits, &hist, s));
    bool ob = false;
    const uint32_t order_blocks = grid_for(E, 256, RADIX_HIST_PARTS);
    if (order_bits <= 32) {
        uint32_t *oka = (uint32_t *)b.keyA, *okb = (uint32_t *)b.keyB;
        stage_order_kernel<uint32_t><<<order_blocks, 256, 0, s>>>(b.rank_rec, b.tile_sums, b.ent_first, b.ent_bseq, b.brank_of,
                                                                 b.head_pos, b.ent_key, W, E, fmax, freq_bits, oka, ova, b.rec,
                                                                 hist, order_bits, p5, one_key ? 1 : 0);
        STAGE_TRY(radix_sort_pairs_u32(oka, okb, ova, ovb, E, 0, order_bits, b.tmp, b.tmp_bytes, &ob, s, order_blocks));
    } else {
        stage_order_kernel<uint64_t><<<order_blocks, 256, 0, s>>>(b.rank_rec, b.tile_sums, b.ent_first, b.ent_bseq, b.brank_of,
                                                                 b.head_pos, b.ent_key, W, E, fmax, freq_bits, b.keyA, ova, b.rec,
                                                                 hist, order_bits, p5, one_key ? 1 : 0);
        STAGE_TRY(radix_sort_pairs_u64(b.keyA, b.keyB, ova, ovb, E, 0, order_bits, b.tmp, b.tmp_bytes, &ob, s, order_blocks));
    }
    const uint32_t *perm_final = ob ? ovb : ova;
    stage_emit_kernel<W><<<grid_for(E), 256, 0, s>>>(perm_final, E, B, use_score ? 1 : 0, umi_len, b.rec, b.ent_key, b.best,
                                                     d_keys, d_nmask, d_freq, d_rep, d_bucket_off);
    STAGE_TRY(hipGetLastError());
    STAGE_TRY(hipStreamSynchronize(s));
    *n_entries_out = E;
    *n_buckets_out = B;
    return 0;
}

} // namespace

size_t stage_workspace_bytes(uint32_t n_reads, int n_words) { return carve(nullptr, n_reads, n_words).total; }

// 0 ok; 1 a character outside ATCGN; negative: -(hipError_t)
int stage_reads_on_device(void *workspace, const uint64_t *d_align, int align_bits, const uint8_t *d_umi,
                          const int32_t *d_score, uint32_t n, int umi_len, int n_words, int merge, uint64_t *d_keys,
                          uint64_t *d_nmask, int32_t *d_freq, uint64_t *d_rep, uint64_t *d_bucket_off,
                          uint64_t *n_entries_out, uint64_t *n_buckets_out, unsigned long long *h_pinned4,
                          hipStream_t s)
{
    *n_entries_out = *n_buckets_out = 0;
    if (n == 0) {
        STAGE_TRY(hipMemsetAsync(d_bucket_off, 0, 8, s));
        STAGE_TRY(hipStreamSynchronize(s));
        return 0;
    }
#define STAGE_W(WN)                                                                                               \
    return stage_impl<WN>(workspace, d_align, align_bits, d_umi, d_score, n, umi_len, merge, d_keys, d_nmask, d_freq, \
                          d_rep, d_bucket_off, n_entries_out, n_buckets_out, h_pinned4, s)
    switch (n_words) {
    case 1: STAGE_W(1);
    case 2: STAGE_W(2);
    case 3: STAGE_W(3);
    case 4: STAGE_W(4);
    default: return -(int)hipErrorInvalidValue;
    }
#undef STAGE_W
}

#undef STAGE_TRY

} // namespace umihip
