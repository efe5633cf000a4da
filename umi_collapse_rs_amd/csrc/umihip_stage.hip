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
//      BASELINE config), else a sort per key word, least significant first (umihip_radix.hip);
//   2. equal neighbours are the Occupied arm: head flags, one scan (entry and position numbers ride
//      in the two halves of a 64-bit word), per entry its first read, its best read (Merge) and
//      its freq, per position its first read;
//   3. what the maps' iteration order leaves open is fixed the canonical way:
//      positions by first appearance in the file, UMIs of a position by freq descending, ties by
//      first appearance.  First appearance needs no sort: flags at the entries' / positions' first
//      reads in FILE order, one scan, and every entry knows its rank; the entries, put in that
//      order, are then radix-sorted by (position rank, max freq - freq) alone -- three digit passes
//      at 10^7 reads where three sorts on 32-bit keys took twelve.
// Keys of several words (umi_len 22..85) take the same route with W words per key.
// Integer / byte work, HBM streams; no MFMA.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>

#include "umihip_internal.h"

namespace umihip {

namespace {

inline uint32_t grid_for(uint64_t n, int block = 256, uint32_t cap = 4096)
{
    uint64_t g = (n + block - 1) / block;
    return (uint32_t)(g < 1 ? 1 : (g > cap ? cap : g));
}

// src/utils/read.rs:23-31: A 000, T 101, C 110, G 011, N 100; anything else is the reference's
// panic "Unknown character in UMI sequence" (src/utils/mod.rs:77-79)
__device__ __forceinline__ uint32_t base_code(uint8_t c)
{
    return c == 'A' ? 0u : c == 'T' ? 5u : c == 'C' ? 6u : c == 'G' ? 3u : c == 'N' ? 4u : 8u;
}

enum StageCounter : int { SC_BAD = 0, SC_ENTRIES = 1, SC_BUCKETS = 2, SC_FMAX = 3, SC_COUNT = 4 };

// k3[i * W + w] = word w of read i's UMI key (base b at bits 3b .. 3b+2 of the word string,
// utils/mod.rs:38-41, bitset.rs:52-61); idx[i] = i; composed (may be null): the sort key
// (alignment << umi bits | UMI) where that fits 64 bits
template <int W>
__global__ __launch_bounds__(256) void stage_encode_kernel(const uint8_t *__restrict__ umi, const uint64_t *__restrict__ align,
                                                           uint32_t n, int umi_len, int align_bits,
                                                           uint64_t *__restrict__ k3, uint32_t *__restrict__ idx,
                                                           uint64_t *__restrict__ composed,
                                                           unsigned long long *__restrict__ counters)
{
    unsigned int bad = 0;
    const bool words_ok = (umi_len & 3) == 0 && ((uintptr_t)umi & 3) == 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint8_t *u = umi + (size_t)i * umi_len;
        uint64_t key[W];
#pragma unroll
        for (int w = 0; w < W; w++) key[w] = 0;
        uint32_t word = 0;
        for (int b = 0; b < umi_len; b++) {
            // (four bases per load where every UMI starts on a 4-byte boundary: twelve byte loads per
            // read made this kernel 0.23 ms at 10^7 reads)
            if (words_ok) {
                if ((b & 3) == 0) word = ((const uint32_t *)u)[b >> 2];
            } else {
                word = (uint32_t)u[b] << (8 * (b & 3));
            }
            const uint32_t c = base_code((uint8_t)(word >> (8 * (b & 3))));
            bad += c > 7u ? 1u : 0u;
            const int bit = 3 * b, w = bit >> 6, sh = bit & 63;
#pragma unroll
            for (int q = 0; q < W; q++) {
                if (q == w) key[q] |= (uint64_t)(c & 7u) << sh;
                if (q == w + 1 && sh > 61) key[q] |= (uint64_t)(c & 7u) >> (64 - sh); // a base across two words
            }
        }
        idx[i] = i;
        if (composed) { // (the UMI key is the low part of it: no array of its own)
            const uint64_t a = align_bits >= 64 ? align[i] : align[i] & ((1ull << align_bits) - 1ull);
            composed[i] = (a << (3 * umi_len)) | key[0];
        } else {
#pragma unroll
            for (int w = 0; w < W; w++) k3[(size_t)i * W + w] = key[w];
        }
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

// reads in (alignment key, UMI, file index) order: where a new UMI entry / a new position begins,
// as the low / high half of a 64-bit flag word.  composed != null: the sorted composed keys say it
// all; else the reads' keys are gathered through the permutation.
template <int W>
__global__ __launch_bounds__(256) void stage_heads_kernel(const uint64_t *__restrict__ composed, int umi_bits,
                                                          const uint64_t *__restrict__ align, int align_bits,
                                                          const uint64_t *__restrict__ k3,
                                                          const uint32_t *__restrict__ perm, uint32_t n,
                                                          uint64_t *__restrict__ flags)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        bool h = true, bh = true;
        if (i > 0) {
            if (composed) {
                const uint64_t a = composed[i], b = composed[i - 1];
                h = a != b;
                bh = umi_bits >= 64 ? false : (a >> umi_bits) != (b >> umi_bits);
            } else {
                const uint32_t r = perm[i], q = perm[i - 1];
                const uint64_t am = align_bits >= 64 ? ~0ull : (1ull << align_bits) - 1ull;
                bh = (align[r] & am) != (align[q] & am);
                h = bh;
#pragma unroll
                for (int w = 0; w < W; w++) h = h || k3[(size_t)r * W + w] != k3[(size_t)q * W + w]; // BitSet equality is
                                                                                                     // on the bits alone (bitset.rs:94-101)
            }
        }
        flags[i] = (h ? 1ull : 0ull) | (bh ? 1ull << 32 : 0ull);
    }
}

// per head read: its entry's place in the sorted order, first read (the sort is stable: the head is
// the entry's first read in the file), position number and key; the counts for the host
template <int W>
__global__ __launch_bounds__(256) void stage_entry_heads_kernel(const uint64_t *__restrict__ flags,
                                                                const uint64_t *__restrict__ numbers,
                                                                const uint32_t *__restrict__ perm,
                                                                const uint64_t *__restrict__ k3,
                                                                const uint64_t *__restrict__ composed, int umi_bits,
                                                                uint32_t n,
                                                                uint32_t *__restrict__ head_pos,
                                                                uint32_t *__restrict__ ent_first,
                                                                uint32_t *__restrict__ ent_bseq,
                                                                uint64_t *__restrict__ ent_key,
                                                                unsigned long long *__restrict__ counters)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t num = numbers[i];
        if (flags[i] & 1ull) {
            const uint32_t e = (uint32_t)num - 1u, r = perm[i];
            head_pos[e] = i;
            ent_first[e] = r;
            ent_bseq[e] = (uint32_t)(num >> 32) - 1u;
            if (composed) { // (W == 1: the UMI key is the low part of the sorted composed key)
                ent_key[e] = composed[i] & ((1ull << umi_bits) - 1ull);
            } else {
#pragma unroll
                for (int w = 0; w < W; w++) ent_key[(size_t)e * W + w] = k3[(size_t)r * W + w];
            }
        }
        if (i == n - 1) {
            head_pos[(uint32_t)num] = n;
            counters[SC_ENTRIES] = (uint32_t)num;
            counters[SC_BUCKETS] = (uint32_t)(num >> 32);
        }
    }
}

// the largest freq of any entry (the width of the freq field of the final sort key)
__global__ __launch_bounds__(256) void stage_fmax_kernel(const uint32_t *__restrict__ head_pos, uint32_t n,
                                                         unsigned long long *__restrict__ counters)
{
    const uint32_t n_entries = (uint32_t)counters[SC_ENTRIES];
    uint32_t m = 0;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x)
        m = max(m, head_pos[e + 1] - head_pos[e]);
    // one atomic per block, and a look first: one word takes ~90 accesses per microsecond
    __shared__ uint32_t wmax[4];
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (m && __hip_atomic_load(&counters[SC_FMAX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)m)
            atomicMax(&counters[SC_FMAX], (unsigned long long)m);
    }
}

// The maximum (or minimum) of v over the lanes of a wave that share a run of equal ids, at the
// last lane of each run (ids are non-decreasing across the lanes): a segmented scan by doubling.
template <bool MAX>
__device__ __forceinline__ unsigned long long run_extreme(unsigned long long v, uint32_t id, bool valid, bool *is_last)
{
    const int lane = threadIdx.x & 63;
    const uint32_t my = valid ? id : 0xFFFFFFFFu;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long ov = __shfl_up(v, off);
        const uint32_t oid = (uint32_t)__shfl_up((int)my, off);
        if (lane >= off && oid == my) v = MAX ? (ov > v ? ov : v) : (ov < v ? ov : v);
    }
    const uint32_t nid = (uint32_t)__shfl_down((int)my, 1);
    *is_last = valid && (lane == 63 || nid != my);
    return v;
}

// per read in sorted order: the entry's best read (Merge) and the position's first read, one
// atomic per run of a wave
__global__ __launch_bounds__(256) void stage_reads_kernel(const uint64_t *__restrict__ numbers,
                                                          const uint32_t *__restrict__ perm,
                                                          const int32_t *__restrict__ score, uint32_t n, int merge,
                                                          unsigned long long *__restrict__ best,
                                                          uint32_t *__restrict__ bfirst)
{
    const uint32_t n_round = (n + 63u) & ~63u; // (whole waves: the shuffles need every lane)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        const bool valid = i < n;
        const uint64_t num = valid ? numbers[i] : 0ull;
        const uint32_t s = (uint32_t)num - 1u, b = (uint32_t)(num >> 32) - 1u;
        const uint32_t r = valid ? perm[i] : 0u;
        bool last;
        if (merge && score) { // merge/mod.rs:35,49: the higher score, the earlier read on a tie
            const unsigned long long packed =
                ((unsigned long long)((uint32_t)score[r] ^ 0x80000000u) << 32) | (unsigned long long)(0xFFFFFFFFu - r);
            const unsigned long long m = run_extreme<true>(valid ? packed : 0ull, s, valid, &last);
            if (last && __hip_atomic_load(&best[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < m) atomicMax(&best[s], m);
        }
        // (a look before the atomic: a deep position is thousands of waves' runs on one word, which takes
        // ~90 atomics per microsecond -- and all but a few of them have nothing to lower)
        const unsigned long long mn = run_extreme<false>(valid ? (unsigned long long)r : ~0ull, b, valid, &last);
        if (last && __hip_atomic_load(&bfirst[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (uint32_t)mn)
            atomicMin(&bfirst[b], (uint32_t)mn);
    }
}

// flags in FILE order: low half at the first read of every entry, high half at the first read of
// every position (the halves are written as separate 32-bit words: a read can be both)
__global__ __launch_bounds__(256) void stage_mark_kernel(const uint32_t *__restrict__ ent_first, uint32_t n_entries,
                                                         const uint32_t *__restrict__ bfirst, uint32_t n_buckets,
                                                         uint32_t *__restrict__ file_flags)
{
    const uint32_t total = n_entries + n_buckets;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        if (j < n_entries) file_flags[2 * (size_t)ent_first[j]] = 1u;
        else file_flags[2 * (size_t)bfirst[j - n_entries] + 1] = 1u;
    }
}

// the entries in order of first appearance, each with the key of the one sort that is left:
// (position's rank of first appearance, max freq - freq)
__global__ __launch_bounds__(256) void stage_order_kernel(const uint64_t *__restrict__ file_numbers,
                                                          const uint32_t *__restrict__ ent_first,
                                                          const uint32_t *__restrict__ ent_bseq,
                                                          const uint32_t *__restrict__ bfirst,
                                                          const uint32_t *__restrict__ head_pos, uint32_t n_entries,
                                                          uint32_t fmax, int freq_bits, uint64_t *__restrict__ okey,
                                                          uint32_t *__restrict__ oval, uint32_t *__restrict__ brank)
{
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += gridDim.x * blockDim.x) {
        const uint32_t at = (uint32_t)file_numbers[ent_first[e]] - 1u;            // rank of the entry's first read
        const uint32_t b = ent_bseq[e];
        const uint32_t br = (uint32_t)(file_numbers[bfirst[b]] >> 32) - 1u;       // rank of the position's first read
        const uint32_t f = head_pos[e + 1] - head_pos[e];
        okey[at] = ((uint64_t)br << freq_bits) | (uint64_t)(fmax - f);
        oval[at] = e;
        brank[e] = br;
    }
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
                                                         const uint64_t *__restrict__ ent_key,
                                                         const uint32_t *__restrict__ ent_first,
                                                         const uint32_t *__restrict__ head_pos,
                                                         const uint32_t *__restrict__ brank,
                                                         const unsigned long long *__restrict__ best,
                                                         uint64_t *__restrict__ keys, uint64_t *__restrict__ nmask,
                                                         int32_t *__restrict__ freq, uint64_t *__restrict__ rep,
                                                         uint64_t *__restrict__ bucket_off)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_entries; j += gridDim.x * blockDim.x) {
        const uint32_t e = perm[j];
        uint64_t key[W];
#pragma unroll
        for (int w = 0; w < W; w++) keys[(size_t)j * W + w] = key[w] = ent_key[(size_t)e * W + w];
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
        freq[j] = (int32_t)(head_pos[e + 1] - head_pos[e]);
        rep[j] = merge ? (uint64_t)(0xFFFFFFFFu - (uint32_t)best[e]) : (uint64_t)ent_first[e];
        const uint32_t br = brank[e];
        if (j == 0 || brank[perm[j - 1]] != br) bucket_off[br] = j;
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
    uint64_t *k3, *keyA, *keyB, *ent_key, *flags, *numbers;
    unsigned long long *best;
    uint32_t *idxA, *idxB, *ent_first, *ent_bseq, *head_pos, *bfirst, *brank;
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
    b.flags = c.take<uint64_t>(m);
    b.numbers = c.take<uint64_t>(m);
    b.best = c.take<unsigned long long>(m);
    b.idxA = c.take<uint32_t>(m);
    b.idxB = c.take<uint32_t>(m);
    b.ent_first = c.take<uint32_t>(m);
    b.ent_bseq = c.take<uint32_t>(m);
    b.head_pos = c.take<uint32_t>(m);
    b.bfirst = c.take<uint32_t>(m);
    b.brank = c.take<uint32_t>(m);
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
    const bool one_key = W == 1 && align_bits + umi_bits <= 64; // the composed sort key fits a word
    STAGE_TRY(hipMemsetAsync(b.counters, 0, SC_COUNT * 8, s));
    stage_encode_kernel<W><<<grid_for(n), 256, 0, s>>>(d_umi, d_align, n, umi_len, align_bits, b.k3, b.idxA,
                                                       one_key ? b.keyA : nullptr, b.counters);
    // ---- 1. reads by (alignment key, UMI, file index)
    uint64_t *ka = b.keyA, *kb = b.keyB;
    uint32_t *va = b.idxA, *vb = b.idxB;
    bool in_b = false;
    auto sort_by = [&](int begin_bit, int end_bit) -> hipError_t {
        const hipError_t e = radix_sort_pairs_u64(ka, kb, va, vb, n, begin_bit, end_bit, b.tmp, b.tmp_bytes, &in_b, s);
        if (e == hipSuccess && in_b) {
            std::swap(ka, kb);
            std::swap(va, vb);
        }
        return e;
    };
    if (one_key) {
        STAGE_TRY(sort_by(0, align_bits + umi_bits));
    } else { // a stable sort per key word, least significant first: the UMI's words, then the alignment key
        for (int w = 0; w < W; w++) {
            stage_gather_u64_kernel<<<grid_for(n), 256, 0, s>>>(b.k3, W, w, va, n, ka);
            STAGE_TRY(sort_by(0, std::min(64, umi_bits - 64 * w)));
        }
        stage_gather_u64_kernel<<<grid_for(n), 256, 0, s>>>(d_align, 1, 0, va, n, ka);
        STAGE_TRY(sort_by(0, align_bits));
    }
    // (va: the reads' file indices in order; ka: the composed keys in order where there is one)
    // ---- 2. entries and positions
    stage_heads_kernel<W><<<grid_for(n), 256, 0, s>>>(one_key ? ka : nullptr, umi_bits, d_align, align_bits, b.k3, va, n,
                                                      b.flags);
    STAGE_TRY(scan_inclusive_u64(b.flags, b.numbers, n, b.tmp, b.tmp_bytes, s));
    stage_entry_heads_kernel<W><<<grid_for(n), 256, 0, s>>>(b.flags, b.numbers, va, b.k3, one_key ? ka : nullptr, umi_bits, n,
                                                           b.head_pos, b.ent_first, b.ent_bseq, b.ent_key, b.counters);
    stage_fmax_kernel<<<grid_for(n, 256 * 16, 512), 256, 0, s>>>(b.head_pos, n, b.counters);
    // the host needs the counts to size what follows (and the verdict on the characters)
    STAGE_TRY(hipMemcpyAsync(h_pinned4, b.counters, SC_COUNT * 8, hipMemcpyDeviceToHost, s));
    STAGE_TRY(hipStreamSynchronize(s));
    if (h_pinned4[SC_BAD]) return 1;
    const uint32_t E = (uint32_t)h_pinned4[SC_ENTRIES], B = (uint32_t)h_pinned4[SC_BUCKETS];
    const uint32_t fmax = (uint32_t)h_pinned4[SC_FMAX];
    if (use_score) STAGE_TRY(hipMemsetAsync(b.best, 0, (size_t)E * 8, s));
    STAGE_TRY(hipMemsetAsync(b.bfirst, 0xFF, (size_t)B * 4, s));
    stage_reads_kernel<<<grid_for(n), 256, 0, s>>>(b.numbers, va, d_score, n, use_score ? 1 : 0, b.best, b.bfirst);
    // ---- 3. the canonical order: first appearance by flags in file order and one scan ...
    STAGE_TRY(hipMemsetAsync(b.flags, 0, (size_t)n * 8, s));
    stage_mark_kernel<<<grid_for((uint64_t)E + B), 256, 0, s>>>(b.ent_first, E, b.bfirst, B, (uint32_t *)b.flags);
    STAGE_TRY(scan_inclusive_u64(b.flags, b.numbers, n, b.tmp, b.tmp_bytes, s));
    // ... and one stable sort of the entries, taken in that order, by (position rank, max freq - freq)
    const int freq_bits = bits_for(fmax), rank_bits = bits_for(B ? B - 1 : 0);
    uint64_t *oka = b.keyA, *okb = b.keyB; // (the read sort's buffers are free: its order lives on in va)
    uint32_t *ova = va == b.idxA ? b.idxB : b.idxA, *ovb = b.ent_bseq; // ent_bseq is read by the order kernel: not yet
    uint32_t *oscratch = nullptr;
    (void)oscratch;
    stage_order_kernel<<<grid_for(E), 256, 0, s>>>(b.numbers, b.ent_first, b.ent_bseq, b.bfirst, b.head_pos, E, fmax,
                                                   freq_bits, oka, ova, b.brank);
    ovb = b.ent_bseq; // (free now)
    bool ob = false;
    STAGE_TRY(radix_sort_pairs_u64(oka, okb, ova, ovb, E, 0, std::min(64, freq_bits + rank_bits), b.tmp, b.tmp_bytes, &ob, s));
    const uint32_t *perm_final = ob ? ovb : ova;
    stage_emit_kernel<W><<<grid_for(E), 256, 0, s>>>(perm_final, E, B, use_score ? 1 : 0, umi_len, b.ent_key, b.ent_first,
                                                     b.head_pos, b.brank, b.best, d_keys, d_nmask, d_freq, d_rep,
                                                     d_bucket_off);
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
