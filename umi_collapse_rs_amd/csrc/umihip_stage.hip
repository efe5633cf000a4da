// Read staging on the device (gfx950): from the reads of a file to the batched hot path's input.
//
// What it replaces in the reference (tkob-vh/umi-collapse-rs): the per-read part of
// DeduplicateSAM::deduplicate_and_merge (src/deduplicate_sam.rs:148-176) -- to_bitset
// (src/utils/mod.rs:63-83, codes src/utils/read.rs:23-31), the per-position map UMI -> ReadFreq
// with its Vacant / Occupied arms (:161-175: freq += 1, the kept read chosen by Merge,
// src/merge/mod.rs:18-51) -- and the order Directional / Adjacency::apply put a position's UMIs in
// (stable by freq descending, src/algo/directional.rs:67-72).  The reference walks the reads one by
// one through two HashMaps; here the reads are sorted by (alignment key, UMI) with their file
// index as the tie-break (two stable radix sorts: rocPRIM, a library primitive as in
// umihip_sort.hip), equal neighbours are the Occupied arm, and what the maps' iteration order
// leaves open is fixed the canonical way: positions by first appearance in
// the file, UMIs of a position by freq descending, ties by first appearance.
//
// Integer / byte work, HBM streams; no MFMA.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "umihip_internal.h"

namespace umihip {

namespace {

constexpr uint32_t SORT_MERGE_LIMIT = 65536;
using sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                               rocprim::default_config, SORT_MERGE_LIMIT>;

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

__global__ __launch_bounds__(256) void stage_encode_kernel(const uint8_t *__restrict__ umi, uint32_t n, int umi_len,
                                                           uint64_t *__restrict__ k3, uint32_t *__restrict__ idx,
                                                           unsigned long long *__restrict__ counters)
{
    unsigned int bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint8_t *u = umi + (size_t)i * umi_len;
        uint64_t key = 0;
        for (int b = 0; b < umi_len; b++) { // base b at bits 3b .. 3b+2 (utils/mod.rs:38-41)
            const uint32_t c = base_code(u[b]);
            bad += c > 7u ? 1u : 0u;
            key |= (uint64_t)(c & 7u) << (3 * b);
        }
        k3[i] = key;
        idx[i] = i;
    }
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicAdd(&counters[0], 1ull);
}

__global__ __launch_bounds__(256) void stage_gather_u64_kernel(const uint64_t *__restrict__ src,
                                                               const uint32_t *__restrict__ pos, uint32_t n,
                                                               uint64_t *__restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = src[pos[i]];
}

// reads in (alignment key, UMI, file index) order: where a new UMI entry / a new position begins;
// the UMI keys gathered into that order
__global__ __launch_bounds__(256) void stage_heads_kernel(const uint64_t *__restrict__ align_sorted,
                                                          const uint64_t *__restrict__ k3,
                                                          const uint32_t *__restrict__ perm, uint32_t n,
                                                          uint64_t *__restrict__ key_sorted,
                                                          uint32_t *__restrict__ head, uint32_t *__restrict__ bhead)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t a = align_sorted[i], kk = k3[perm[i]];
        key_sorted[i] = kk;
        bool h = true, bh = true;
        if (i > 0) {
            bh = a != align_sorted[i - 1];
            h = bh || kk != k3[perm[i - 1]]; // BitSet equality is on the bits alone (bitset.rs:94-101)
        }
        head[i] = h ? 1u : 0u;
        bhead[i] = bh ? 1u : 0u;
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

// per read in sorted order: its entry's head data; the entry's best read (Merge) and the
// position's first read by one atomic per run of a wave
__global__ __launch_bounds__(256) void stage_entries_kernel(const uint64_t *__restrict__ key_sorted,
                                                            const uint32_t *__restrict__ perm,
                                                            const uint32_t *__restrict__ head,
                                                            const uint32_t *__restrict__ segid,
                                                            const uint32_t *__restrict__ bseq,
                                                            const int32_t *__restrict__ score, uint32_t n,
                                                            int merge, uint32_t n_entries,
                                                            uint64_t *__restrict__ ent_key,
                                                            uint32_t *__restrict__ ent_first,
                                                            uint32_t *__restrict__ ent_bseq,
                                                            uint32_t *__restrict__ head_pos,
                                                            unsigned long long *__restrict__ best,
                                                            uint32_t *__restrict__ bfirst)
{
    const uint32_t n_round = (n + 63u) & ~63u; // (whole waves: the shuffles need every lane)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        const bool valid = i < n;
        const uint32_t s = valid ? segid[i] - 1u : 0u, b = valid ? bseq[i] - 1u : 0u;
        const uint32_t r = valid ? perm[i] : 0u;
        if (valid && head[i]) { // (the sort is stable: the head is the entry's first read in the file)
            head_pos[s] = i;
            ent_key[s] = key_sorted[i];
            ent_first[s] = r;
            ent_bseq[s] = b;
        }
        bool last;
        if (merge && score) { // merge/mod.rs:35,49: the higher score, the earlier read on a tie
            const unsigned long long packed =
                ((unsigned long long)((uint32_t)score[r] ^ 0x80000000u) << 32) | (unsigned long long)(0xFFFFFFFFu - r);
            const unsigned long long m = run_extreme<true>(valid ? packed : 0ull, s, valid, &last);
            if (last) atomicMax(&best[s], m);
        }
        const unsigned long long mn = run_extreme<false>(valid ? (unsigned long long)r : ~0ull, b, valid, &last);
        if (last) atomicMin(&bfirst[b], (uint32_t)mn);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) head_pos[n_entries] = n;
}

// sort key of the entries for one of the three stable passes of their final order:
// 0 first appearance of the UMI, 1 freq descending, 2 first appearance of the position
__global__ __launch_bounds__(256) void stage_order_key_kernel(int pass, const uint32_t *__restrict__ perm_in,
                                                              uint32_t n_entries, uint32_t n_reads,
                                                              const uint32_t *__restrict__ ent_first,
                                                              const uint32_t *__restrict__ head_pos,
                                                              const uint32_t *__restrict__ ent_bseq,
                                                              const uint32_t *__restrict__ bfirst,
                                                              uint32_t *__restrict__ key_out,
                                                              uint32_t *__restrict__ perm_iota)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_entries; j += gridDim.x * blockDim.x) {
        const uint32_t e = perm_in ? perm_in[j] : j;
        if (pass == 0) {
            key_out[j] = ent_first[e];
            perm_iota[j] = j;
        } else if (pass == 1) {
            key_out[j] = n_reads - (head_pos[e + 1] - head_pos[e]); // freq descending
        } else {
            key_out[j] = bfirst[ent_bseq[e]];
        }
    }
}

// bit 2 set and bits 0, 1 clear: the N code; its three bits go to the mask (utils/mod.rs:45-50,74-76)
__device__ __forceinline__ uint64_t nmask_of(uint64_t key)
{
    const uint64_t b2 = key & 0x4924924924924924ull;
    const uint64_t n = b2 & ~((key << 1) | (key << 2)); // bit 2 of the N bases
    return n | (n >> 1) | (n >> 2);
}

// the entries in their final order; where a new position begins
__global__ __launch_bounds__(256) void stage_emit_kernel(const uint32_t *__restrict__ perm, uint32_t n_entries,
                                                         int merge, const uint64_t *__restrict__ ent_key,
                                                         const uint32_t *__restrict__ ent_first,
                                                         const uint32_t *__restrict__ head_pos,
                                                         const uint32_t *__restrict__ ent_bseq,
                                                         const unsigned long long *__restrict__ best,
                                                         uint64_t *__restrict__ keys, uint64_t *__restrict__ nmask,
                                                         int32_t *__restrict__ freq, uint64_t *__restrict__ rep,
                                                         uint32_t *__restrict__ bhead)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_entries; j += gridDim.x * blockDim.x) {
        const uint32_t e = perm[j];
        const uint64_t key = ent_key[e];
        keys[j] = key;
        if (nmask) nmask[j] = nmask_of(key);
        freq[j] = (int32_t)(head_pos[e + 1] - head_pos[e]);
        rep[j] = merge ? (uint64_t)(0xFFFFFFFFu - (uint32_t)best[e]) : (uint64_t)ent_first[e];
        bhead[j] = (j == 0 || ent_bseq[perm[j - 1]] != ent_bseq[e]) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void stage_offsets_kernel(const uint32_t *__restrict__ bhead,
                                                            const uint32_t *__restrict__ bno, uint32_t n_entries,
                                                            uint32_t n_buckets, uint64_t *__restrict__ bucket_off)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_entries; j += gridDim.x * blockDim.x)
        if (bhead[j]) bucket_off[bno[j] - 1u] = j;
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
    uint64_t *k3, *keyA, *keyB, *ent_key;
    unsigned long long *best;
    uint32_t *idxA, *idxB, *head, *bhead, *segid, *bseq, *ent_first, *ent_bseq, *head_pos, *bfirst, *k32a, *k32b;
    unsigned long long *counters; // [0] bad characters
    void *tmp;
    size_t tmp_bytes, total;
};

size_t prim_temp_bytes(uint32_t n)
{
    size_t a = 0, b = 0, c = 0;
    (void)rocprim::radix_sort_pairs<sort_config>(nullptr, a, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                                 (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 64);
    (void)rocprim::radix_sort_pairs<sort_config>(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                                 (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 32);
    (void)rocprim::inclusive_scan(nullptr, c, (const uint32_t *)nullptr, (uint32_t *)nullptr, n,
                                  rocprim::plus<uint32_t>());
    return std::max(a, std::max(b, c)) + 256;
}

StageBufs carve(void *ws, uint32_t n)
{
    Carver c{(char *)ws};
    StageBufs b;
    const size_t m = (size_t)n + 1;
    b.k3 = c.take<uint64_t>(m);
    b.keyA = c.take<uint64_t>(m);
    b.keyB = c.take<uint64_t>(m);
    b.ent_key = c.take<uint64_t>(m);
    b.best = c.take<unsigned long long>(m);
    b.idxA = c.take<uint32_t>(m);
    b.idxB = c.take<uint32_t>(m);
    b.head = c.take<uint32_t>(m);
    b.bhead = c.take<uint32_t>(m);
    b.segid = c.take<uint32_t>(m);
    b.bseq = c.take<uint32_t>(m);
    b.ent_first = c.take<uint32_t>(m);
    b.ent_bseq = c.take<uint32_t>(m);
    b.head_pos = c.take<uint32_t>(m + 1);
    b.bfirst = c.take<uint32_t>(m);
    b.k32a = c.take<uint32_t>(m);
    b.k32b = c.take<uint32_t>(m);
    b.counters = c.take<unsigned long long>(4);
    b.tmp_bytes = prim_temp_bytes(n);
    b.tmp = c.take<char>(b.tmp_bytes);
    b.total = c.off;
    return b;
}

} // namespace

size_t stage_workspace_bytes(uint32_t n_reads) { return carve(nullptr, n_reads).total; }

// 0 ok; 1 a character outside ATCGN; negative: -(hipError_t)
int stage_reads_on_device(void *workspace, const uint64_t *d_align, int align_bits, const uint8_t *d_umi,
                          const int32_t *d_score, uint32_t n, int umi_len, int merge, uint64_t *d_keys,
                          uint64_t *d_nmask, int32_t *d_freq, uint64_t *d_rep, uint64_t *d_bucket_off,
                          uint64_t *n_entries_out, uint64_t *n_buckets_out, unsigned long long *h_pinned4,
                          hipStream_t s)
{
#define STAGE_TRY(x)                                   \
    do {                                               \
        const hipError_t e_ = (x);                     \
        if (e_ != hipSuccess) return -(int)e_;         \
    } while (0)
    *n_entries_out = *n_buckets_out = 0;
    if (n == 0) {
        STAGE_TRY(hipMemsetAsync(d_bucket_off, 0, 8, s));
        STAGE_TRY(hipStreamSynchronize(s));
        return 0;
    }
    StageBufs b = carve(workspace, n);
    const bool use_score = merge != 0 && d_score != nullptr;
    STAGE_TRY(hipMemsetAsync(b.counters, 0, 32, s));
    stage_encode_kernel<<<grid_for(n), 256, 0, s>>>(d_umi, n, umi_len, b.k3, b.idxA, b.counters);
    // reads by (alignment key, UMI, file index): two stable passes, the minor key first
    STAGE_TRY(rocprim::radix_sort_pairs<sort_config>(b.tmp, b.tmp_bytes, b.k3, b.keyB, b.idxA, b.idxB, n, 0,
                                                     std::min(64, 3 * umi_len), s));
    stage_gather_u64_kernel<<<grid_for(n), 256, 0, s>>>(d_align, b.idxB, n, b.keyA);
    STAGE_TRY(rocprim::radix_sort_pairs<sort_config>(b.tmp, b.tmp_bytes, b.keyA, b.keyB, b.idxB, b.idxA, n, 0,
                                                     align_bits, s));
    // keyB: alignment keys in order; idxA: the reads' file indices in order
    stage_heads_kernel<<<grid_for(n), 256, 0, s>>>(b.keyB, b.k3, b.idxA, n, b.keyA, b.head, b.bhead);
    STAGE_TRY(rocprim::inclusive_scan(b.tmp, b.tmp_bytes, b.head, b.segid, n, rocprim::plus<uint32_t>(), s));
    STAGE_TRY(rocprim::inclusive_scan(b.tmp, b.tmp_bytes, b.bhead, b.bseq, n, rocprim::plus<uint32_t>(), s));
    // the host needs the two counts to size what follows (and the verdict on the characters)
    h_pinned4[0] = h_pinned4[1] = 0; // (4-byte counts into 8-byte slots)
    STAGE_TRY(hipMemcpyAsync(&h_pinned4[0], b.segid + (n - 1), 4, hipMemcpyDeviceToHost, s));
    STAGE_TRY(hipMemcpyAsync(&h_pinned4[1], b.bseq + (n - 1), 4, hipMemcpyDeviceToHost, s));
    STAGE_TRY(hipMemcpyAsync(&h_pinned4[2], b.counters, 8, hipMemcpyDeviceToHost, s));
    STAGE_TRY(hipStreamSynchronize(s));
    if (h_pinned4[2]) return 1;
    const uint32_t E = (uint32_t)h_pinned4[0], B = (uint32_t)h_pinned4[1];
    if (use_score) STAGE_TRY(hipMemsetAsync(b.best, 0, (size_t)E * 8, s));
    STAGE_TRY(hipMemsetAsync(b.bfirst, 0xFF, (size_t)B * 4, s));
    stage_entries_kernel<<<grid_for(n), 256, 0, s>>>(b.keyA, b.idxA, b.head, b.segid, b.bseq, d_score, n,
                                                     use_score ? 1 : 0, E, b.ent_key, b.ent_first, b.ent_bseq,
                                                     b.head_pos, b.best, b.bfirst);
    // entries by (position's first read, freq descending, UMI's first read): three stable passes,
    // the minor key first; all three keys are below n + 1
    const int nbits = bits_for(n);
    uint32_t *perm_in = nullptr, *pa = b.idxB, *pb = b.segid; // (segid, bseq, head are free again)
    for (int pass = 0; pass < 3; pass++) {
        stage_order_key_kernel<<<grid_for(E), 256, 0, s>>>(pass, perm_in, E, n, b.ent_first, b.head_pos, b.ent_bseq,
                                                           b.bfirst, b.k32a, pass == 0 ? b.bseq : nullptr);
        const uint32_t *vin = pass == 0 ? b.bseq : perm_in;
        uint32_t *vout = pass == 0 ? pa : (perm_in == pa ? pb : pa);
        STAGE_TRY(rocprim::radix_sort_pairs<sort_config>(b.tmp, b.tmp_bytes, b.k32a, b.k32b, vin, vout, E, 0, nbits, s));
        perm_in = vout;
    }
    stage_emit_kernel<<<grid_for(E), 256, 0, s>>>(perm_in, E, use_score ? 1 : 0, b.ent_key, b.ent_first, b.head_pos,
                                                  b.ent_bseq, b.best, d_keys, d_nmask, d_freq, d_rep, b.head);
    STAGE_TRY(rocprim::inclusive_scan(b.tmp, b.tmp_bytes, b.head, b.bhead, E, rocprim::plus<uint32_t>(), s));
    stage_offsets_kernel<<<grid_for(E), 256, 0, s>>>(b.head, b.bhead, E, B, d_bucket_off);
    STAGE_TRY(hipGetLastError());
    STAGE_TRY(hipStreamSynchronize(s));
    *n_entries_out = E;
    *n_buckets_out = B;
    return 0;
#undef STAGE_TRY
}

} // namespace umihip
