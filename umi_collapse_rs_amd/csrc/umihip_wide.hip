// Multi-word UMI keys (gfx950): umi_len 22..85, 2..4 words of 64 bits per key.
//
// What it replaces in the reference (tkob-vh/umi-collapse-rs): the same lines as the one-word
// kernels -- Naive::remove_near's scans (src/data/naive.rs:26-40) over BitSet::bit_count_xor's
// per-word loop (src/utils/bitset.rs:77-91) and umi_dist (src/utils/mod.rs:24-26) -- for keys of
// more than one word.  The arithmetic is the reference's word by word, its quirk included: a base
// that straddles two words (base 21: bits 63..65) has its N mask split 1 + 2 over them, so
// popcount(x) / 3 is 0 in both and an N mismatch there counts 3 bits, not 2 (SURVEY.md 8a A2,
// KAT G5).  No BASELINE config has such UMIs: this is the plain exact all-pairs evaluation, one
// wave per 64-row chunk, every pair's distance from all its words -- no filter keys, no n-gram
// partition, no fused kernel.  The pairs go to the same edge list and the same collapse as the
// one-word kernels'.  Integer / bitwise work, 64-lane waves, no MFMA.
#include <hip/hip_runtime.h>

#include "umihip_internal.h"
#include "umihip_device.h"

namespace umihip {

namespace {

// bitset.rs:77-91, word by word, then utils/mod.rs:25
template <int W>
__device__ __forceinline__ int wide_dist(const uint64_t (&ka)[W], const uint64_t (&na)[W], const uint64_t (&kb)[W],
                                         const uint64_t (&nb)[W])
{
    int res = 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        const uint64_t x = na[w] ^ nb[w];
        res += __builtin_popcountll(x | (ka[w] ^ kb[w])) - __builtin_popcountll(x) / 3;
    }
    return res / 2;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// One task: rows [row0, min(row0 + 64, row_end)) (one per lane, the key's words in registers)
// against columns [col0, col1), 64 at a time, one per lane as well and broadcast with v_readlane;
// only pairs with row < column count.
template <int W, bool HAS_N>
__global__ __launch_bounds__(64) void wide_pair_kernel(PairArgs a, int n_words_unused)
{
    __shared__ EdgeStage stage;
    const int lane = threadIdx.x;
    if (lane == 0) {
        stage.count = 0;
        stage.candidates = 0;
    }
    __syncthreads();
    const PairTask t = a.tasks[blockIdx.x];
    const bool with_dist = a.mode == MODE_NEIGHBOURS;
    const uint32_t r = t.row0 + (uint32_t)lane;
    const bool row_ok = r < t.row_end && r < t.row0 + 64u;
    uint64_t kr[W], nr[W];
#pragma unroll
    for (int w = 0; w < W; w++) {
        kr[w] = row_ok ? a.keys[(size_t)r * W + w] : 0ull;
        nr[w] = (HAS_N && row_ok) ? a.nmask[(size_t)r * W + w] : 0ull;
    }
    const int32_t fr = row_ok ? a.freq[r] : 0, tr = row_ok ? a.thr[r] : 0;
    unsigned int n_cand = 0;
    for (uint32_t c0 = t.col0; c0 < t.col1; c0 += 64) {
        const uint32_t c = c0 + (uint32_t)lane;
        const bool col_ok = c < t.col1;
        uint64_t kc[W], nc[W];
#pragma unroll
        for (int w = 0; w < W; w++) {
            kc[w] = col_ok ? a.keys[(size_t)c * W + w] : 0ull;
            nc[w] = (HAS_N && col_ok) ? a.nmask[(size_t)c * W + w] : 0ull;
        }
        const int32_t fc = col_ok ? a.freq[c] : 0, tc = col_ok ? a.thr[c] : 0;
        const uint32_t ncols = min(64u, t.col1 - c0);
        for (uint32_t j = 0; j < ncols; j++) { // (wave-uniform trip count)
            uint64_t kb[W], nb[W];
#pragma unroll
            for (int w = 0; w < W; w++) {
                kb[w] = readlane64(kc[w], (int)j);
                nb[w] = HAS_N ? readlane64(nc[w], (int)j) : 0ull;
            }
            const int32_t fj = __builtin_amdgcn_readlane(fc, (int)j), tj = __builtin_amdgcn_readlane(tc, (int)j);
            const uint32_t gj = c0 + j;
            if (!row_ok || r >= gj) continue;
            const int dist = wide_dist<W>(kr, nr, kb, nb);
            if (dist > a.k) continue;
            n_cand++;
            if (a.mode == MODE_NEIGHBOURS) {
                emit_edge(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, r, gj, dist, true);
                continue;
            }
            bool fwd, bwd;
            if (a.mode == MODE_DIRECTIONAL) { // naive.rs:31 with max_freq = threshold(start) (directional.rs:38-39)
                fwd = fj <= tr;
                bwd = fr <= tj;
            } else { // adjacency.rs:56: a root only ever sees entries of larger rank
                fwd = fj <= a.adj_max_freq;
                bwd = false;
            }
            if (fwd && bwd) emit_edge(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, r | SYM_FLAG, gj, dist, false);
            else if (fwd) emit_edge(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, r, gj, dist, false);
            else if (bwd) emit_edge(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, gj, r, dist, false);
        }
        flush_edges<64>(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, with_dist, false);
    }
    flush_edges<64>(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, with_dist, true);
    for (int off = 32; off > 0; off >>= 1) n_cand += __shfl_down(n_cand, off);
    if (lane == 0 && n_cand) atomicAdd(&a.counters[CNT_CANDIDATES], (unsigned long long)n_cand);
}

} // namespace

hipError_t launch_wide_pairs(const PairArgs &a, uint32_t n_tasks, int n_words, hipStream_t s)
{
    if (n_tasks == 0) return hipSuccess;
    const bool has_n = a.nmask != nullptr;
#define UMI_WIDE(Wn)                                                                        \
    do {                                                                                    \
        if (has_n) wide_pair_kernel<Wn, true><<<n_tasks, 64, 0, s>>>(a, n_words);           \
        else wide_pair_kernel<Wn, false><<<n_tasks, 64, 0, s>>>(a, n_words);                \
    } while (0)
    switch (n_words) {
    case 2: UMI_WIDE(2); break;
    case 3: UMI_WIDE(3); break;
    case 4: UMI_WIDE(4); break;
    default: return hipErrorInvalidValue;
    }
#undef UMI_WIDE
    return hipGetLastError();
}

} // namespace umihip
