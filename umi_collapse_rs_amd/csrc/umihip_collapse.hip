// Directional collapse, phase 1 by union-find (gfx950).
//
// What it replaces in the reference (tkob-vh/umi-collapse-rs): Directional::visit_and_remove and
// the root loop of Directional::apply (src/algo/directional.rs:30-54,78-88), as in
// umihip_kernels.hip: label[v] = smallest rank that reaches v.  Reachability inside a set of
// entries joined by symmetric pairs (both directions permitted) is symmetric, so those sets are
// plain connected components; this file finds them with a lock-free union-find over the edge list
// (one pass, any number of hops) where the hook/jump rounds of cc_hook_kernel need one launch per
// halving of the longest chain.  The one-way pairs (a DAG over the sets) are then propagated by
// dag_hook_kernel as before.
#include <hip/hip_runtime.h>

#include "umihip_internal.h"
#include "umihip_device.h"

namespace umihip {

namespace {

__global__ __launch_bounds__(256) void uf_union_kernel(const uint2 *__restrict__ edges,
                                                       const unsigned long long *counters,
                                                       uint32_t edge_cap, uint32_t *parent)
{
    const unsigned long long ne = counters[CNT_EDGES];
    const uint32_t E = ne < edge_cap ? (uint32_t)ne : edge_cap;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const uint2 uv = edges[e];
        if (uv.x & SYM_FLAG) uf_union(parent, uv.x & ~SYM_FLAG, uv.y);
    }
}

// root of v in the forest (read-only: the unions are over)
__device__ __forceinline__ uint32_t uf_root(const uint32_t *__restrict__ parent, uint32_t v)
{
    uint32_t r = parent[v];
    if (r != v)
        for (;;) {
            const uint32_t p = parent[r];
            if (p == r) break;
            r = p;
        }
    return r;
}

// One round along the one-way pairs over the flattened sets: lab[set of v] = min(.., lab[set of u])
// for every listed pair u -> v (directional.rs:38-39: v falls to whatever removes u).  No pointer
// jump over lab[] behind it: a chain of one-way pairs is as deep as the freq ladder it descends,
// two or three steps, and a pass over all of lab[] per round costs more than the round it might
// save.  (Climbing the forest as the unions left it instead of flattening it first was measured
// and dropped: its trees are deep enough that the climbs of this kernel and of the final pass cost
// more than the flatten pass, 36 + 44 us against 24 + 5 + 11.)  Only as many blocks work as the list needs (the grid is sized for the list's
// capacity, the count is on the device): the fewer waves, the more of a hot word's hooks meet in
// one wave's registers.
__global__ __launch_bounds__(256) void dag_flat_hook_kernel(const uint2 *__restrict__ edges,
                                                            const unsigned long long *counters,
                                                            uint32_t edge_cap,
                                                            const uint32_t *__restrict__ comp,
                                                            uint32_t *lab, uint32_t *changed, int round)
{
    if (round > 0 && changed[round - 1] == 0) return;
    const unsigned long long ne = counters[CNT_EDGES];
    const uint32_t E = ne < edge_cap ? (uint32_t)ne : edge_cap;
    constexpr uint32_t PER_BLOCK = 256 * 8;
    const uint32_t active = min(gridDim.x, (E + PER_BLOCK - 1) / PER_BLOCK);
    if (blockIdx.x >= active) return;
    bool any = false;
    HotMin hot;
    // (every lane of a wave makes the same number of trips: the shuffles below need them all)
    for (uint32_t e0 = blockIdx.x * blockDim.x; e0 < E; e0 += active * blockDim.x) {
        const uint32_t e = e0 + threadIdx.x;
        const uint2 uv = e < E ? edges[e] : make_uint2(SYM_FLAG, 0u);
        bool todo = false;
        uint32_t cv = 0, lu = 0;
        if (!(uv.x & SYM_FLAG)) {
            const uint32_t cu = comp[uv.x];
            cv = comp[uv.y];
            lu = lab[cu];
            todo = lu < lab[cv];
        }
        any |= todo;
        wave_atomic_min(lab, cv, lu, todo, hot);
    }
    hot_flush(lab, hot);
    if (any) changed[round] = 1;
}

// comp[v] = root of v (the smallest index of its set); lab[v] = v (lab may be null)
__global__ __launch_bounds__(256) void uf_flatten_kernel(uint32_t *parent, uint32_t *lab, uint32_t n)
{
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        uint32_t r = ld_parent(&parent[v]);
        if (r != v) {
            for (;;) {
                const uint32_t p = ld_parent(&parent[r]);
                if (p == r) break;
                r = p;
            }
            st_parent(&parent[v], r);
        }
        if (lab) lab[v] = v;
    }
}

// kept / root / survivor count of the entries of ranges (null: all n) from comp and lab:
// label[v] = lab[comp[v]] (directional.rs:30-54,78-88), kept <=> label == v
// (deduplicate_sam.rs:217-231)
template <bool FIND>
__global__ __launch_bounds__(256) void map_finalize_kernel(const uint32_t *__restrict__ comp,
                                                           const uint32_t *__restrict__ lab,
                                                           const RangeTask *__restrict__ ranges, uint32_t n,
                                                           uint8_t *__restrict__ kept,
                                                           uint32_t *__restrict__ root,
                                                           unsigned long long *counters)
{
    unsigned int cnt = 0;
    auto f = [&](uint32_t i) {
        const uint32_t l = lab[FIND ? uf_root(comp, i) : comp[i]];
        const bool kp = l == i;
        kept[i] = kp ? 1 : 0;
        if (root) root[i] = l;
        cnt += kp ? 1u : 0u;
    };
    if (ranges) {
        const RangeTask r = ranges[blockIdx.x];
        for (uint32_t i = r.start + threadIdx.x; i < r.end; i += blockDim.x) f(i);
    } else {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) f(i);
    }
    block_count_add(cnt, &counters[CNT_KEPT]);
}

// kept[] (one byte per entry) as one bit per entry, little-endian inside a byte: a wave packs 64
// entries with one ballot (the multi-GPU gather moves n / 8 bytes instead of n)
__global__ __launch_bounds__(256) void pack_mask_kernel(const uint8_t *__restrict__ kept, uint64_t n,
                                                        uint8_t *__restrict__ bits)
{
    const uint64_t n_round = (n + 63) & ~63ull; // (whole waves: the ballot needs every lane)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long bal = __ballot(i < n && kept[i] != 0);
        const uint32_t lane = threadIdx.x & 63;
        if (lane < 8 && (i - lane) + 8ull * lane < n) bits[((i - lane) >> 3) + lane] = (uint8_t)(bal >> (8 * lane));
    }
}

inline uint32_t grid_of(uint64_t work, int block, uint32_t cap)
{
    uint64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

} // namespace

hipError_t launch_uf_components(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                                uint32_t *comp, uint32_t *lab, uint32_t n, uint32_t n_edges_hint,
                                hipStream_t s)
{
    if (n == 0) return hipSuccess;
    uf_union_kernel<<<grid_of(n_edges_hint, 256, 4096), 256, 0, s>>>(edges, counters, edge_cap, comp);
    uf_flatten_kernel<<<grid_of(n, 256, 4096), 256, 0, s>>>(comp, lab, n);
    return hipGetLastError();
}

hipError_t launch_pack_mask(const uint8_t *kept, uint64_t n, uint8_t *bits, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    pack_mask_kernel<<<grid_of(n, 256, 2048), 256, 0, s>>>(kept, n, bits);
    return hipGetLastError();
}

hipError_t launch_map_finalize(const uint32_t *comp, const uint32_t *lab, const RangeTask *ranges,
                               uint32_t n_ranges, uint32_t n, uint8_t *kept, uint32_t *root,
                               unsigned long long *counters, bool find, hipStream_t s)
{
    if (n == 0 || (ranges && n_ranges == 0)) return hipSuccess;
    const uint32_t grid = ranges ? n_ranges : grid_of(n, 1024, 1024);
    if (find) map_finalize_kernel<true><<<grid, 256, 0, s>>>(comp, lab, ranges, n, kept, root, counters);
    else map_finalize_kernel<false><<<grid, 256, 0, s>>>(comp, lab, ranges, n, kept, root, counters);
    return hipGetLastError();
}

hipError_t launch_uf_union_list(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                                uint32_t *parent, uint32_t n_edges_hint, hipStream_t s)
{
    uf_union_kernel<<<grid_of(n_edges_hint, 256, 4096), 256, 0, s>>>(edges, counters, edge_cap, parent);
    return hipGetLastError();
}

hipError_t launch_dag_flat_round(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                                 const uint32_t *comp, uint32_t *lab, uint32_t *changed, int round,
                                 hipStream_t s)
{
    dag_flat_hook_kernel<<<grid_of(edge_cap, 256 * 8, 512), 256, 0, s>>>(edges, counters, edge_cap, comp, lab,
                                                                        changed, round);
    return hipGetLastError();
}

hipError_t launch_uf_flatten(uint32_t *parent, uint32_t *lab, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    uf_flatten_kernel<<<grid_of(n, 256, 4096), 256, 0, s>>>(parent, lab, n);
    return hipGetLastError();
}

} // namespace umihip
