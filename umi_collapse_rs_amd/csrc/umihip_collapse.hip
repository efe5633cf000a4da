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
#include <algorithm>

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

// ---- the batched directional collapse behind the pair kernels ------------------------------------
// Inputs: parent[] (= label[]) holds the union-find forest of the symmetric pairs; the one-way pairs
// sit in the edge list and in the private slots the segment index's pair kernel left its last staged
// edges in (one slot per block of that kernel: nothing is appended to the list at the end of it, so
// no storm of atomics on the list's counter and no launches to move the slots).
struct CollapseArgs {
    uint32_t *parent;       // in: the forest; out: comp[v] = root of v's set (flat)
    uint32_t *lab;          // lab[c] = smallest set that reaches set c along one-way pairs
    const uint2 *edges;     // the list (one-way pairs and, flagged, symmetric ones: skipped here)
    uint32_t edge_cap;
    const uint2 *priv_edges; // [n_slots * SEG_PRIV_CAP], may be null
    const uint32_t *priv_cnt; // [n_slots] entries in each slot
    uint32_t n_slots;
    const RangeTask *ranges; // the entries the collapse covers (the fused kernel finished the others); null: all n
    uint32_t n_ranges, n;
    uint8_t *kept;
    uint32_t *root;          // may be null
    unsigned long long *counters;
    uint32_t *changed;       // [MAX_ROUNDS_PER_SYNC] round r moved a label
    uint32_t *sync;          // [COLLAPSE_SYNC_WORDS]: barrier arrivals, give-up flag, rounds run, slot edges
    uint32_t *flags;         // [grid] the barrier's flag word of every block (values only ever grow: epoch + barrier number)
    uint32_t epoch;
    uint32_t max_rounds;
};

template <class F> __device__ __forceinline__ void collapse_entries(const CollapseArgs &a, F f)
{
    if (a.ranges) {
        for (uint32_t r = blockIdx.x; r < a.n_ranges; r += gridDim.x) {
            const RangeTask rt = a.ranges[r];
            for (uint32_t i = rt.start + threadIdx.x; i < rt.end; i += blockDim.x) f(i);
        }
    } else {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) f(i);
    }
}

// comp[v] = root of v (the smallest index of its set); lab[v] = v
__device__ __forceinline__ void flatten_entry(uint32_t *parent, uint32_t *lab, uint32_t v)
{
    uint32_t r = ld_parent(&parent[v]);
    if (r != v) {
        for (;;) {
            const uint32_t p = ld_parent(&parent[r]);
            if (p == r) break;
            r = p;
        }
        st_parent(&parent[v], r);
    }
    if (lab) st_parent(&lab[v], v); // (agent scope like parent[]: the fused kernel's later phases read it on other XCDs)
}

// One round along the one-way pairs over the flattened sets: lab[set of v] = min(.., lab[set of u])
// for every pair u -> v of the list and of the slots (directional.rs:38-39: v falls to whatever
// removes u).  No pointer jump over lab[] behind it: a chain of one-way pairs is as deep as the freq
// ladder it descends, two or three steps, and a pass over all of lab[] per round costs more than the
// round it might save.  Only as many blocks walk the list as it needs (the count is on the
// device): the fewer waves, the more of a hot word's hooks meet in one wave's registers.
// Returns whether this thread saw a label to move.  Every lane of a wave makes the same trips.
__device__ __forceinline__ bool one_way_round(const CollapseArgs &a, bool count_slots)
{
    const unsigned long long ne = a.counters[CNT_EDGES];
    const uint32_t E = ne < a.edge_cap ? (uint32_t)ne : a.edge_cap;
    constexpr uint32_t PER_BLOCK = 256 * 8;
    const uint32_t active = min(gridDim.x, (E + PER_BLOCK - 1) / PER_BLOCK);
    bool any = false;
    HotMin hot;
    auto pair = [&](uint2 uv) {
        bool todo = false;
        uint32_t cv = 0, lu = 0;
        if (!(uv.x & SYM_FLAG)) {
            const uint32_t cu = ld_parent(&a.parent[uv.x]);
            cv = ld_parent(&a.parent[uv.y]);
            lu = ld_parent(&a.lab[cu]);
            todo = lu < ld_parent(&a.lab[cv]);
        }
        any |= todo;
        wave_atomic_min(a.lab, cv, lu, todo, hot);
    };
    if (blockIdx.x < active)
        for (uint32_t e0 = blockIdx.x * blockDim.x; e0 < E; e0 += active * blockDim.x) {
            const uint32_t e = e0 + threadIdx.x;
            pair(e < E ? a.edges[e] : make_uint2(SYM_FLAG, 0u));
        }
    // the slots: a wave per slot
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    unsigned int in_slots = 0;
    for (uint32_t sl = blockIdx.x * wpb + wave; sl < a.n_slots; sl += gridDim.x * wpb) {
        const uint32_t cnt = min((uint32_t)__builtin_amdgcn_readfirstlane((int)a.priv_cnt[sl]), SEG_PRIV_CAP);
        in_slots += lane == 0 ? cnt : 0u;
        for (uint32_t i0 = 0; i0 < cnt; i0 += 64)
            pair(i0 + lane < cnt ? a.priv_edges[(size_t)sl * SEG_PRIV_CAP + i0 + lane] : make_uint2(SYM_FLAG, 0u));
    }
    hot_flush(a.lab, hot);
    if (count_slots && in_slots) atomicAdd(&a.sync[3], in_slots); // (statistics: the list's counter does not see them)
    return any;
}

__device__ __forceinline__ unsigned int finalize_entry(const CollapseArgs &a, uint32_t i)
{
    const uint32_t l = ld_parent(&a.lab[ld_parent(&a.parent[i])]); // label[v] = lab[comp[v]] (directional.rs:30-54,78-88)
    const bool kp = l == i;                // (deduplicate_sam.rs:217-231)
    a.kept[i] = kp ? 1 : 0;
    if (a.root) a.root[i] = l;
    return kp ? 1u : 0u;
}

// All of it in one launch: flatten | rounds until one moves nothing | kept, root, survivor count, with
// grid-wide barriers in between (the grid is small enough to be resident at once: two blocks per
// CU).  A barrier that is not passed within COLLAPSE_WAIT_TICKS -- blocks of this grid that cannot
// start because other work holds the CUs: several contexts on one card -- makes every block give up
// (sync[1] = 1): each phase is idempotent or monotone, so the host then runs the same phases as
// separate launches over whatever state it finds.  Every wave reaches an exit either way.
constexpr unsigned long long COLLAPSE_WAIT_TICKS = 400000; // 4 ms of the 100 MHz wall clock
__global__ __launch_bounds__(256) void collapse_fused_kernel(CollapseArgs a)
{
    // The barrier.  Arrival: a block adds itself to the counter of its group of 32 blocks, the block
    // that completes a group adds the group to the grid's counter, the one that completes that clears
    // the counters (nobody arrives again before it lets them go) and writes the barrier's number into
    // one flag word PER BLOCK; every other block polls its own word.  Every counter and flag has a
    // 64-byte line of its own: a line is served at ~90 accesses per microsecond, which 1,500 arrivals
    // or pollers sharing lines would queue at (one word for all: 0.5 ms for the launch; 16 flags per
    // line: 0.23).  No cache maintenance: everything the phases hand each other (parent[], lab[], the
    // flags) is read and written at agent scope, past the XCDs' L2s -- a release / acquire fence per
    // wave writes back and invalidates its XCD's L2 each time (0.37 ms with 512 blocks, 0.90 with 1,536).
    __shared__ uint32_t pass, last, want;
    uint32_t gen = 0;
    constexpr uint32_t GROUP = 32, LINE = COLLAPSE_LINE_WORDS;
    const uint32_t n_groups = (gridDim.x + GROUP - 1) / GROUP;
    const uint32_t my_group = blockIdx.x / GROUP, group_size = min(GROUP, gridDim.x - my_group * GROUP);
    uint32_t *my_flag = a.flags + (size_t)blockIdx.x * LINE;
    uint32_t *group_cnt = a.flags + (size_t)COLLAPSE_MAX_GRID * LINE, *top_cnt = group_cnt + (size_t)COLLAPSE_GROUP_WORDS * LINE;
    auto barrier = [&]() -> bool {
        __syncthreads(); // (every wave's accesses of the phase are complete: s_waitcnt vmcnt(0) before s_barrier)
        if (threadIdx.x == 0) {
            gen++;
            want = a.epoch + gen;
            pass = 1;
            uint32_t fin = 0;
            if (atomicAdd(&group_cnt[(size_t)my_group * LINE], 1u) + 1u == group_size)
                fin = atomicAdd(top_cnt, 1u) + 1u == n_groups ? 1u : 0u;
            last = fin;
        }
        __syncthreads();
        if (last) {
            for (uint32_t i = threadIdx.x; i < n_groups; i += blockDim.x)
                __hip_atomic_store(&group_cnt[(size_t)i * LINE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) __hip_atomic_store(top_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads(); // (the counters are clear before anybody is let go)
            for (uint32_t i = threadIdx.x; i < gridDim.x; i += blockDim.x)
                __hip_atomic_store(&a.flags[(size_t)i * LINE], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            for (uint32_t spin = 0;; spin++) {
                const uint32_t f = __hip_atomic_load(my_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int32_t)(f - want) >= 0) break;
                if ((spin & 15u) == 15u &&
                    (__hip_atomic_load(&a.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                     wall_clock64() - t0 > COLLAPSE_WAIT_TICKS)) {
                    __hip_atomic_store(&a.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pass = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();
        return pass != 0;
    };
    collapse_entries(a, [&](uint32_t v) { flatten_entry(a.parent, a.lab, v); });
    if (!barrier()) return;
    uint32_t r = 0;
    for (; r < a.max_rounds; r++) {
        if (one_way_round(a, r == 0)) __hip_atomic_store(&a.changed[r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!barrier()) return;
        if (__hip_atomic_load(&a.changed[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            r++;
            break;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.sync[2] = r; // rounds run (the last one quiet, unless max_rounds)
    unsigned int cnt = 0;
    collapse_entries(a, [&](uint32_t i) { cnt += finalize_entry(a, i); });
    block_count_add(cnt, &a.counters[CNT_KEPT]);
}

// ---- the same phases as separate launches (what the host falls back to, and the rounds beyond
// the fused kernel's) ----
__global__ __launch_bounds__(256) void dag_flat_hook_kernel(CollapseArgs a, int round)
{
    if (round > 0 && a.changed[round - 1] == 0) return;
    if (one_way_round(a, false)) a.changed[round] = 1;
}

__global__ __launch_bounds__(256) void uf_flatten_kernel(CollapseArgs a) { collapse_entries(a, [&](uint32_t v) { flatten_entry(a.parent, a.lab, v); }); }

// kept / root / survivor count of the entries of ranges (null: all n) from comp and lab
__global__ __launch_bounds__(256) void map_finalize_kernel(CollapseArgs a)
{
    unsigned int cnt = 0;
    collapse_entries(a, [&](uint32_t i) { cnt += finalize_entry(a, i); });
    block_count_add(cnt, &a.counters[CNT_KEPT]);
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

CollapseArgs collapse_args(const CollapseDesc &d)
{
    CollapseArgs a;
    a.parent = d.parent;
    a.lab = d.lab;
    a.edges = d.edges;
    a.edge_cap = d.edge_cap;
    a.priv_edges = d.priv_edges;
    a.priv_cnt = d.priv_cnt;
    a.n_slots = d.priv_edges ? d.n_slots : 0u;
    a.ranges = d.ranges;
    a.n_ranges = d.n_ranges;
    a.n = d.n;
    a.kept = d.kept;
    a.root = d.root;
    a.counters = d.counters;
    a.changed = d.changed;
    a.sync = d.sync;
    a.flags = d.flags;
    a.epoch = d.epoch;
    a.max_rounds = MAX_ROUNDS_PER_SYNC;
    return a;
}

uint32_t entries_grid(const CollapseDesc &d, uint32_t cap)
{
    return d.ranges ? std::max(1u, std::min(d.n_ranges, cap)) : grid_of(d.n, 256, cap);
}

} // namespace

hipError_t launch_collapse_fused(const CollapseDesc &d, uint32_t n_cus, hipStream_t s)
{
    if (d.n == 0 || (d.ranges && d.n_ranges == 0)) return hipSuccess;
    // flatten and finalize chase pointers, one dependent trip to the memory side per step: they want
    // every wave the chip holds.  The grid must also be resident at once (the barriers): what the
    // runtime says fits, less a quarter -- it names an upper bound (the segment index's pair kernel
    // found the last blocks of a nominally resident grid starting late).
    static int per_cu = 0;
    if (!per_cu) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, collapse_fused_kernel, 256, 0) != hipSuccess || nb < 1) nb = 2;
        per_cu = std::max(1, std::min(8, nb) * 3 / 4);
    }
    collapse_fused_kernel<<<std::min((uint32_t)per_cu * n_cus, COLLAPSE_MAX_GRID), 256, 0, s>>>(collapse_args(d));
    return hipGetLastError();
}

// the control block to the host's pinned mirror by a wave's own stores (a DMA copy of 256 bytes
// starts ~20 us after the kernel before it ends; a launch follows it within a few)
namespace {
__global__ __launch_bounds__(64) void control_to_host_kernel(const unsigned long long *__restrict__ d_ctrl,
                                                             unsigned long long *h_ctrl, uint32_t n_words)
{
    for (uint32_t i = threadIdx.x; i < n_words; i += 64)
        __hip_atomic_store(&h_ctrl[i], __hip_atomic_load(&d_ctrl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
}
} // namespace
hipError_t launch_control_to_host(const void *d_ctrl, void *h_ctrl, size_t bytes, hipStream_t s)
{
    control_to_host_kernel<<<1, 64, 0, s>>>((const unsigned long long *)d_ctrl, (unsigned long long *)h_ctrl,
                                            (uint32_t)(bytes / 8));
    return hipGetLastError();
}

hipError_t launch_collapse_flatten(const CollapseDesc &d, hipStream_t s)
{
    if (d.n == 0 || (d.ranges && d.n_ranges == 0)) return hipSuccess;
    uf_flatten_kernel<<<entries_grid(d, 4096), 256, 0, s>>>(collapse_args(d));
    return hipGetLastError();
}

hipError_t launch_collapse_round(const CollapseDesc &d, int round, hipStream_t s)
{
    const uint32_t by_list = grid_of(d.edge_cap, 256 * 8, 512), by_slots = d.priv_edges ? (d.n_slots + 3) / 4 : 0u;
    dag_flat_hook_kernel<<<std::max(by_list, std::min(by_slots, 512u)), 256, 0, s>>>(collapse_args(d), round);
    return hipGetLastError();
}

hipError_t launch_collapse_finalize(const CollapseDesc &d, hipStream_t s)
{
    if (d.n == 0 || (d.ranges && d.n_ranges == 0)) return hipSuccess;
    map_finalize_kernel<<<entries_grid(d, 2048), 256, 0, s>>>(collapse_args(d));
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void uf_flatten_all_kernel(uint32_t *parent, uint32_t *lab, uint32_t n)
{
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) flatten_entry(parent, lab, v);
}
} // namespace

hipError_t launch_uf_components(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                                uint32_t *comp, uint32_t *lab, uint32_t n, uint32_t n_edges_hint,
                                hipStream_t s)
{
    if (n == 0) return hipSuccess;
    uf_union_kernel<<<grid_of(n_edges_hint, 256, 4096), 256, 0, s>>>(edges, counters, edge_cap, comp);
    uf_flatten_all_kernel<<<grid_of(n, 256, 4096), 256, 0, s>>>(comp, lab, n);
    return hipGetLastError();
}

hipError_t launch_pack_mask(const uint8_t *kept, uint64_t n, uint8_t *bits, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    pack_mask_kernel<<<grid_of(n, 256, 2048), 256, 0, s>>>(kept, n, bits);
    return hipGetLastError();
}

hipError_t launch_uf_union_list(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                                uint32_t *parent, uint32_t n_edges_hint, hipStream_t s)
{
    uf_union_kernel<<<grid_of(n_edges_hint, 256, 4096), 256, 0, s>>>(edges, counters, edge_cap, parent);
    return hipGetLastError();
}

} // namespace umihip
