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
// sit in the edge list.  (Reading the last staged edges of the segment index's pair kernel where its
// blocks leave them, 4,096 private slots, instead of moving them to the list first saved two small
// launches and cost the first round 47 us instead of 13: a slot per wave means 2,000 waves, and every
// one of them sends its own atomic for the giant component's word.)
struct CollapseArgs {
    uint32_t *parent;       // in: the forest; out: comp[v] = root of v's set (flat)
    uint32_t *lab;          // lab[c] = smallest set that reaches set c along one-way pairs
    const uint2 *edges;     // the list (one-way pairs and, flagged, symmetric ones: skipped here)
    uint32_t edge_cap;
    const RangeTask *ranges; // the entries the collapse covers (the fused kernel finished the others); null: all n
    uint32_t n_ranges, n;
    uint8_t *kept;
    uint32_t *root;          // may be null
    unsigned long long *counters;
    uint32_t *changed;       // [MAX_ROUNDS_PER_SYNC] round r moved a label
    const uint2 *priv_edges; // the pair kernel's private slots (may be null), priv_blocks of SEG_PRIV_CAP edges
    const uint32_t *priv_cnt;
    uint32_t priv_blocks;
};

// the length of the list: what the pair kernels appended themselves and what the flatten launch moved behind it
__device__ __forceinline__ uint32_t list_length(const CollapseArgs &a)
{
    const unsigned long long ne = a.counters[CNT_EDGES] + a.counters[CNT_EDGES_MOVED];
    return ne < a.edge_cap ? (uint32_t)ne : a.edge_cap;
}

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
    if (lab) lab[v] = v;
}

// One round along the one-way pairs over the flattened sets: lab[set of v] = min(.., lab[set of u])
// for every pair u -> v of the list (directional.rs:38-39: v falls to whatever
// removes u).  No pointer jump over lab[] behind it: a chain of one-way pairs is as deep as the freq
// ladder it descends, two or three steps, and a pass over all of lab[] per round costs more than the
// round it might save.  Only as many blocks walk the list as it needs (the count is on the
// device): the fewer waves, the more of a hot word's hooks meet in one wave's registers.
// Returns whether this thread saw a label to move.  Every lane of a wave makes the same trips.
__device__ __forceinline__ bool one_way_round(const CollapseArgs &a)
{
    const uint32_t E = list_length(a);
    constexpr uint32_t PER_BLOCK = 256 * 8;
    const uint32_t active = min(gridDim.x, (E + PER_BLOCK - 1) / PER_BLOCK);
    bool any = false;
    HotMin hot;
    auto pair = [&](uint2 uv) {
        bool todo = false;
        uint32_t cv = 0, lu = 0;
        if (!(uv.x & SYM_FLAG)) {
            const uint32_t cu = a.parent[uv.x];
            cv = a.parent[uv.y];
            lu = a.lab[cu];
            todo = lu < a.lab[cv];
        }
        any |= todo;
        wave_atomic_min(a.lab, cv, lu, todo, hot);
    };
    if (blockIdx.x < active)
        for (uint32_t e0 = blockIdx.x * blockDim.x; e0 < E; e0 += active * blockDim.x) {
            const uint32_t e = e0 + threadIdx.x;
            pair(e < E ? a.edges[e] : make_uint2(SYM_FLAG, 0u));
        }
    hot_flush(a.lab, hot);
    return any;
}

// The same walk without a store: would a round still move a label?  (Rides in the finalize launch as
// blocks of its own, first_block .. gridDim.x - 1: lab[] is only read there, by both.)
__device__ __forceinline__ bool one_way_check(const CollapseArgs &a, uint32_t first_block)
{
    const uint32_t E = list_length(a);
    const uint32_t n_blocks = gridDim.x - first_block, me = blockIdx.x - first_block;
    bool any = false;
    for (uint32_t e = me * blockDim.x + threadIdx.x; e < E; e += n_blocks * blockDim.x) {
        const uint2 uv = a.edges[e];
        if (!(uv.x & SYM_FLAG)) any |= a.lab[a.parent[uv.x]] < a.lab[a.parent[uv.y]];
    }
    return any;
}

__device__ __forceinline__ unsigned int finalize_entry(const CollapseArgs &a, uint32_t i)
{
    const uint32_t l = a.lab[a.parent[i]]; // label[v] = lab[comp[v]] (directional.rs:30-54,78-88)
    const bool kp = l == i;                // (deduplicate_sam.rs:217-231)
    a.kept[i] = kp ? 1 : 0;
    if (a.root) a.root[i] = l;
    return kp ? 1u : 0u;
}

// The phases are separate launches: flatten | a round per launch, each a no-op once the one before it
// was quiet | kept, root, survivor count.  One launch for all of them with grid-wide barriers in
// between was built and measured: 0.21 ms against 0.065 for the five launches -- a barrier that
// 1,536 blocks pass costs ~10 us however it is built (one counter: the word is served at ~90
// accesses per microsecond; grouped arrivals and a flag line per block: the slowest block's phase
// plus two trips to the memory side), a release / acquire fence per wave writes back and invalidates
// its XCD's L2 every time (0.37 ms with 512 blocks, 0.90 with 1,536), and without fences every load
// of parent[] and lab[] has to go past the L2 -- while launches the host has enqueued ahead follow
// each other without a gap on this chip.
__global__ __launch_bounds__(256) void dag_flat_hook_kernel(CollapseArgs a, int round)
{
    if (round > 0 && a.changed[round - 1] == 0) return;
    if (one_way_round(a)) a.changed[round] = 1;
}

// The private slots to the list, 64 slots per block (blocks behind the flatten pass's own, in the same
// launch: was a scan kernel and a move kernel, 11 us).  A block adds up the counts of the slots before
// its group itself (at most 4,096 words), scans its own 64, and moves its slots' edges behind what the
// list holds; the last group says how many there were in all.  Nobody writes CNT_EDGES meanwhile.
constexpr uint32_t GATHER_SLOTS = 64;
__device__ __forceinline__ void gather_private_edges(const CollapseArgs &a, uint32_t group)
{
    __shared__ unsigned long long wsum[4];
    __shared__ uint32_t pre[GATHER_SLOTS + 1];
    const uint32_t s0 = group * GATHER_SLOTS, s1 = min(a.priv_blocks, s0 + GATHER_SLOTS);
    unsigned long long before = 0;
    for (uint32_t i = threadIdx.x; i < s0; i += blockDim.x) before += a.priv_cnt[i];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = before;
    if (threadIdx.x < 64) { // the group's own counts, exclusive
        const uint32_t c = s0 + threadIdx.x < s1 ? min(a.priv_cnt[s0 + threadIdx.x], SEG_PRIV_CAP) : 0u;
        uint32_t incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += up;
        }
        pre[threadIdx.x] = incl - c;
        if (threadIdx.x == 63) pre[GATHER_SLOTS] = incl;
    }
    __syncthreads();
    const unsigned long long first = a.counters[CNT_EDGES] + wsum[0] + wsum[1] + wsum[2] + wsum[3];
    uint2 *list = const_cast<uint2 *>(a.edges);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t j = wave; s0 + j < s1; j += blockDim.x >> 6) {
        const uint32_t cnt = pre[j + 1] - pre[j];
        for (uint32_t i = lane; i < cnt; i += 64) {
            const unsigned long long pos = first + pre[j] + i;
            if (pos < a.edge_cap) list[pos] = a.priv_edges[(size_t)(s0 + j) * SEG_PRIV_CAP + i];
        }
    }
    if (s1 == a.priv_blocks && threadIdx.x == 0) a.counters[CNT_EDGES_MOVED] = first + pre[GATHER_SLOTS] - a.counters[CNT_EDGES];
}

__global__ __launch_bounds__(256) void uf_flatten_kernel(CollapseArgs a, uint32_t entry_blocks)
{
    if (blockIdx.x >= entry_blocks) {
        gather_private_edges(a, blockIdx.x - entry_blocks);
        return;
    }
    if (a.ranges) {
        for (uint32_t r = blockIdx.x; r < a.n_ranges; r += entry_blocks) {
            const RangeTask rt = a.ranges[r];
            for (uint32_t i = rt.start + threadIdx.x; i < rt.end; i += blockDim.x) flatten_entry(a.parent, a.lab, i);
        }
    } else {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += entry_blocks * blockDim.x)
            flatten_entry(a.parent, a.lab, i);
    }
}

// kept / root / survivor count of the entries of ranges (null: all n) from comp and lab, by the first
// entry_blocks blocks; the blocks behind them (check_round >= 0) look whether round check_round would
// still move a label -- the last of the rounds enqueued ahead of the host's look changes nothing in
// the common case and is there only to say so: as a check beside the finalize pass it costs no launch
// of its own (the result stands if changed[check_round] stays 0; else the host runs on and finalizes again)
__global__ __launch_bounds__(256) void map_finalize_kernel(CollapseArgs a, uint32_t entry_blocks, int check_round)
{
    if (blockIdx.x >= entry_blocks) {
        if (check_round > 0 && a.changed[check_round - 1] == 0) return; // (the round before was quiet already)
        if (one_way_check(a, entry_blocks)) a.changed[check_round] = 1;
        return;
    }
    unsigned int cnt = 0;
    if (a.ranges) {
        for (uint32_t r = blockIdx.x; r < a.n_ranges; r += entry_blocks) {
            const RangeTask rt = a.ranges[r];
            for (uint32_t i = rt.start + threadIdx.x; i < rt.end; i += blockDim.x) cnt += finalize_entry(a, i);
        }
    } else {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += entry_blocks * blockDim.x) cnt += finalize_entry(a, i);
    }
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
    a.ranges = d.ranges;
    a.n_ranges = d.n_ranges;
    a.n = d.n;
    a.kept = d.kept;
    a.root = d.root;
    a.counters = d.counters;
    a.changed = d.changed;
    a.priv_edges = d.priv_edges;
    a.priv_cnt = d.priv_cnt;
    a.priv_blocks = d.priv_blocks;
    return a;
}

uint32_t entries_grid(const CollapseDesc &d, uint32_t cap)
{
    return d.ranges ? std::max(1u, std::min(d.n_ranges, cap)) : grid_of(d.n, 256, cap);
}

} // namespace

// the control block to the host's pinned mirror by a wave's own stores (a DMA copy of 256 bytes
// starts ~20 us after the kernel before it ends; a launch follows it within a few), then a sequence
// number in a line of its own: the host may watch for it instead of asking the runtime to wait for
// the stream (whose wake-up costs a call of 0.12 ms some 10 us)
namespace {
__global__ __launch_bounds__(64) void control_to_host_kernel(const unsigned long long *__restrict__ d_ctrl,
                                                             unsigned long long *h_ctrl, uint32_t n_words,
                                                             unsigned long long *h_seq, unsigned long long seq)
{
    for (uint32_t i = threadIdx.x; i < n_words; i += 64)
        __hip_atomic_store(&h_ctrl[i], __hip_atomic_load(&d_ctrl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __builtin_amdgcn_s_waitcnt(0); // (every lane's stores have left before the one below)
    if (h_seq && threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
} // namespace
hipError_t launch_control_to_host(const void *d_ctrl, void *h_ctrl, size_t bytes, hipStream_t s, unsigned long long *h_seq,
                                  unsigned long long seq)
{
    control_to_host_kernel<<<1, 64, 0, s>>>((const unsigned long long *)d_ctrl, (unsigned long long *)h_ctrl,
                                            (uint32_t)(bytes / 8), h_seq, seq);
    return hipGetLastError();
}

hipError_t launch_collapse_flatten(const CollapseDesc &d, hipStream_t s)
{
    const bool entries = !(d.n == 0 || (d.ranges && d.n_ranges == 0));
    const uint32_t entry_blocks = entries ? entries_grid(d, 4096) : 0u;
    const uint32_t gather_blocks = d.priv_blocks ? (d.priv_blocks + GATHER_SLOTS - 1) / GATHER_SLOTS : 0u;
    if (entry_blocks + gather_blocks == 0) return hipSuccess;
    uf_flatten_kernel<<<entry_blocks + gather_blocks, 256, 0, s>>>(collapse_args(d), entry_blocks);
    return hipGetLastError();
}

hipError_t launch_collapse_round(const CollapseDesc &d, int round, hipStream_t s)
{
    dag_flat_hook_kernel<<<grid_of(d.edge_cap, 256 * 8, 512), 256, 0, s>>>(collapse_args(d), round);
    return hipGetLastError();
}

hipError_t launch_collapse_finalize(const CollapseDesc &d, hipStream_t s, int check_round)
{
    if (d.n == 0 || (d.ranges && d.n_ranges == 0)) return hipSuccess;
    const uint32_t entry_blocks = entries_grid(d, 2048);
    const uint32_t check_blocks = check_round >= 0 ? grid_of(d.edge_cap, 256 * 8, 256) : 0u;
    map_finalize_kernel<<<entry_blocks + check_blocks, 256, 0, s>>>(collapse_args(d), entry_blocks, check_round);
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
