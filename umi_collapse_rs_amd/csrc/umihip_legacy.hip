// Round-1 tile kernels of the large buckets (gfx950), kept as cross-checks: the bit-sliced all-pairs
// mask kernel (K1b), the key-sorted scan + item walk (K1r / K1t) and the exact check of their
// overflow list.  Built only with -DUMIHIP_DEV (make dev -> libumihip_dev.so); the shipped
// library takes every bucket through the fused kernel, the popcount chunks / tiles and the
// segment index (umihip_kernels.hip, umihip_seg.hip).
//
// What they replace in the reference (tkob-vh/umi-collapse-rs): Naive::remove_near's linear scans
// (src/data/naive.rs:26-40) with BitSet::bit_count_xor / umi_dist (src/utils/bitset.rs:77-91,
// src/utils/mod.rs:24-26), all pairs of a bucket at once.  Integer/bitwise, no MFMA.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <utility>

#include "umihip_internal.h"
#include "umihip_device.h"

namespace umihip {

namespace {

inline uint32_t grid_for(uint64_t work, int block, uint32_t cap = 2048)
{
    uint64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

// ---- bit-sliced all-pairs filter (large buckets) --------------------------------
// Rows are held as bit planes of the 2-bit base code (plane 2i+j = bit j of base i; one
// VGPR word = that bit of 32 consecutive rows), so one lane carries 32*G rows.  One
// column at a time is applied as 2L wave-uniform masks (0 / ~0) read from an LDS tile:
//     mismatch_i = (P[2i] ^ c[2i]) | (P[2i+1] ^ c[2i+1])          v_xor + v_bitop3
// and a sticky counter saturating at K+1 runs over the bases in v_bitop3_b32 (any
// 3-input boolean in one full-rate op on gfx950).  After the last base ~s[K+1] marks
// the rows within distance K of the column: exact on the N-folded 2-bit code, hence
// never above the reference distance; hits go through verify_pair like the other kernel.
// K = 1 costs 3.5 full-rate VALU ops per base per 32 pairs (2 for the mask, 1.5 for the
// counter with two bases folded per step) = 1.3 lane-ops per pair at L = 12.
__device__ __forceinline__ uint32_t bit_of(uint32_t k, int b) { return (k >> b) & 1u; }
__device__ __forceinline__ uint32_t bit_of(uint64_t k3, int b)
{ // 64-bit filter keys keep the 3-bit layout: 2-bit code bit j of base i sits at 3i+j
    const int pos = 3 * (b >> 1) + (b & 1);
    return pos < 63 ? (uint32_t)(k3 >> pos) & 1u : 0u; // base 21 is padding (umi_len <= 21)
}

template <typename KeyT>
__global__ __launch_bounds__(64) void build_planes_kernel(const KeyT *__restrict__ fkey,
                                                          const PlaneTask *__restrict__ tasks,
                                                          uint32_t *__restrict__ planes, int np)
{
    const PlaneTask t = tasks[blockIdx.x];
    const uint32_t row = t.row0 + threadIdx.x;
    const KeyT key = row < t.bucket_end ? fkey[row] : (KeyT)0;
    unsigned long long mine = 0; // lane b keeps plane b of the task's two row groups
    for (int b = 0; b < np; b++) {
        const unsigned long long bal = __ballot(bit_of(key, b));
        if ((int)threadIdx.x == b) mine = bal;
    }
    if ((int)threadIdx.x < np) { // group-major: a group's np plane words are contiguous
        uint32_t *dst = planes + t.plane_off + (uint64_t)t.group * np + threadIdx.x;
        dst[0] = (uint32_t)mine;
        if (t.group + 1 < t.ngroups) dst[np] = (uint32_t)(mine >> 32);
    }
}


// (any, two) = rows where at least one / at least two of the unit masks m(u), u in [U0, U1),
// are set: units in triples give (any, two) with one or3 and one majority each, two groups
// merge as two = twoA | twoB | (anyA & anyB).  6 ops for 6 units.
template <int U0, int U1, class F>
__device__ __forceinline__ void any_two_of_units(F m, uint32_t &any_acc, uint32_t &two_acc)
{
    any_acc = 0;
    two_acc = 0;
#pragma unroll
    for (int u = U0; u < U1; u += 3) {
        const int rem = U1 - u < 3 ? U1 - u : 3;
        uint32_t any, two = 0;
        if (rem == 3) {
            const uint32_t ma = m(u), mb = m(u + 1), mc = m(u + 2);
            any = BITOP3(ma, mb, mc, TT_A | TT_B | TT_C);
            two = BITOP3(ma, mb, mc, (TT_A & TT_B) | (TT_A & TT_C) | (TT_B & TT_C));
        } else if (rem == 2) {
            const uint32_t ma = m(u), mb = m(u + 1);
            any = ma | mb;
            two = ma & mb;
        } else {
            any = m(u);
        }
        if (u == U0) {
            any_acc = any;
            two_acc = two;
        } else {
            if (rem >= 2) two_acc |= BITOP3(two, any_acc, any, TT_A | (TT_B & TT_C));
            else two_acc = BITOP3(two_acc, any_acc, any, TT_A | (TT_B & TT_C));
            any_acc |= any; // dead (and dropped) when the caller only wants `two`
        }
    }
}

// sticky counters: s[l] |= rows with at least l set masks among m(u), u in [U0, U1); l = 1..K+1
template <int K, int U0, int U1, class F>
__device__ __forceinline__ void count_units(F m, uint32_t (&s)[K + 2])
{
#pragma unroll
    for (int u = U0; u < U1; u += (K == 0 ? 2 : 1)) {
        if (K == 0) { // "any unit differs": two units per op
            if (u + 1 < U1) s[1] = BITOP3(s[1], m(u), m(u + 1), TT_A | TT_B | TT_C);
            else s[1] |= m(u);
        } else {
            const uint32_t ma = m(u);
#pragma unroll
            for (int l = K + 1; l >= 2; l--) s[l] = BITOP3(s[l], s[l - 1], ma, TT_A | (TT_B & TT_C));
            s[1] |= ma;
        }
    }
}

// COLSPLIT = false: the block's 4 waves hold 4 x 64 x G different row groups and all walk
// every column (tiles of 8192*G rows, for very large buckets).  COLSPLIT = true: the 4 waves
// hold the SAME 64 x G row groups and take every 4th column of the staged tile, so a bucket of
// a few thousand entries still fills its lanes while the LDS staging is shared by 4 waves.
//
// PU > 0 (keys of the bucket sorted, so neighbouring columns share their high bases): the
// counter state after the PU highest units is kept per row group and recomputed only when a
// column's high bases differ from its predecessor's (one flag bit per column, set while the
// tile is staged); every column then costs its U - PU low units plus the merge.  Every pair is
// still evaluated; what is shared is the part of the evaluation that is equal for both columns.
template <typename KeyT, int LP, int G, int K, bool COLSPLIT, int GB, int PU>
__global__ __launch_bounds__(256) void bs_pair_kernel(PairArgs a)
{
    constexpr int THREADS = 256;
    constexpr int NP = 2 * LP;
    constexpr int U = LP / GB;    // units per key
    constexpr int LIVE = U - PU;  // units evaluated for every column (the low ones)
    static_assert(LP % GB == 0 && NP % 4 == 0, "padded base count must be a multiple of the unit");
    static_assert(PU == 0 || (GB == 2 && !COLSPLIT && PU < U), "prefix caching: 2-base units, wide tiles");
    __shared__ __attribute__((aligned(16))) uint32_t cmask[BS_COL_TILE * NP];
    __shared__ uint32_t runbits[BS_COL_TILE / 32];
    // filter hits of the current column tile: queued by the lane that finds them, checked
    // exactly by all 256 threads once the tile's columns are done (a hit found inside the
    // column loop would otherwise hold its whole wave for one serial verify per set bit)
    constexpr uint32_t HITQ = 1024;
    __shared__ uint2 hitq[HITQ];
    __shared__ unsigned int hitq_count;
    __shared__ EdgeStage stage;
    const BsTask *__restrict__ tp = a.bs_tasks + blockIdx.x;
    const uint32_t bucket_start = __builtin_amdgcn_readfirstlane(tp->bucket_start);
    const uint32_t bucket_end = __builtin_amdgcn_readfirstlane(tp->bucket_end);
    const uint32_t group0 = __builtin_amdgcn_readfirstlane(tp->group0);
    const uint32_t ngroups = __builtin_amdgcn_readfirstlane(tp->ngroups);
    const uint32_t col0 = __builtin_amdgcn_readfirstlane(tp->col0);
    const uint32_t col1 = __builtin_amdgcn_readfirstlane(tp->col1);
    const bool diag = __builtin_amdgcn_readfirstlane(tp->diag) != 0; // wave-uniform
    const uint64_t plane_off = tp->plane_off;
    const KeyT *__restrict__ fkey = (const KeyT *)a.fkey;
    const uint32_t *__restrict__ planes = a.planes + plane_off;
    const int tid = threadIdx.x;
    const bool with_dist = a.mode == MODE_NEIGHBOURS;
    const uint32_t n_rows = bucket_end - bucket_start;

    if (tid == 0) {
        stage.count = 0;
        stage.candidates = 0;
        hitq_count = 0;
    }

    uint32_t p[G][NP];
    uint32_t valid[G];
    uint32_t rbase[G]; // bucket-relative index of the group's first row
    uint32_t pre[G][K + 2]; // PU > 0: counter state after the prefix units ((any, two) for K = 1)
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int l = 0; l < K + 2; l++) pre[g][l] = 0u;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const uint32_t grp = COLSPLIT ? group0 + g * 64 + (tid & 63) : group0 + g * THREADS + tid;
        // four planes per 128-bit load: they arrive, and stay, in an aligned register quad
#pragma unroll
        for (int q = 0; q < NP / 4; q++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (grp < ngroups) v = *reinterpret_cast<const uint4 *>(planes + (uint64_t)grp * NP + 4 * q);
            p[g][4 * q] = v.x;
            p[g][4 * q + 1] = v.y;
            p[g][4 * q + 2] = v.z;
            p[g][4 * q + 3] = v.w;
        }
        rbase[g] = grp * 32;
        valid[g] = rbase[g] >= n_rows ? 0u
                   : (n_rows - rbase[g] >= 32 ? 0xFFFFFFFFu : ((1u << (n_rows - rbase[g])) - 1u));
    }

    for (uint32_t c0 = col0; c0 < col1; c0 += BS_COL_TILE) {
        const uint32_t nc = min((uint32_t)BS_COL_TILE, col1 - c0);
        __syncthreads();
        // stage the tile: thread -> (column, quad of planes); the keys of a thread's columns are
        // loaded together, each quad is one 128-bit LDS write.  Word j of quad q holds the mask
        // of plane 4q + (j ^ 1): see unit_mask.
        {
            constexpr int QPC = NP / 4; // quads per column
            constexpr int ITEMS = (BS_COL_TILE * QPC + THREADS - 1) / THREADS;
            KeyT ck[ITEMS];
#pragma unroll
            for (int it = 0; it < ITEMS; it++) {
                const uint32_t w = (uint32_t)tid + (uint32_t)it * THREADS;
                const uint32_t c = w / QPC;
                ck[it] = c < nc ? fkey[c0 + c] : (KeyT)0;
            }
#pragma unroll
            for (int it = 0; it < ITEMS; it++) {
                const uint32_t w = (uint32_t)tid + (uint32_t)it * THREADS;
                const uint32_t c = w / QPC, q = w % QPC;
                if (c < nc) {
                    uint4 v;
                    v.x = bit_of(ck[it], (int)(4 * q + 1)) ? 0xFFFFFFFFu : 0u;
                    v.y = bit_of(ck[it], (int)(4 * q)) ? 0xFFFFFFFFu : 0u;
                    v.z = bit_of(ck[it], (int)(4 * q + 3)) ? 0xFFFFFFFFu : 0u;
                    v.w = bit_of(ck[it], (int)(4 * q + 2)) ? 0xFFFFFFFFu : 0u;
                    *reinterpret_cast<uint4 *>(&cmask[c * NP + 4 * q]) = v;
                }
            }
        }
        if (PU > 0 && tid < BS_COL_TILE) { // waves 0 and 1: one flag per column of the tile
            bool newrun = false;
            if ((uint32_t)tid < nc) {
                newrun = (tid & 31) == 0; // the flags are consumed in words of 32 columns
                if (!newrun) {
                    constexpr int shift = PU > 0 ? (sizeof(KeyT) == 4 ? 2 : 3) * GB * LIVE : 0; // bits below the prefix
                    newrun = ((fkey[c0 + tid] ^ fkey[c0 + tid - 1]) >> shift) != 0;
                }
            }
            const unsigned long long bal = __ballot(newrun);
            if ((tid & 63) == 0) {
                runbits[2 * (tid >> 6)] = (uint32_t)bal;
                runbits[2 * (tid >> 6) + 1] = (uint32_t)(bal >> 32);
            }
        }
        __syncthreads();

        // The walk over the tile's columns, compiled twice: tiles on the bucket's diagonal mask
        // every hit word with "row < column", the others do not carry that code at all.
        auto walk_columns = [&](auto diag_tag) {
            constexpr bool DIAG = decltype(diag_tag)::value;
            constexpr int LQ = PU > 0 ? LIVE : NP / 4; // quads of the units evaluated per column
            // columns per group: one "does a run start here?" test and one "any hit?" test per
            // group (both are VALU->scalar round trips or scalar branches: a wave issues one
            // instruction per 4 cycles, so every one of them costs as much as a bitop3)
            constexpr int NCOL = COLSPLIT ? 2 : (LQ <= 2 ? 4 : 2);
            constexpr bool PRELOAD = LQ <= 2; // all masks of a group are fetched up front
            constexpr uint32_t CSTEP = COLSPLIT ? 4u : 1u;

            auto load_quads = [&](uint32_t c, int q0, int q1, uint32_t (&cm)[NP]) {
#pragma unroll
                for (int q = q0; q < q1; q++)
                    *reinterpret_cast<uint4 *>(&cm[4 * q]) =
                        *reinterpret_cast<const uint4 *>(&cmask[c * NP + 4 * q]);
            };
            // mismatch mask of a unit of GB consecutive bases (any of its 2*GB code bits
            // differs): "P ^ c", then one bitop3 "acc | (P ^ c)" per further plane.
            // gfx950 issues a VOP3 at half rate when its three source registers all have the
            // same parity (tools/bankprobe.hip).  Planes and masks both sit in aligned
            // register quads (128-bit loads), and the mask of plane b is word b ^ 1 of its
            // quad, so P and c always differ in parity, whatever register holds acc; the
            // leading xor is a bitop3 too (a VOP2 xor of mixed parity issues slower).
            auto unit_mask = [&](const uint32_t (&cm)[NP], int g, int u) -> uint32_t {
                uint32_t m = BITOP3(p[g][2 * GB * u], cm[(2 * GB * u) ^ 1], cm[(2 * GB * u) ^ 1], TT_A ^ TT_B);
#pragma unroll
                for (int b = 1; b < 2 * GB; b++)
                    m = BITOP3(m, p[g][2 * GB * u + b], cm[(2 * GB * u + b) ^ 1], TT_A | (TT_B ^ TT_C));
                return m;
            };
            // a column whose high bases differ from its predecessor's: new prefix state
            auto update_prefix = [&](uint32_t c) {
                uint32_t cm[NP];
                load_quads(c, LIVE, U, cm);
#pragma unroll
                for (int g = 0; g < G; g++) {
                    auto unit = [&](int u) { return unit_mask(cm, g, u); };
                    if (K == 1) {
                        any_two_of_units<LIVE, U>(unit, pre[g][0], pre[g][1]);
                    } else {
                        uint32_t s[K + 2];
#pragma unroll
                        for (int l = 0; l < K + 2; l++) s[l] = 0;
                        count_units<K, LIVE, U>(unit, s);
#pragma unroll
                        for (int l = 0; l < K + 2; l++) pre[g][l] = s[l];
                    }
                }
            };
            // hit masks of one column from its live units (cm[0 .. 4 LQ)) and the prefix state:
            // h[g] = rows of group g within the filter's reach
            auto eval_column = [&](uint32_t c, const uint32_t (&cm)[NP], uint32_t (&h)[G]) -> uint32_t {
                uint32_t anyhit = 0;
#pragma unroll
                for (int g = 0; g < G; g++) {
                    auto unit = [&](int u) { return unit_mask(cm, g, u); };
                    uint32_t hg; // rows with at most K mismatching units
                    if (K == 1) {
                        uint32_t any, two;
                        if (PU == 0) {
                            any_two_of_units<0, U>(unit, any, two);
                            hg = ~two & valid[g];
                        } else if (LIVE == 2) { // two in all = twoP | maj(anyP, m0, m1)
                            const uint32_t t = BITOP3(pre[g][0], unit(0), unit(1),
                                                      (TT_A & TT_B) | (TT_A & TT_C) | (TT_B & TT_C));
                            hg = BITOP3(pre[g][1], t, valid[g], ~(TT_A | TT_B) & TT_C);
                        } else {
                            any_two_of_units<0, LIVE>(unit, any, two);
                            const uint32_t t = BITOP3(two, pre[g][0], any, TT_A | (TT_B & TT_C));
                            hg = BITOP3(pre[g][1], t, valid[g], ~(TT_A | TT_B) & TT_C);
                        }
                    } else {
                        uint32_t s[K + 2]; // s[l] = rows with at least l mismatching units (l = 1..K+1)
#pragma unroll
                        for (int l = 0; l < K + 2; l++) s[l] = PU > 0 ? pre[g][l] : 0u;
                        count_units<K, 0, (PU > 0 ? LIVE : U)>(unit, s);
                        hg = ~s[K + 1] & valid[g];
                    }
                    if (DIAG) { // only rows before the column: keeps the self pair and i > j out
                        const int d = (int)(c0 + c - bucket_start) - (int)rbase[g];
                        const uint32_t lt = d <= 0 ? 0u : (d >= 32 ? 0xFFFFFFFFu : ((1u << d) - 1u));
                        hg &= lt;
                    }
                    h[g] = hg;
                    anyhit |= hg;
                }
                return anyhit;
            };
            auto queue_hits = [&](uint32_t c, const uint32_t (&h)[G]) {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    uint32_t hh = h[g];
                    while (hh) {
                        const int j = __builtin_ctz(hh);
                        hh &= hh - 1;
                        const uint32_t row = bucket_start + rbase[g] + j;
                        const unsigned int slot = atomicAdd(&hitq_count, 1u);
                        if (slot < HITQ) {
                            hitq[slot] = make_uint2(row, c0 + c);
                        } else { // queue full (a very dense tile): to the global overflow list,
                                 // which verify_list_kernel works off after this launch
                            const unsigned long long pos = atomicAdd(&a.counters[CNT_OVF], 1ull);
                            if (pos < a.ovf_cap) a.ovf[pos] = make_uint2(row, c0 + c);
                        }
                    }
                }
            };

            constexpr uint32_t CBLK = PU > 0 ? 32u : (uint32_t)BS_COL_TILE; // columns per flag word
            for (uint32_t cb = 0; cb < nc; cb += CBLK) {
                const uint32_t runs = PU > 0 ? __builtin_amdgcn_readfirstlane(runbits[cb >> 5]) : 0u;
                const uint32_t ce = min(nc, cb + CBLK);
                uint32_t c = cb + (COLSPLIT ? (uint32_t)(tid >> 6) : 0u);
                for (; c + (NCOL - 1) * CSTEP < ce; c += NCOL * CSTEP) { // whole groups
                    uint32_t h[NCOL][G];
                    uint32_t cm[NCOL][NP];
                    uint32_t anyhit = 0;
                    const uint32_t starts = PU > 0 ? (runs >> (c & 31)) & ((1u << NCOL) - 1u) : 0u;
                    if (PRELOAD && starts == 0) { // the common case: straight-line code
#pragma unroll
                        for (int i = 0; i < NCOL; i++) load_quads(c + i * CSTEP, 0, LQ, cm[i]);
#pragma unroll
                        for (int i = 0; i < NCOL; i++) anyhit |= eval_column(c + i * CSTEP, cm[i], h[i]);
                    } else {
#pragma unroll
                        for (int i = 0; i < NCOL; i++) {
                            if (PU > 0 && ((starts >> i) & 1u)) update_prefix(c + i * CSTEP);
                            load_quads(c + i * CSTEP, 0, LQ, cm[i]);
                            anyhit |= eval_column(c + i * CSTEP, cm[i], h[i]);
                        }
                    }
                    if (__any(anyhit != 0)) {
#pragma unroll
                        for (int i = 0; i < NCOL; i++) queue_hits(c + i * CSTEP, h[i]);
                    }
                }
                for (; c < ce; c += CSTEP) { // the columns that do not fill a group
                    uint32_t h[G];
                    uint32_t cm[NP];
                    if (PU > 0 && ((runs >> (c & 31)) & 1u)) update_prefix(c);
                    load_quads(c, 0, LQ, cm);
                    if (__any(eval_column(c, cm, h) != 0)) queue_hits(c, h);
                }
            }
        };
        if (diag) walk_columns(std::true_type{});
        else walk_columns(std::false_type{});

        // exact check of the tile's filter hits, one per thread
        // The queued hits are checked when the queue is half full or the task ends, not after
        // every tile: the check is a chain of dependent global gathers (permutation, keys, freq,
        // thresholds) that all four waves of the block would sit out 16 times per task.
        __syncthreads();
        const uint32_t queued = hitq_count; // the same for every thread: stable between the barriers
        if (queued >= HITQ / 2 || c0 + BS_COL_TILE >= col1) {
            const uint32_t nq = min(queued, HITQ);
            for (uint32_t i = tid; i < nq; i += THREADS) {
                const uint2 h = hitq[i];
                if (filter_key_distance(fkey[h.x], fkey[h.y]) > a.k) continue; // two bases of one unit
                verify_pair(a.keys, a.nmask, a.freq, a.thr, a.edges, a.edge_dist, a.counters, &stage,
                            a.edge_cap, a.k, a.mode, a.adj_max_freq, bucket_end, col1, h.x, h.y, a.perm);
            }
            __syncthreads();
            if (tid == 0) hitq_count = 0;
            flush_edges<THREADS>(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, with_dist, true);
        }
    }
    __syncthreads();
    if (tid == 0 && stage.candidates)
        atomicAdd(&a.counters[CNT_CANDIDATES], (unsigned long long)stage.candidates);
}

// ---- key-sorted buckets with 32-bit keys: scan + item walk -----------------------------------
// A bucket sorted by filter key is cut into row tiles (one wave: 64 lanes x 32 rows x BS_TAB_G) and
// 256-column tiles; tab_scan_kernel lists the (row tile, column tile) items that can hold a pair
// within k at all, and one of two kernels walks them:
//  * bs_run_kernel (default): per run of columns with equal high bases, the lanes that still have
//    an open row are handled one by one with the run's columns spread over the lanes;
//  * bs_tab_kernel (bs_transposed = 0): every column against all rows.  The mask "rows whose
//    2-base unit u differs from value v" depends on the rows only, so for the LIVE lowest units a
//    lane keeps all 16 of them in registers (one table of 16 words per unit, built once per row
//    tile: two bitop3 per entry).  A column selects its LIVE entries with its own unit values as
//    the index (wave-uniform, s_set_gpr_idx windows) and merges them with the cached state of the
//    PU = U - LIVE high units.  The common column costs LIVE register-indexed operands and two
//    bitop3 and touches no LDS: the mask formulation above is bound by the LDS return path (1 KB
//    per 128-bit broadcast read and wave), not by VALU.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// rows whose unit (planes q.x .. q.w) differs from the 4-bit value V:
// (P0 ^ b0) | (P1 ^ b1) | (P2 ^ b2) in one op, then | (P3 ^ b3); b = bits of V
template <int V> __device__ __forceinline__ uint32_t unit_differs_from(const uint4 &q)
{
    constexpr unsigned t3 = ((V & 1) ? (~TT_A & 0xFFu) : TT_A) | ((V & 2) ? (~TT_B & 0xFFu) : TT_B) |
                            ((V & 4) ? (~TT_C & 0xFFu) : TT_C);
    constexpr unsigned t2 = TT_A | ((V & 8) ? (~TT_B & 0xFFu) : TT_B);
    const uint32_t x = BITOP3(q.x, q.y, q.z, t3);
    return BITOP3(x, q.w, q.w, t2);
}
template <int... V>
__device__ __forceinline__ u32x16 unit_table(const uint4 &q, std::integer_sequence<int, V...>)
{
    u32x16 t;
    ((t[V] = unit_differs_from<V>(q)), ...);
    return t;
}

// rows of a group (prefix planes pp, plane j = bit j of a column's prefix bits) whose prefix unit
// u differs from that unit of the wave-uniform prefix bits pk: the column's bits become 0 / ~0
// masks on the scalar side
template <int PW>
__device__ __forceinline__ uint32_t tab_prefix_unit(const uint32_t (&pp)[PW], int u, uint32_t pk)
{
    uint32_t m = pp[4 * u] ^ (0u - ((pk >> (4 * u)) & 1u));
#pragma unroll
    for (int b = 1; b < 4; b++)
        m = BITOP3(m, pp[4 * u + b], 0u - ((pk >> (4 * u + b)) & 1u), TT_A | (TT_B ^ TT_C));
    return m;
}

// ---- which (row tile, column tile) pairs of a key-sorted bucket the table kernel must walk ----
// The columns of a 256-column tile share their highest bases (the bucket is sorted by key).  A tile
// whose first and last column agree in the NA highest units (or are a few values of them apart:
// every value in between is tried) is walked only if those units leave some row of the row tile
// within k: mismatches are sticky, so no column of the tile can hit otherwise.  Every pair is
// still decided by its own bits -- the high ones, which suffice.
// One block per row tile.  Its waves first fill a bitmap over all 16^NA values of the NA highest
// units -- "some row of the tile is within k of this prefix": the rows' planes against the value's
// bits, the two highest units alone first (they rule out most values) -- then every thread looks
// its column tiles up in it, and the tiles to walk are appended to the item list (one
// reservation per block and round, so a row tile's items sit together; the tiles on the bucket's
// diagonal go to a list of their own, from the array's end down).
constexpr int TAB_SCAN_THREADS = 1024;
template <int LP, int K>
__global__ __launch_bounds__(TAB_SCAN_THREADS) void tab_scan_kernel(PairArgs a, const TabRowTile *__restrict__ rts,
                                                                    TabItem *__restrict__ items, uint32_t item_cap,
                                                                    uint32_t part, uint32_t n_parts)
{
    constexpr int LIVE = 2, G = BS_TAB_G;
    constexpr int NP = 2 * LP, U = LP / 2, PU = U - LIVE, TILE = BS_TAB_TILE;
    constexpr int NA = PU - 1 < 3 ? PU - 1 : 3; // units of the bitmap (the lowest prefix unit is never one)
    constexpr bool CAN_SKIP = NA > K;           // enough units to pass k mismatches
    constexpr bool TWO_LEVEL = NA == 3 && 2 > K;
    constexpr int NVAL = CAN_SKIP ? 1 << (4 * NA) : 32;
    constexpr int WAVES = TAB_SCAN_THREADS / 64;
    constexpr uint32_t SEGS = 512; // words of 32 column tiles per reservation round
    __shared__ uint32_t alive_bits[NVAL >= 32 ? NVAL / 32 : 1];
    __shared__ uint32_t segmask[SEGS];
    __shared__ uint32_t segoff[SEGS];
    __shared__ uint32_t round_total, round_base;
    const TabRowTile *__restrict__ rt = rts + blockIdx.x;
    const uint32_t bucket_start = __builtin_amdgcn_readfirstlane(rt->bucket_start);
    const uint32_t bucket_end = __builtin_amdgcn_readfirstlane(rt->bucket_end);
    const uint32_t group0 = __builtin_amdgcn_readfirstlane(rt->group0);
    const uint32_t ngroups = __builtin_amdgcn_readfirstlane(rt->ngroups);
    const uint32_t *__restrict__ fkey = (const uint32_t *)a.fkey;
    const uint32_t *__restrict__ planes = a.planes + rt->plane_off;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n_rows = bucket_end - bucket_start;
    const uint32_t r_lo = bucket_start + group0 * 32;               // columns start at the tile's first row
    const uint32_t r_hi = min(bucket_end, r_lo + 64u * G * 32u);
    const uint32_t n_tiles = (bucket_end - r_lo + TILE - 1) / TILE;
    const uint32_t n_segs = (n_tiles + 31) / 32;

    if (CAN_SKIP) {
        for (uint32_t w = threadIdx.x; w < (uint32_t)(NVAL + 31) / 32; w += TAB_SCAN_THREADS) alive_bits[w] = 0;
        __syncthreads();
        // planes of the NA highest units of the tile's rows (every wave holds all of them)
        uint32_t valid[G], pa[G][4 * NA];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint32_t grp = group0 + (uint32_t)g * 64 + lane;
            const uint32_t rb = grp * 32;
            valid[g] = rb >= n_rows ? 0u : (n_rows - rb >= 32 ? 0xFFFFFFFFu : ((1u << (n_rows - rb)) - 1u));
#pragma unroll
            for (int q = 0; q < NA; q++) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (grp < ngroups) v = *reinterpret_cast<const uint4 *>(planes + (uint64_t)grp * NP + 4 * (U - NA + q));
                pa[g][4 * q] = v.x;
                pa[g][4 * q + 1] = v.y;
                pa[g][4 * q + 2] = v.z;
                pa[g][4 * q + 3] = v.w;
            }
        }
        // do the units FROM .. NA-1 of the value pv (unit q = bits 4q .. 4q+3) leave a row within k?
        auto open_after = [&](auto from_tag, uint32_t pv) {
            constexpr int FROM = decltype(from_tag)::value;
            uint32_t open_rows = 0;
#pragma unroll
            for (int g = 0; g < G; g++) {
                auto unit = [&](int u) { return tab_prefix_unit<4 * NA>(pa[g], u, pv); };
                if (K == 1) {
                    uint32_t any = 0, two = 0;
                    any_two_of_units<FROM, NA>(unit, any, two);
                    open_rows |= ~two & valid[g];
                } else {
                    uint32_t sc[K + 2];
#pragma unroll
                    for (int l = 0; l < K + 2; l++) sc[l] = 0;
                    count_units<K, FROM, NA>(unit, sc);
                    open_rows |= ~sc[K + 1] & valid[g];
                }
            }
            return __any(open_rows != 0);
        };
        constexpr uint32_t NQ = (uint32_t)NVAL / 16; // values of all units but the lowest of the NA
        for (uint32_t q = wave; q < NQ; q += WAVES) {
            if (TWO_LEVEL && !open_after(std::integral_constant<int, (TWO_LEVEL ? 1 : 0)>{}, q << 4)) continue;
            uint32_t bits = 0;
            for (uint32_t v = 0; v < 16; v++)
                bits |= (open_after(std::integral_constant<int, 0>{}, (q << 4) | v) ? 1u : 0u) << v;
            if (lane == 0 && bits) atomicOr(&alive_bits[q >> 1], bits << ((q & 1u) * 16));
        }
        __syncthreads();
    }

    for (uint32_t seg_base = 0; seg_base < n_segs; seg_base += SEGS) {
        const uint32_t segs_here = min(SEGS, n_segs - seg_base);
        for (uint32_t i0 = wave * 64; i0 < segs_here * 32; i0 += TAB_SCAN_THREADS) {
            const uint32_t tile = seg_base * 32 + i0 + lane;
            bool walk = tile < n_tiles;
            if (walk && CAN_SKIP) {
                const uint32_t c = r_lo + tile * TILE;
                const uint32_t pf = fkey[c] >> (4 * (U - NA));
                const uint32_t pl = fkey[min(c + TILE, bucket_end) - 1u] >> (4 * (U - NA));
                if (pl - pf <= 3u) { // (sorted: pl >= pf; more values in between than that: walk)
                    walk = false;
                    for (uint32_t p = pf; p <= pl; p++) walk = walk || ((alive_bits[p >> 5] >> (p & 31)) & 1u) != 0;
                }
            }
            if (n_parts > 1) // a split call: this rank's share of the tiles (the same on every rank)
                walk = walk && (blockIdx.x + tile) % n_parts == part;
            // Dense tiles -- on the bucket's diagonal, or with the highest units of some row of the
            // tile (sorted neighbours: many rows stay open there, and nearly all filter hits fall
            // there) -- take several times as long as the others: a list of their own, from the
            // array's end down, worked off first.
            bool dense = false;
            if (walk) {
                const uint32_t c = r_lo + tile * TILE;
                dense = c < r_hi;
                if (!dense && CAN_SKIP && NA >= 2) {
                    const uint32_t qf = fkey[c] >> (4 * (U - NA + 1));
                    dense = qf >= (fkey[r_lo] >> (4 * (U - NA + 1))) && qf <= (fkey[r_hi - 1u] >> (4 * (U - NA + 1)));
                }
            }
            const unsigned long long dbal = __ballot(dense);
            if (dbal) {
                uint32_t dbase = 0;
                if (lane == 0)
                    dbase = (uint32_t)min(atomicAdd(&a.counters[CNT_DIAG_ITEMS], (unsigned long long)__builtin_popcountll(dbal)),
                                          (unsigned long long)0xFFFFFFFFu);
                dbase = (uint32_t)__builtin_amdgcn_readfirstlane(dbase);
                if (dense) {
                    const uint64_t pos = (uint64_t)dbase + (uint32_t)__builtin_popcountll(dbal & ((1ull << lane) - 1ull));
                    const uint32_t c0 = r_lo + tile * TILE;
                    if (pos < item_cap)
                        items[item_cap - 1 - (uint32_t)pos] =
                            TabItem{blockIdx.x, c0, min((uint32_t)TILE, bucket_end - c0), c0 < r_hi ? 1u : 0u};
                }
                walk = walk && !dense;
            }
            const unsigned long long bal = __ballot(walk);
            if (lane == 0) {
                segmask[i0 >> 5] = (uint32_t)bal;
                if ((i0 >> 5) + 1 < SEGS) segmask[(i0 >> 5) + 1] = (uint32_t)(bal >> 32);
            }
        }
        __syncthreads();
        if (wave == 0) { // exclusive prefix sum of the words' item counts
            uint32_t cnt[SEGS / 64], sum = 0;
#pragma unroll
            for (uint32_t q = 0; q < SEGS / 64; q++) {
                const uint32_t sl = lane * (SEGS / 64) + q;
                cnt[q] = sl < segs_here ? (uint32_t)__builtin_popcount(segmask[sl]) : 0u;
                sum += cnt[q];
            }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d);
                if ((int)lane >= d) incl += up;
            }
            uint32_t off = incl - sum;
#pragma unroll
            for (uint32_t q = 0; q < SEGS / 64; q++) {
                segoff[lane * (SEGS / 64) + q] = off;
                off += cnt[q];
            }
            if (lane == 63) round_total = incl;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            round_base = (uint32_t)min(atomicAdd(&a.counters[CNT_ITEMS], (unsigned long long)round_total),
                                       (unsigned long long)0xFFFFFFFFu);
        __syncthreads();
        const uint32_t base = round_base;
        for (uint32_t idx = threadIdx.x; idx < segs_here * 32; idx += TAB_SCAN_THREADS) {
            const uint32_t sl = idx >> 5, t = idx & 31;
            const uint32_t m = segmask[sl];
            if ((m >> t) & 1u) {
                const uint64_t pos = (uint64_t)base + segoff[sl] + (uint32_t)__builtin_popcount(m & ((1u << t) - 1u));
                const uint32_t c0 = r_lo + ((seg_base + sl) * 32 + t) * TILE;
                if (pos < item_cap)
                    items[pos] = TabItem{blockIdx.x, c0, min((uint32_t)TILE, bucket_end - c0), c0 < r_hi ? 1u : 0u};
            }
        }
        __syncthreads();
    }
}

// ---- the table kernel proper: persistent waves over the item list ----------------------------
// One wave per block (no wave waits for another one's columns), as many blocks as the chip holds;
// the waves take items off the two lists until both are used up (see the loops at the end).
template <int LP, int K, int LIVE, int G>
__global__ __launch_bounds__(64, G == 1 ? 4 : 3) void bs_tab_kernel(PairArgs a, const TabRowTile *__restrict__ rts,
                                                       const TabItem *__restrict__ items, uint32_t item_cap)
{
    constexpr int THREADS = 64;
    constexpr int NP = 2 * LP;
    constexpr int U = LP / 2;      // 2-base units per key
    constexpr int PU = U - LIVE;   // prefix units: state cached per column run
    constexpr int PW = 4 * PU;     // prefix planes per row group
    constexpr int TILE = BS_TAB_TILE; // columns per item
    constexpr uint32_t BATCH = 4;
    static_assert(LIVE >= 1 && LIVE <= 4 && PU >= 1, "table variant: 1..4 live units and a prefix");
    static_assert(LIVE == 2 && (G == 1 || G == 2), "only the tuned shapes are instantiated (launch_bs_tab)");
    static_assert(4 * PU <= 32, "prefix bits of a column in one word");
    __shared__ uint32_t runbits[TILE / 32];
    __shared__ uint32_t ckey[TILE + 8]; // live unit values of the item's columns (+ the group read ahead past its end)
    __shared__ uint32_t pkey[TILE];     // their prefix bits (unit q of the prefix = bits 4q .. 4q+3)
    __shared__ uint32_t nxt[TILE];      // first column after c that starts a run (or the item's end)
    constexpr uint32_t HITQ = 256;
    __shared__ uint2 hitq[HITQ];
    __shared__ unsigned int hitq_count;
    __shared__ EdgeStage stage;
    const uint32_t *__restrict__ fkey = (const uint32_t *)a.fkey;
    const int tid = threadIdx.x;
    const bool with_dist = a.mode == MODE_NEIGHBOURS;
    constexpr bool EARLY = true;
    const uint32_t n_items = (uint32_t)min((unsigned long long)item_cap, a.counters[CNT_ITEMS]);
    const uint32_t n_diag = (uint32_t)min((unsigned long long)item_cap, a.counters[CNT_DIAG_ITEMS]);
    // the next `count` items of a list (wave-uniform; >= the list's length when it is used up)
    auto grab = [&](int counter, uint32_t count) {
        uint32_t got = 0;
        if (tid == 0)
            got = (uint32_t)min((unsigned long long)0xFFFFFFFFu,
                                atomicAdd(&a.counters[counter], (unsigned long long)count));
        return (uint32_t)__builtin_amdgcn_readfirstlane(got); // lane 0's
    };

    if (tid == 0) {
        stage.count = 0;
        stage.candidates = 0;
        hitq_count = 0;
    }
    __syncthreads();

    // state of the row tile in hand
    uint32_t cur_row_tile = 0xFFFFFFFFu;
    uint32_t bucket_start = 0, bucket_end = 0;
    uint32_t rbase[G] = {}, valid[G] = {};
    uint32_t pp[G][PW] = {};     // planes of the prefix units (plane j of the prefix = bit j of a column's pkey)
    // tXY[v] = rows of group X whose live unit Y differs from value v.  Four named vectors, not
    // an array: an array of vectors this large stays in scratch memory instead of registers.
    u32x16 t00 = {}, t01 = {}, t10 = {}, t11 = {};
    uint32_t pre[G][K + 2] = {}; // counter state after the prefix units ((any, two) for K = 1)
    bool dead = false;           // ... puts every row of the wave beyond k (wave-uniform)
    uint32_t c0 = 0, nc = 0;

    auto prefix_unit = [&](int g, int u, uint32_t pk) { return tab_prefix_unit<PW>(pp[g], u, pk); };

    // Exact check of the queued filter hits, one per lane: a base-level test on the two filter
    // keys first (the unit-level filter lets pairs through that differ in two bases of one
    // unit), then the reference arithmetic.  The block is one wave, so the queue can be worked
    // off wherever the wave stands -- when it is half full, or the wave is done.
    auto drain = [&](bool final) {
        __syncthreads();
        const uint32_t nq = min(hitq_count, HITQ);
        for (uint32_t i = tid; i < nq; i += THREADS) {
            const uint2 h = hitq[i];
            if (filter_key_distance(fkey[h.x], fkey[h.y]) > a.k) continue;
            verify_pair(a.keys, a.nmask, a.freq, a.thr, a.edges, a.edge_dist, a.counters, &stage,
                        a.edge_cap, a.k, a.mode, a.adj_max_freq, 0xFFFFFFFFu, 0xFFFFFFFFu, h.x, h.y, a.perm);
        }
        __syncthreads();
        if (tid == 0) hitq_count = 0;
        flush_edges<THREADS>(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, with_dist, final);
    };

    auto walk_columns = [&](auto diag_tag) {
        constexpr bool DIAG = decltype(diag_tag)::value;
        constexpr int NCOL = 4; // columns per "any hit?" test
        auto update_prefix = [&](uint32_t c) {
            const uint32_t pk = __builtin_amdgcn_readfirstlane(pkey[c]);
#pragma unroll
            for (int g = 0; g < G; g++) {
                auto unit = [&](int u) { return prefix_unit(g, u, pk); }; // u = 0 .. PU-1
                if (K == 1) {
                    any_two_of_units<0, PU>(unit, pre[g][0], pre[g][1]);
                } else {
                    uint32_t s[K + 2];
#pragma unroll
                    for (int l = 0; l < K + 2; l++) s[l] = 0;
                    count_units<K, 0, PU>(unit, s);
#pragma unroll
                    for (int l = 0; l < K + 2; l++) pre[g][l] = s[l];
                }
            }
        };
        // rows within the filter's reach of column c, whose (sorted) key is `key`
        auto eval_column = [&](uint32_t c, uint32_t key, uint32_t (&h)[G]) -> uint32_t {
            uint32_t anyhit = 0;
            uint32_t e2[2] = {0u, 0u}, f2[2] = {0u, 0u}, maj2[2] = {0u, 0u};
            if (LIVE == 2) {
                // The four lookups of a column under one index-mode window: the compiler
                // brackets every indexed move with its own s_set_gpr_idx_on/off, and the one
                // scalar unit of a CU (one instruction per 4 cycles and SIMD) is what bounds
                // this loop.  The tables are pinned to v[64:127] for the statement.  For
                // K = 1 the second pair of lookups is the indexed source of the majority op
                // itself (two moves fewer per column).
                // (m0 is rewritten by the window; the compiler never keeps a value in m0
                // across statements, and lists it as reserved, so it is not a clobber here)
                const uint32_t ukey = __builtin_amdgcn_readfirstlane(key); // (already uniform)
                const uint32_t i0 = ukey, i1 = ukey >> 8; // ckey[] holds them one per byte
                if (G == 1 && K == 1) {
                    asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC0)\n\t"
                                 "v_mov_b32 %0, v64\n\t"
                                 "s_set_gpr_idx_idx %3\n\t"
                                 "v_bitop3_b32 %1, v80, %4, %0 bitop3:0xe8\n\t"
                                 "s_set_gpr_idx_off"
                                 : "=&v"(e2[0]), "=&v"(maj2[0])
                                 : "s"(i0), "s"(i1), "v"(pre[0][0]), "{v[64:79]}"(t00), "{v[80:95]}"(t01));
                } else if (G == 1) {
                    asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC0)\n\t"
                                 "v_mov_b32 %0, v64\n\t"
                                 "s_set_gpr_idx_idx %3\n\t"
                                 "v_mov_b32 %1, v80\n\t"
                                 "s_set_gpr_idx_off"
                                 : "=&v"(e2[0]), "=&v"(f2[0])
                                 : "s"(i0), "s"(i1), "{v[64:79]}"(t00), "{v[80:95]}"(t01));
                } else if (K == 1) {
                    asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\t"
                                 "v_mov_b32 %0, v64\n\t"
                                 "v_mov_b32 %1, v96\n\t"
                                 "s_set_gpr_idx_idx %5\n\t"
                                 "v_bitop3_b32 %2, v80, %6, %0 bitop3:0xe8\n\t"
                                 "v_bitop3_b32 %3, v112, %7, %1 bitop3:0xe8\n\t"
                                 "s_set_gpr_idx_off"
                                 : "=&v"(e2[0]), "=&v"(e2[1]), "=&v"(maj2[0]), "=&v"(maj2[1])
                                 : "s"(i0), "s"(i1), "v"(pre[0][0]), "v"(pre[G - 1][0]), "{v[64:79]}"(t00),
                                   "{v[80:95]}"(t01), "{v[96:111]}"(t10), "{v[112:127]}"(t11));
                } else {
                    asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\t"
                                 "v_mov_b32 %0, v64\n\t"
                                 "v_mov_b32 %1, v96\n\t"
                                 "s_set_gpr_idx_idx %5\n\t"
                                 "v_mov_b32 %2, v80\n\t"
                                 "v_mov_b32 %3, v112\n\t"
                                 "s_set_gpr_idx_off"
                                 : "=&v"(e2[0]), "=&v"(e2[1]), "=&v"(f2[0]), "=&v"(f2[1])
                                 : "s"(i0), "s"(i1), "{v[64:79]}"(t00), "{v[80:95]}"(t01), "{v[96:111]}"(t10),
                                   "{v[112:127]}"(t11));
                }
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                auto unit = [&](int u) {
                    return u == 0 ? e2[g] : f2[g];
                };
                uint32_t hg;
                if (K == 1) {
                    if (LIVE == 1) {
                        hg = BITOP3(pre[g][1], pre[g][0] & unit(0), valid[g], ~(TT_A | TT_B) & TT_C);
                    } else if (LIVE == 2) { // two in all = twoP | maj(anyP, m0, m1)
                        const uint32_t t = maj2[g]; // maj(anyP, m0, m1), from the lookup window
                        hg = BITOP3(pre[g][1], t, valid[g], ~(TT_A | TT_B) & TT_C);
                    } else {
                        uint32_t any, two;
                        any_two_of_units<0, LIVE>(unit, any, two);
                        const uint32_t t = BITOP3(two, pre[g][0], any, TT_A | (TT_B & TT_C));
                        hg = BITOP3(pre[g][1], t, valid[g], ~(TT_A | TT_B) & TT_C);
                    }
                } else {
                    uint32_t s[K + 2];
#pragma unroll
                    for (int l = 0; l < K + 2; l++) s[l] = pre[g][l];
                    count_units<K, 0, LIVE>(unit, s);
                    hg = ~s[K + 1] & valid[g];
                }
                if (DIAG) { // only rows before the column: keeps the self pair and i > j out
                    const int d = (int)(c0 + c - bucket_start) - (int)rbase[g];
                    const uint32_t lt = d <= 0 ? 0u : (d >= 32 ? 0xFFFFFFFFu : ((1u << d) - 1u));
                    hg &= lt;
                }
                h[g] = hg;
                anyhit |= hg;
            }
            return anyhit;
        };
        auto queue_hits = [&](uint32_t c, const uint32_t (&h)[G]) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                uint32_t hh = h[g];
                while (hh) {
                    const int j = __builtin_ctz(hh);
                    hh &= hh - 1;
                    const uint32_t row = bucket_start + rbase[g] + j;
                    const unsigned int slot = atomicAdd(&hitq_count, 1u);
                    if (slot < HITQ) {
                        hitq[slot] = make_uint2(row, c0 + c);
                    } else { // queue full (a very dense tile): to the global overflow list,
                             // which verify_list_kernel works off after this launch
                        const unsigned long long pos = atomicAdd(&a.counters[CNT_OVF], 1ull);
                        if (pos < a.ovf_cap) a.ovf[pos] = make_uint2(row, c0 + c);
                    }
                }
            }
        };
        // run by run: new prefix state at the run's first column, then its columns in groups
        // of NCOL (one hit test per group); the last group of a run is evaluated whole and
        // the hit words of the columns past the run's end are dropped (their prefix state is
        // not theirs).  The bookkeeping is wave-uniform: scalar instructions and branches.
        // Column keys come from the LDS copy one group ahead of their use (wave-wide reads of
        // one address, then v_readfirstlane: LDS returns in order, so the wait for a group's
        // keys does not drain the reads issued after them; scalar loads from the key array
        // itself would, and they miss the scalar cache every 16 columns).
        uint32_t next[NCOL];
        bool starts = true; // an item's first column begins a run
        uint32_t c = 0;
        while (c < nc) {
            c = __builtin_amdgcn_readfirstlane(c); // (uniform already; keeps it in an SGPR)
            if (starts) {
                update_prefix(c);
                // early out: when the high units alone put every row of the wave beyond k, the
                // run's columns cannot hit (the state is sticky) -- skip their low units
                uint32_t open_rows = 0;
#pragma unroll
                for (int g = 0; g < G; g++) open_rows |= ~pre[g][K == 1 ? 1 : K + 1] & valid[g];
                dead = EARLY && !__any(open_rows != 0);
            }
            starts = true; // every later run of the tile begins at a flagged column
            const uint32_t e = __builtin_amdgcn_readfirstlane(nxt[c]);
            if (dead) {
                c = e;
                continue;
            }
#pragma unroll
            for (int i = 0; i < NCOL; i++) next[i] = ckey[c + i];
            for (; c + NCOL <= e; c += NCOL) { // whole groups: straight-line code
                uint32_t key[NCOL], h[NCOL][G];
#pragma unroll
                for (int i = 0; i < NCOL; i++) key[i] = __builtin_amdgcn_readfirstlane(next[i]);
#pragma unroll
                for (int i = 0; i < NCOL; i++) next[i] = ckey[c + NCOL + i];
                uint32_t anyhit = 0;
#pragma unroll
                for (int i = 0; i < NCOL; i++) anyhit |= eval_column(c + i, key[i], h[i]);
                if (__any(anyhit != 0)) {
#pragma unroll
                    for (int i = 0; i < NCOL; i++) queue_hits(c + i, h[i]);
                    if ((uint32_t)__builtin_amdgcn_readfirstlane(*(volatile unsigned int *)&hitq_count) >= HITQ / 2)
                        drain(false);
                }
            }
            if (c < e) { // the run's last 1..NCOL-1 columns: one more hit test for them
                uint32_t h[NCOL][G];
                const uint32_t cnt = e - c;
                uint32_t anyhit = 0;
#pragma unroll
                for (int i = 0; i < NCOL - 1; i++) {
                    if ((uint32_t)i < cnt) { // (wave-uniform)
                        anyhit |= eval_column(c + i, __builtin_amdgcn_readfirstlane(next[i]), h[i]);
                    } else {
#pragma unroll
                        for (int g = 0; g < G; g++) h[i][g] = 0u;
                    }
                }
                if (__any(anyhit != 0)) {
#pragma unroll
                    for (int i = 0; i < NCOL - 1; i++) queue_hits(c + i, h[i]);
                    if ((uint32_t)__builtin_amdgcn_readfirstlane(*(volatile unsigned int *)&hitq_count) >= HITQ / 2)
                        drain(false);
                }
                c = e;
            }
        }
    };
    auto process = [&](const TabItem &item) {
        const uint32_t row_tile = __builtin_amdgcn_readfirstlane(item.row_tile);
        c0 = __builtin_amdgcn_readfirstlane(item.col0);
        nc = __builtin_amdgcn_readfirstlane(item.ncols);
        const bool diag = __builtin_amdgcn_readfirstlane(item.diag) != 0;

        if (row_tile != cur_row_tile) { // another row tile: its planes and tables
            cur_row_tile = row_tile;
            const TabRowTile *__restrict__ rt = rts + row_tile;
            bucket_start = __builtin_amdgcn_readfirstlane(rt->bucket_start);
            bucket_end = __builtin_amdgcn_readfirstlane(rt->bucket_end);
            const uint32_t group0 = __builtin_amdgcn_readfirstlane(rt->group0);
            const uint32_t ngroups = __builtin_amdgcn_readfirstlane(rt->ngroups);
            const uint32_t *__restrict__ planes = a.planes + rt->plane_off;
            const uint32_t n_rows = bucket_end - bucket_start;
            uint32_t grp[G];
            auto load_quad = [&](int g, int q) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (grp[g] < ngroups) v = *reinterpret_cast<const uint4 *>(planes + (uint64_t)grp[g] * NP + 4 * q);
                return v;
            };
#pragma unroll
            for (int g = 0; g < G; g++) {
                grp[g] = group0 + (uint32_t)g * THREADS + (uint32_t)tid; // the wave's 64 * G groups are adjacent
                rbase[g] = grp[g] * 32;
                valid[g] = rbase[g] >= n_rows ? 0u
                           : (n_rows - rbase[g] >= 32 ? 0xFFFFFFFFu : ((1u << (n_rows - rbase[g])) - 1u));
#pragma unroll
                for (int q = 0; q < PU; q++) {
                    const uint4 v = load_quad(g, LIVE + q);
                    pp[g][4 * q] = v.x;
                    pp[g][4 * q + 1] = v.y;
                    pp[g][4 * q + 2] = v.z;
                    pp[g][4 * q + 3] = v.w;
                }
            }
            t00 = unit_table(load_quad(0, 0), std::make_integer_sequence<int, 16>{});
            t01 = unit_table(load_quad(0, 1), std::make_integer_sequence<int, 16>{});
            if (G == 2) {
                t10 = unit_table(load_quad(G - 1, 0), std::make_integer_sequence<int, 16>{});
                t11 = unit_table(load_quad(G - 1, 1), std::make_integer_sequence<int, 16>{});
            }
        }

        // stage the item's columns
        __syncthreads();
        for (uint32_t cc = tid; cc < (uint32_t)TILE + 8; cc += THREADS) { // (padded array)
            // the two live unit values of the column, one per byte: s_set_gpr_idx_on / _idx take
            // the low byte of their operand as the index, so no masking is left for the walk
            const uint32_t kq = fkey[min(c0 + cc, bucket_end - 1u)]; // (columns past the end: never used)
            ckey[cc] = (kq & 15u) | (((kq >> 4) & 15u) << 8);
            if (cc < (uint32_t)TILE) pkey[cc] = kq >> (4 * LIVE);
        }
        __syncthreads();
        for (uint32_t cc = tid; cc < (uint32_t)TILE; cc += THREADS) { // does column cc start a run of equal high bases?
            const bool newrun = cc < nc && (cc == 0 || pkey[cc] != pkey[cc - 1]);
            const unsigned long long bal = __ballot(newrun);
            if (tid == 0) {
                runbits[2 * (cc >> 6)] = (uint32_t)bal;
                runbits[2 * (cc >> 6) + 1] = (uint32_t)(bal >> 32);
            }
        }
        __syncthreads();
        for (uint32_t cc = tid; cc < nc; cc += THREADS) { // where does the run after column cc begin?
            uint32_t q = cc + 1, res = nc;
            for (uint32_t w = q >> 5; w < (uint32_t)TILE / 32; w++) {
                const uint32_t m = runbits[w] & (w == (q >> 5) ? 0xFFFFFFFFu << (q & 31) : 0xFFFFFFFFu);
                if (m) {
                    res = min(nc, w * 32 + (uint32_t)__builtin_ctz(m));
                    break;
                }
            }
            nxt[cc] = res;
        }
        __syncthreads();

        if (diag) walk_columns(std::true_type{});
        else walk_columns(std::false_type{});
    };

    // The items on a bucket's diagonal first, one at a time (sorted neighbours: nearly all filter
    // hits of the bucket fall there, an item takes several times as long as the others), then the
    // rest in batches of BATCH neighbours of the list, which mostly share their row tile.  Items
    // are handed out through a counter, the next grab is under way while an item is walked; the
    // loops end for every wave once the lists are used up.
    for (uint32_t cur = grab(CNT_DIAG_GRAB, 1); cur < n_diag;) {
        const uint32_t ahead = grab(CNT_DIAG_GRAB, 1);
        process(items[item_cap - 1 - cur]);
        cur = ahead;
    }
    for (uint32_t cur = grab(CNT_GRAB, BATCH); cur < n_items;) {
        const uint32_t ahead = grab(CNT_GRAB, BATCH);
        for (uint32_t m = 0; m < BATCH && cur + m < n_items; m++) process(items[cur + m]);
        cur = ahead;
    }
    drain(true);
    __syncthreads();
    if (tid == 0 && stage.candidates)
        atomicAdd(&a.counters[CNT_CANDIDATES], (unsigned long long)stage.candidates);
}

// ---- the item walk with the columns across the lanes ----------------------------------------
// Same items, same counters as bs_tab_kernel, other shape of the inner loop.  In a walked item the
// state after the prefix units (row-parallel: a lane = 32 rows, recomputed per run of columns with
// equal high bases) leaves only a few rows within k -- those whose high bases all but agree with the
// run's, one or two lanes' worth of the 2048 in the common item.  Walking every column of the run
// against all 64 lanes spends 63 of them on rows that are already decided.  So per run the wave
// takes the ballot of the lanes that still have an open row and, for each of them in turn, spreads
// the run's COLUMNS over the lanes (a run is ~15 columns; longer ones go in chunks of 64): the open
// lane's live-unit planes, counters and validity are broadcast with v_readlane (wave-uniform), each
// lane compares them with its own column's low bits, and 32 rows x up to 64 columns are decided by
// a dozen instructions.  No tables, no index-mode windows, no LDS in the loop.
template <int LP, int K>
__global__ __launch_bounds__(64, 5) void bs_run_kernel(PairArgs a, const TabRowTile *__restrict__ rts,
                                                       const TabItem *__restrict__ items, uint32_t item_cap)
{
    constexpr int THREADS = 64;
    constexpr int LIVE = 2, NP = 2 * LP, U = LP / 2, PU = U - LIVE, PW = 4 * PU, LW = 4 * LIVE;
    constexpr int TILE = BS_TAB_TILE;
    constexpr uint32_t BATCH = 4; // ordinary items per turn: neighbours of the list, mostly one row tile
    static_assert(BS_TAB_G == 1, "one 32-row group per lane");
    static_assert(PU >= 1 && 4 * PU <= 32, "prefix bits of a column in one word");
    __shared__ uint32_t runbits[TILE / 32];
    __shared__ uint32_t kcol[TILE];     // the item's column keys
    __shared__ uint32_t nxt[TILE];      // first column after c that starts a run (or the item's end)
    constexpr uint32_t HITQ = 128;
    __shared__ uint2 hitq[HITQ];
    __shared__ unsigned int hitq_count;
    __shared__ EdgeStage stage;
    const uint32_t *__restrict__ fkey = (const uint32_t *)a.fkey;
    const int tid = threadIdx.x;
    const bool with_dist = a.mode == MODE_NEIGHBOURS;
    const uint32_t n_items = (uint32_t)min((unsigned long long)item_cap, a.counters[CNT_ITEMS]);
    const uint32_t n_diag = (uint32_t)min((unsigned long long)item_cap, a.counters[CNT_DIAG_ITEMS]);
    if (tid == 0) {
        stage.count = 0;
        stage.candidates = 0;
        hitq_count = 0;
    }
    __syncthreads();

    uint32_t cur_row_tile = 0xFFFFFFFFu;
    uint32_t bucket_start = 0, bucket_end = 0, group0 = 0;
    uint32_t valid = 0;
    uint32_t pp[PW] = {}; // planes of the prefix units of the lane's 32 rows
    uint32_t lp[LW] = {}; // ... of the two live units
    uint32_t pre[K + 2] = {};
    uint32_t hi[K + 2] = {}; // state after the prefix units above the lowest, for the high bits hi_bits
    uint32_t hi_bits = 0xFFFFFFFFu;

    auto drain = [&](bool final) {
        __syncthreads();
        const uint32_t nq = min(hitq_count, HITQ);
        for (uint32_t i = tid; i < nq; i += THREADS) {
            const uint2 h = hitq[i];
            if (filter_key_distance(fkey[h.x], fkey[h.y]) > a.k) continue; // two bases of one unit
            verify_pair(a.keys, a.nmask, a.freq, a.thr, a.edges, a.edge_dist, a.counters, &stage,
                        a.edge_cap, a.k, a.mode, a.adj_max_freq, 0xFFFFFFFFu, 0xFFFFFFFFu, h.x, h.y, a.perm);
        }
        __syncthreads();
        if (tid == 0) hitq_count = 0;
        flush_edges<THREADS>(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, with_dist, final);
    };

    constexpr int KPL = TILE / THREADS; // column keys per lane and item
    // an item's column keys, one load per 64 columns (issued an item ahead of their use)
    auto load_keys = [&](const TabItem &item, uint32_t (&k)[KPL]) {
        const uint32_t col0 = __builtin_amdgcn_readfirstlane(item.col0);
#pragma unroll
        for (int j = 0; j < KPL; j++) k[j] = fkey[min(col0 + (uint32_t)tid + 64u * j, a.n_entries - 1u)];
    };
    uint32_t c0 = 0, nc = 0; // the item in hand
    bool diag = false;
    auto stage_item = [&](const TabItem &item, const uint32_t (&k)[KPL]) {
        const uint32_t row_tile = __builtin_amdgcn_readfirstlane(item.row_tile);
        c0 = __builtin_amdgcn_readfirstlane(item.col0);
        nc = __builtin_amdgcn_readfirstlane(item.ncols);
        diag = __builtin_amdgcn_readfirstlane(item.diag) != 0;
        if (row_tile != cur_row_tile) { // another row tile: its planes
            cur_row_tile = row_tile;
            hi_bits = 0xFFFFFFFFu;
            const TabRowTile *__restrict__ rt = rts + row_tile;
            bucket_start = __builtin_amdgcn_readfirstlane(rt->bucket_start);
            bucket_end = __builtin_amdgcn_readfirstlane(rt->bucket_end);
            group0 = __builtin_amdgcn_readfirstlane(rt->group0);
            const uint32_t ngroups = __builtin_amdgcn_readfirstlane(rt->ngroups);
            const uint32_t *__restrict__ planes = a.planes + rt->plane_off;
            const uint32_t n_rows = bucket_end - bucket_start;
            const uint32_t grp = group0 + (uint32_t)tid;
            const uint32_t rb = grp * 32;
            valid = rb >= n_rows ? 0u : (n_rows - rb >= 32 ? 0xFFFFFFFFu : ((1u << (n_rows - rb)) - 1u));
#pragma unroll
            for (int q = 0; q < U; q++) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (grp < ngroups) v = *reinterpret_cast<const uint4 *>(planes + (uint64_t)grp * NP + 4 * q);
                uint32_t *dst = q < LIVE ? &lp[4 * q] : &pp[4 * (q - LIVE)];
                dst[0] = v.x;
                dst[1] = v.y;
                dst[2] = v.z;
                dst[3] = v.w;
            }
        }

        // Stage the item's column keys, flag the run starts (a run: neighbours that agree in the
        // prefix units), link every column to its run's end.  The block is one wave: LDS accesses
        // of a wave complete in order, so the compiler-level wave barriers are all that is needed --
        // a __syncthreads() would also wait for the key loads of the next item.
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < KPL; j++) {
            const uint32_t cc = (uint32_t)tid + 64u * j;
            uint32_t prev = __shfl_up(k[j], 1);
            if (tid == 0) prev = j ? (uint32_t)__builtin_amdgcn_readlane(k[j ? j - 1 : 0], 63) : ~k[0];
            kcol[cc] = k[j];
            const bool newrun = cc < nc && (cc == 0 || ((k[j] ^ prev) >> (4 * LIVE)) != 0);
            const unsigned long long bal = __ballot(newrun);
            if (tid == 0) {
                runbits[2 * j] = (uint32_t)bal;
                runbits[2 * j + 1] = (uint32_t)(bal >> 32);
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t cc = tid; cc < nc; cc += THREADS) {
            uint32_t q = cc + 1, res = nc;
            for (uint32_t w = q >> 5; w < (uint32_t)TILE / 32; w++) {
                const uint32_t m = runbits[w] & (w == (q >> 5) ? 0xFFFFFFFFu << (q & 31) : 0xFFFFFFFFu);
                if (m) {
                    res = min(nc, w * 32 + (uint32_t)__builtin_ctz(m));
                    break;
                }
            }
            nxt[cc] = res;
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto walk_item = [&]() {
        for (uint32_t c = 0; c < nc;) { // run by run
            c = __builtin_amdgcn_readfirstlane(c);
            const uint32_t e = __builtin_amdgcn_readfirstlane(nxt[c]);
            const uint32_t pk = __builtin_amdgcn_readfirstlane(kcol[c]) >> (4 * LIVE);
            { // counter state of every row after the prefix units (lane = 32 rows).  Sorted keys:
              // from one run to the next mostly the lowest prefix unit alone changes, so the state
              // of the units above it is kept (hi[], for the bits hi_bits) and only that unit is redone
                auto unit = [&](int u) { return tab_prefix_unit<PW>(pp, u, pk); };
                if (PU >= 2 && (pk >> 4) != hi_bits) { // (wave-uniform)
                    hi_bits = pk >> 4;
                    if (K == 1) {
                        any_two_of_units<1, PU>(unit, hi[0], hi[1]);
                    } else {
#pragma unroll
                        for (int l = 0; l < K + 2; l++) hi[l] = 0;
                        count_units<K, 1, PU>(unit, hi);
                    }
                }
                const uint32_t m0 = unit(0);
                if (K == 1) {
                    pre[1] = PU >= 2 ? BITOP3(hi[1], hi[0], m0, TT_A | (TT_B & TT_C)) : 0u;
                    pre[0] = PU >= 2 ? (hi[0] | m0) : m0;
                } else {
#pragma unroll
                    for (int l = 0; l < K + 2; l++) pre[l] = PU >= 2 ? hi[l] : 0u;
                    count_units<K, 0, 1>([&](int) { return m0; }, pre);
                }
            }
            const uint32_t open_rows = ~pre[K == 1 ? 1 : K + 1] & valid;
            unsigned long long open_lanes = __ballot(open_rows != 0); // (wave-uniform)
            if (open_lanes) {
                for (uint32_t cs = c; cs < e; cs += THREADS) { // the run's columns across the lanes
                    const uint32_t col = cs + (uint32_t)tid;   // this lane's column (if < e)
                    const uint32_t kq = kcol[min(col, (uint32_t)TILE - 1u)];
                    uint32_t cm[LW]; // its live bits as 0 / ~0 masks
#pragma unroll
                    for (int b = 0; b < LW; b++) cm[b] = 0u - ((kq >> b) & 1u);
                    const uint32_t gcol = c0 + col - bucket_start; // bucket-relative column index
                    for (unsigned long long todo = open_lanes; todo;) {
                        const int la = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        uint32_t pl[LW]; // the open lane's live-unit planes, on the scalar side (all reads
#pragma unroll                           // first: each one's use would otherwise wait out its hazard)
                        for (int b = 0; b < LW; b++) pl[b] = (uint32_t)__builtin_amdgcn_readlane(lp[b], la);
                        auto unit = [&](int u) { // rows of lane la whose live unit u differs from this column's
                            uint32_t m = pl[4 * u] ^ cm[4 * u];
#pragma unroll
                            for (int b = 1; b < 4; b++) m = BITOP3(m, pl[4 * u + b], cm[4 * u + b], TT_A | (TT_B ^ TT_C));
                            return m;
                        };
                        uint32_t hg;
                        const uint32_t va = (uint32_t)__builtin_amdgcn_readlane(valid, la);
                        if (K == 1) {
                            const uint32_t any_a = (uint32_t)__builtin_amdgcn_readlane(pre[0], la);
                            const uint32_t two_a = (uint32_t)__builtin_amdgcn_readlane(pre[1], la);
                            const uint32_t t = BITOP3(any_a, unit(0), unit(1), (TT_A & TT_B) | (TT_A & TT_C) | (TT_B & TT_C));
                            hg = ~two_a & ~t & va;
                        } else {
                            uint32_t sc[K + 2];
#pragma unroll
                            for (int l = 0; l < K + 2; l++) sc[l] = (uint32_t)__builtin_amdgcn_readlane(pre[l], la);
                            count_units<K, 0, LIVE>(unit, sc);
                            hg = ~sc[K + 1] & va;
                        }
                        const uint32_t rbase = (group0 + (uint32_t)la) * 32; // first row of lane la
                        if (diag) { // only rows before the column: keeps the self pair and i > j out
                            const int d = (int)gcol - (int)rbase;
                            hg &= d <= 0 ? 0u : (d >= 32 ? 0xFFFFFFFFu : ((1u << d) - 1u));
                        }
                        if (col >= e) hg = 0;
                        if (__any(hg != 0)) {
                            while (hg) {
                                const int j = __builtin_ctz(hg);
                                hg &= hg - 1;
                                const unsigned int slot = atomicAdd(&hitq_count, 1u);
                                const uint2 hit = make_uint2(bucket_start + rbase + (uint32_t)j, c0 + col);
                                if (slot < HITQ) {
                                    hitq[slot] = hit;
                                } else { // queue full: to the global overflow list
                                    const unsigned long long pos = atomicAdd(&a.counters[CNT_OVF], 1ull);
                                    if (pos < a.ovf_cap) a.ovf[pos] = hit;
                                }
                            }
                            if ((uint32_t)__builtin_amdgcn_readfirstlane(*(volatile unsigned int *)&hitq_count) >= HITQ / 2)
                                drain(false);
                        }
                    }
                }
            }
            c = e;
        }
    };

    // The dense items first, one at a time (they take several times as long as the others), then
    // the rest in batches of BATCH neighbours of the list, which mostly share their row tile.
    // Both lists are dealt round-robin over the blocks (no hand-out counter: one hot word takes
    // ~90 atomics/us, tens of thousands of grabs cost more than the imbalance they remove), and
    // the grid is large enough that a block gets one dense item and one batch at most at config-2
    // sizes: the dispatcher then evens out what is left (blocks with nothing to do exit at
    // once).  pull() walks this block's share and ends with NONE for every block.  The loop runs
    // two items ahead with the records and one ahead with the column keys: they are loaded once
    // the keys in hand are in LDS, so that the walk never waits for memory.
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t phase = 0, b_cur = blockIdx.x, b_pos = 0; // b_cur: batch index within the phase's list
    auto pull = [&]() -> uint32_t {
        for (;;) {
            const uint32_t n_list = phase ? n_items : n_diag, bs = phase ? BATCH : 1u;
            const uint64_t first = (uint64_t)b_cur * bs;
            if (first + b_pos < n_list && b_pos < bs) {
                const uint32_t idx = (uint32_t)first + b_pos++;
                return phase ? idx : item_cap - 1 - idx;
            }
            if (first < n_list) { // the batch is done: this wave's next one
                b_cur += gridDim.x;
                b_pos = 0;
            } else if (phase == 0) { // the diagonal list is used up
                phase = 1;
                b_cur = blockIdx.x;
                b_pos = 0;
            } else {
                return NONE;
            }
        }
    };
    const TabItem none_item = {0u, 0u, 0u, 0u};
    uint32_t i0 = pull(), i1 = pull();
    TabItem rec0 = i0 != NONE ? items[i0] : none_item, rec1 = i1 != NONE ? items[i1] : none_item;
    uint32_t k0[KPL] = {}, k1[KPL] = {};
    if (i0 != NONE) load_keys(rec0, k0);
    while (i0 != NONE) {
        const uint32_t i2 = pull();
        const TabItem rec2 = i2 != NONE ? items[i2] : none_item;
        stage_item(rec0, k0);
        if (i1 != NONE) load_keys(rec1, k1); // under way while this item is walked
        walk_item();
        i0 = i1;
        rec0 = rec1;
#pragma unroll
        for (int j = 0; j < KPL; j++) k0[j] = k1[j];
        i1 = i2;
        rec1 = rec2;
    }
    drain(true);
    __syncthreads();
    if (tid == 0 && stage.candidates)
        atomicAdd(&a.counters[CNT_CANDIDATES], (unsigned long long)stage.candidates);
}

// ---- the overflow list of the bit-sliced kernels ------------------------------------
// Filter hits that found their block's LDS queue full: (row, column) in tile indices, one per
// thread through the same base-level pre-check and exact check as the queued ones.
template <typename KeyT>
__global__ __launch_bounds__(256) void verify_list_kernel(PairArgs a, uint32_t n_entries)
{
    __shared__ EdgeStage stage;
    if (threadIdx.x == 0) {
        stage.count = 0;
        stage.candidates = 0;
    }
    __syncthreads();
    const KeyT *__restrict__ fkey = (const KeyT *)a.fkey;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n_entries; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        if (i < n_entries) {
            const uint2 h = a.ovf[i];
            if (filter_key_distance(fkey[h.x], fkey[h.y]) <= a.k)
                verify_pair(a.keys, a.nmask, a.freq, a.thr, a.edges, a.edge_dist, a.counters, &stage,
                            a.edge_cap, a.k, a.mode, a.adj_max_freq, 0xFFFFFFFFu, 0xFFFFFFFFu, h.x, h.y,
                            a.perm);
        }
        flush_edges<256>(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, a.mode == MODE_NEIGHBOURS, true);
    }
    __syncthreads();
    if (threadIdx.x == 0 && stage.candidates)
        atomicAdd(&a.counters[CNT_CANDIDATES], (unsigned long long)stage.candidates);
}


// ---- collapse: directed min-rank label propagation ---------------------------
__global__ __launch_bounds__(256) void hook_kernel(const uint2 *__restrict__ edges,
                                                   const unsigned long long *counters,
                                                   uint32_t edge_cap, uint32_t *label,
                                                   uint32_t *changed, int round)
{
    if (round > 0 && changed[round - 1] == 0) return;
    unsigned long long ne = counters[CNT_EDGES];
    const uint32_t E = ne < edge_cap ? (uint32_t)ne : edge_cap;
    bool any = false;
    // four edges per thread per trip, all loads issued before the first use: the trip is a
    // chain of dependent L2 round trips otherwise
    constexpr int ILP = 4;
    const uint32_t nth = gridDim.x * blockDim.x;
    for (uint32_t e0 = blockIdx.x * blockDim.x + threadIdx.x; e0 < E; e0 += ILP * nth) {
        uint2 uv[ILP];
        uint32_t lu[ILP], lv[ILP];
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            const uint32_t e = e0 + i * nth;
            uv[i] = e < E ? edges[e] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            lu[i] = label[uv[i].x & ~SYM_FLAG];
            lv[i] = label[uv[i].y];
        }
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (lu[i] < lv[i]) {
                atomicMin(&label[uv[i].y], lu[i]);
                any = true;
            } else if ((uv[i].x & SYM_FLAG) && lv[i] < lu[i]) {
                atomicMin(&label[uv[i].x & ~SYM_FLAG], lv[i]);
                any = true;
            }
        }
    }
    if (any) changed[round] = 1;
}


// ---- directional collapse in two phases ------------------------------------------
// Reachability inside a set of entries joined by symmetric pairs (both directions permitted)
// is symmetric, and a one-way pair always leads to a strictly lower freq (thr is monotone in
// freq), so the one-way pairs form a DAG over those sets.  Phase 1: connected components over
// the symmetric pairs, comp[v] = smallest index of v's set, by hooking parents and
// grandparents and pointer jumping (a handful of rounds where plain label propagation needs as
// many as the longest chain is long).  Phase 2: lab[c] = smallest set index that reaches set c,
// propagated along the one-way pairs (rounds <= depth of the DAG).  Then label[v] = lab[comp[v]],
// the smallest rank that reaches v: what directional.rs:30-54,78-88 removes v under.
__global__ __launch_bounds__(256) void cc_hook_kernel(const uint2 *__restrict__ edges,
                                                      const unsigned long long *counters,
                                                      uint32_t edge_cap, uint32_t *comp,
                                                      uint32_t *changed, int round)
{
    if (round > 0 && changed[round - 1] == 0) return;
    unsigned long long ne = counters[CNT_EDGES];
    const uint32_t E = ne < edge_cap ? (uint32_t)ne : edge_cap;
    bool any = false;
    HotMin hot;
    constexpr int ILP = 4;
    const uint32_t nth = gridDim.x * blockDim.x;
    for (uint32_t b0 = blockIdx.x * blockDim.x; b0 < E; b0 += ILP * nth) { // (wave-uniform trip count)
        const uint32_t e0 = b0 + threadIdx.x;
        uint2 uv[ILP];
        uint32_t fu[ILP], fv[ILP], gu[ILP], gv[ILP];
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            const uint32_t e = e0 + i * nth;
            uv[i] = e < E ? edges[e] : make_uint2(0u, 0u); // (0, 0) is not a symmetric pair
        }
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            const bool sym = (uv[i].x & SYM_FLAG) != 0;
            fu[i] = sym ? comp[uv[i].x & ~SYM_FLAG] : 0u;
            fv[i] = sym ? comp[uv[i].y] : 0u;
        }
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            const bool live = fu[i] != fv[i]; // same parent: nothing to learn from this pair
            gu[i] = live ? comp[fu[i]] : 0u;
            gv[i] = live ? comp[fv[i]] : 0u;
        }
#pragma unroll
        for (int i = 0; i < ILP; i++) { // (every lane makes every trip: the shuffles need them all)
            const bool live = fu[i] != fv[i];
            const uint32_t lo = min(gu[i], gv[i]); // a member of the set, <= everything below
            // the parent with the larger grandparent: one target per pair, often a shared root
            const bool hook_u = live && lo < gu[i];
            wave_atomic_min(comp, hook_u ? fu[i] : fv[i], lo, live && (lo < gu[i] || lo < gv[i]), hot);
            any |= live; // parents differ: one of them moved, or will once the jump has run
        }
    }
    hot_flush(comp, hot);
    if (any) changed[round] = 1;
}


// (the pointer jump the hook rounds alternate with; the shipped library has its own copy next to
// the one-way rounds in umihip_kernels.hip)
__global__ __launch_bounds__(256) void jump_kernel(uint32_t *label, uint32_t n, uint32_t *changed,
                                                   int round)
{
    if (round > 0 && changed[round - 1] == 0) return;
    bool any = false;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        const uint32_t l = label[v];
        if (l != v) {
            const uint32_t ll = label[l];
            if (ll < l) {
                atomicMin(&label[v], ll);
                any = true;
            }
        }
    }
    if (any) changed[round] = 1;
}

} // namespace

hipError_t launch_prop_round(const uint2 *edges, const unsigned long long *counters,
                             uint32_t edge_cap, uint32_t *label, uint32_t n, uint32_t *changed,
                             int round, uint32_t n_edges_hint, hipStream_t s)
{
    hook_kernel<<<grid_for(n_edges_hint, 256), 256, 0, s>>>(edges, counters, edge_cap, label,
                                                            changed, round);
    jump_kernel<<<grid_for(n, 256), 256, 0, s>>>(label, n, changed, round);
    return hipGetLastError();
}

hipError_t launch_cc_round(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                           uint32_t *comp, uint32_t n, uint32_t *changed, int round,
                           uint32_t n_edges_hint, hipStream_t s)
{
    // few enough waves that most of them find a hot root already lowered when they get to it
    cc_hook_kernel<<<grid_for(n_edges_hint, 256, 512), 256, 0, s>>>(edges, counters, edge_cap, comp,
                                                                    changed, round);
    // two jumps per hook: a jump costs a twentieth of a hook and flattens the trees the next
    // hook walks
    jump_kernel<<<grid_for(n, 256), 256, 0, s>>>(comp, n, changed, round);
    jump_kernel<<<grid_for(n, 256), 256, 0, s>>>(comp, n, changed, round);
    return hipGetLastError();
}


hipError_t launch_build_planes(const void *fkey2, bool key32, const PlaneTask *tasks,
                               uint32_t n_tasks, uint32_t *planes, int umi_len, hipStream_t s)
{
    if (n_tasks == 0) return hipSuccess;
    const int np = 2 * bs_padded_len(umi_len);
    if (key32)
        build_planes_kernel<uint32_t><<<n_tasks, 64, 0, s>>>((const uint32_t *)fkey2, tasks, planes, np);
    else
        build_planes_kernel<uint64_t><<<n_tasks, 64, 0, s>>>((const uint64_t *)fkey2, tasks, planes, np);
    return hipGetLastError();
}

namespace {
template <typename KeyT, int LP, int G, int K>
void launch_bs_k(const PairArgs &a, uint32_t n_tasks, bool wide, int unit, int pu, hipStream_t s)
{
    // unit = bases per counted unit: 2 by default (filter = "at most K units differ", a
    // superset of distance <= K that verify_pair makes exact), 1 = exact base count,
    // 3 = fewer ops but 7x the false candidates on random 12-mers (k = 1, L' % 3 == 0 only).
    // pu = prefix units kept per run of columns (wide tiles of key-sorted buckets, unit 2)
    if (unit == 1) {
        if (wide) bs_pair_kernel<KeyT, LP, G, K, false, 1, 0><<<n_tasks, 256, 0, s>>>(a);
        else bs_pair_kernel<KeyT, LP, G, K, true, 1, 0><<<n_tasks, 256, 0, s>>>(a);
    } else if (unit == 3 && LP % 3 == 0 && K == 1) {
        if (wide) bs_pair_kernel<KeyT, (LP % 3 == 0 ? LP : 12), G, K, false, 3, 0><<<n_tasks, 256, 0, s>>>(a);
        else bs_pair_kernel<KeyT, (LP % 3 == 0 ? LP : 12), G, K, true, 3, 0><<<n_tasks, 256, 0, s>>>(a);
    } else if (wide && pu == 3) {
        bs_pair_kernel<KeyT, LP, G, K, false, 2, 3><<<n_tasks, 256, 0, s>>>(a);
    } else if (wide && pu == 4 && LP / 2 > 4) {
        bs_pair_kernel<KeyT, LP, G, K, false, 2, (LP / 2 > 4 ? 4 : 3)><<<n_tasks, 256, 0, s>>>(a);
    } else {
        if (wide) bs_pair_kernel<KeyT, LP, G, K, false, 2, 0><<<n_tasks, 256, 0, s>>>(a);
        else bs_pair_kernel<KeyT, LP, G, K, true, 2, 0><<<n_tasks, 256, 0, s>>>(a);
    }
}
template <typename KeyT, int LP, int G>
void launch_bs_lp(const PairArgs &a, uint32_t n_tasks, bool wide, int unit, int pu, hipStream_t s)
{
    switch (a.k) {
    case 0: launch_bs_k<KeyT, LP, G, 0>(a, n_tasks, wide, unit, pu, s); break;
    case 1: launch_bs_k<KeyT, LP, G, 1>(a, n_tasks, wide, unit, pu, s); break;
    case 2: launch_bs_k<KeyT, LP, G, 2>(a, n_tasks, wide, unit, pu, s); break;
    default: launch_bs_k<KeyT, LP, G, 3>(a, n_tasks, wide, unit, pu, s); break;
    }
}
} // namespace

namespace {
template <int LP, int K>
void launch_tab_k(const PairArgs &a, const TabRowTile *rts, uint32_t n_row_tiles, TabItem *items,
                  uint32_t item_cap, uint32_t part, uint32_t n_parts, uint32_t n_waves, bool transposed,
                  hipStream_t s)
{
    tab_scan_kernel<LP, K><<<n_row_tiles, TAB_SCAN_THREADS, 0, s>>>(a, rts, items, item_cap, part, n_parts);
    if (transposed) // columns of a run across the lanes, open row lanes one by one
        bs_run_kernel<LP, K><<<n_waves, 64, 0, s>>>(a, rts, items, item_cap);
    else // two live units looked up in register tables, every column against all rows
        bs_tab_kernel<LP, K, 2, BS_TAB_G><<<n_waves, 64, 0, s>>>(a, rts, items, item_cap);
}
template <int LP>
void launch_tab_lp(const PairArgs &a, const TabRowTile *rts, uint32_t n_row_tiles, TabItem *items,
                   uint32_t item_cap, uint32_t part, uint32_t n_parts, uint32_t n_waves, bool transposed,
                   hipStream_t s)
{
    switch (a.k) {
    case 0: launch_tab_k<LP, 0>(a, rts, n_row_tiles, items, item_cap, part, n_parts, n_waves, transposed, s); break;
    case 1: launch_tab_k<LP, 1>(a, rts, n_row_tiles, items, item_cap, part, n_parts, n_waves, transposed, s); break;
    case 2: launch_tab_k<LP, 2>(a, rts, n_row_tiles, items, item_cap, part, n_parts, n_waves, transposed, s); break;
    default: launch_tab_k<LP, 3>(a, rts, n_row_tiles, items, item_cap, part, n_parts, n_waves, transposed, s); break;
    }
}
} // namespace

// table variant: 32-bit keys, key-sorted buckets, BS_TAB_G row groups per lane, 2 live units.
// a.counters[CNT_ITEMS] must be 0; the scan fills items[] (capacity item_cap = the row tiles'
// column tiles, all of them), n_waves persistent one-wave blocks work it off.
hipError_t launch_bs_tab(const PairArgs &a, const TabRowTile *rts, uint32_t n_row_tiles, TabItem *items,
                         uint32_t item_cap, int umi_len, uint32_t part, uint32_t n_parts, uint32_t n_waves,
                         bool transposed, hipStream_t s)
{
    if (n_row_tiles == 0 || item_cap == 0) return hipSuccess;
    const int lp = bs_padded_len(umi_len);
    n_waves = std::max(1u, std::min(n_waves, (item_cap + 3) / 4));
    if (lp == 8) launch_tab_lp<8>(a, rts, n_row_tiles, items, item_cap, part, n_parts, n_waves, transposed, s);
    else if (lp == 12) launch_tab_lp<12>(a, rts, n_row_tiles, items, item_cap, part, n_parts, n_waves, transposed, s);
    else launch_tab_lp<16>(a, rts, n_row_tiles, items, item_cap, part, n_parts, n_waves, transposed, s);
    return hipGetLastError();
}

hipError_t launch_verify_list(const PairArgs &a, bool key32, uint32_t n_entries, hipStream_t s)
{
    if (n_entries == 0) return hipSuccess;
    if (key32) verify_list_kernel<uint32_t><<<grid_for(n_entries, 256, 1024), 256, 0, s>>>(a, n_entries);
    else verify_list_kernel<uint64_t><<<grid_for(n_entries, 256, 1024), 256, 0, s>>>(a, n_entries);
    return hipGetLastError();
}

hipError_t launch_bs_pairs(const PairArgs &a, uint32_t n_tasks, bool wide, bool key32,
                           int umi_len, int unit, int pu, hipStream_t s)
{
    if (n_tasks == 0) return hipSuccess;
    const int lp = bs_padded_len(umi_len);
    if (key32) {
        if (lp == 8) launch_bs_lp<uint32_t, 8, 2>(a, n_tasks, wide, unit, pu, s);
        else if (lp == 12) launch_bs_lp<uint32_t, 12, 2>(a, n_tasks, wide, unit, pu, s);
        else launch_bs_lp<uint32_t, 16, 2>(a, n_tasks, wide, unit, pu, s);
    } else {
        launch_bs_lp<uint64_t, 22, 1>(a, n_tasks, wide, unit, pu, s);
    }
    return hipGetLastError();
}


} // namespace umihip
