// gfx950 (MI355X / CDNA4) kernels of the UMI-collapse hot path.
//
// What they replace in the reference (tkob-vh/umi-collapse-rs):
//   pair_kernel      Naive::remove_near's linear scans (src/data/naive.rs:26-40) with the
//                    XOR-popcount distance of BitSet::bit_count_xor / umi_dist
//                    (src/utils/bitset.rs:77-91, src/utils/mod.rs:24-26), evaluated for all
//                    pairs of a bucket at once instead of row by row.
//   hook/jump        Directional::visit_and_remove + the root loop
//                    (src/algo/directional.rs:30-54,78-88) as a directed min-rank label
//                    fixed point: survivor(v) <=> no entry of smaller rank reaches v.
//   adj_*            Adjacency::apply's root loop (src/algo/adjacency.rs:52-60).
//
// Integer/bitwise work on 64-lane wavefronts: v_xor + v_bcnt + v_min3 per pair on a
// 2-bit-per-base filter key, column keys broadcast from an LDS tile, the exact
// 5-symbol distance only on the rare filter hits.  No MFMA (nothing here is a
// dense contraction).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

#include "umihip_internal.h"
#include "umihip_device.h"

namespace umihip {

namespace {

// Entry ranges a per-entry kernel works on: block b takes ranges[b] (<= RANGE_CHUNK entries);
// ranges == nullptr: all n entries, grid-stride.  The fused small-bucket kernel validates,
// thresholds and finalises its own buckets, so on a batch of 10^5 small positions these
// kernels have nothing to do.
template <class F>
__device__ __forceinline__ void for_entries(const RangeTask *__restrict__ ranges, uint32_t n, F f)
{
    if (ranges) {
        const RangeTask r = ranges[blockIdx.x];
        for (uint32_t i = r.start + threadIdx.x; i < r.end; i += blockDim.x) f(i);
    } else {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) f(i);
    }
}

__global__ __launch_bounds__(256) void prep_kernel(const uint64_t *__restrict__ keys,
                                                   const uint64_t *__restrict__ nmask,
                                                   const int32_t *__restrict__ freq,
                                                   const RangeTask *__restrict__ ranges, uint32_t n,
                                                   int umi_len, float percentage, int key32,
                                                   void *__restrict__ fkey,
                                                   int32_t *__restrict__ thr,
                                                   uint32_t *__restrict__ label,
                                                   uint32_t *__restrict__ lab,
                                                   unsigned long long *__restrict__ counters,
                                                   const SegDesc *__restrict__ segs, int n_seg_parts,
                                                   uint32_t *__restrict__ bin_cnt, int skip_seg, int key_words,
                                                   int full_umi_len)
{
    if (skip_seg && ranges && ranges[blockIdx.x].seg != SEG_NONE) return; // the count kernel's entries
    unsigned int bad = 0, rises = 0;
    // segment of this block's range (wave-uniform): its entries count themselves into the bins of
    // the segment's parts (the histogram of the counting sort, umihip_seg.hip)
    const uint32_t seg = ranges && segs ? ranges[blockIdx.x].seg : SEG_NONE;
    const SegDesc *__restrict__ sd = seg != SEG_NONE ? segs + seg : nullptr;
    for_entries(ranges, n, [&](uint32_t i) {
        // (keys of key_words > 1 words, umi_len 22..85: the filter key is made of the first word's 21
        // bases -- umi_len is 21 here -- and every filter hit is decided on all words)
        const uint64_t key = keys[(size_t)i * key_words];
        const uint64_t nm = nmask ? nmask[(size_t)i * key_words] : 0ull;
        const int32_t f = freq[i];
        thr[i] = threshold_of(percentage, f);
        label[i] = i;
        if (lab) lab[i] = i; // (the one-way rounds' labels: smallest set that reaches this one)
        // filter key: N (100, masked by n_bits) folded onto A (000) so that the
        // filter distance never exceeds the exact one.
        const uint64_t k3 = key & ~nm & (umi_len >= 21 ? 0x7FFFFFFFFFFFFFFFull : ((1ull << (3 * umi_len)) - 1ull));
        uint32_t fk = 0;
        if (key32) {
            for (int b = 0; b < umi_len; b++) fk |= (uint32_t)((k3 >> (3 * b)) & 3ull) << (2 * b);
            ((uint32_t *)fkey)[i] = fk;
        } else {
            ((uint64_t *)fkey)[i] = k3;
        }
        if (sd) {
            for (int j = 0; j < n_seg_parts; j++) {
                const uint32_t v = key32 ? seg_part_bits(fk, sd->b0[j], sd->nb[j])
                                         : seg_part_bits(k3, sd->b0[j], sd->nb[j]);
                atomicAdd(&bin_cnt[sd->bin_off[j] + v], 1u);
            }
        }
        // contract check: freq >= 1 and non-increasing inside a bucket.  A rise is legal only
        // at the first entry of a bucket: rises are counted here, rises at the starts of the
        // same buckets by bucket_rise_kernel, and the host requires the two counts to agree
        // (a per-entry search of the bucket table costs a chain of dependent loads in nearly
        // every wave).
        bad += f < 1 ? 1u : 0u;
        // nmask == NULL promises that no key holds the N code (100): the kernels then take the
        // filter key for the key.  A folded key has no base with bit 2 set and bits 0, 1 clear.
        const uint64_t b2 = k3 & 0x4924924924924924ull;
        bad += (b2 & ~((k3 << 1) | (k3 << 2))) != 0 ? 1u : 0u;
        if (key_words > 1 && !nmask) bad += wide_n_codes(keys + (size_t)i * key_words, key_words, full_umi_len);
        rises += (i > 0 && f > freq[i - 1]) ? 1u : 0u;
    });
    block_count_add(bad, &counters[CNT_ERROR]);
    block_count_add(rises, &counters[CNT_RISES]);
}

// rises at the first entry of the buckets the per-entry kernels cover (larger than min_size_m1)
__global__ __launch_bounds__(256) void bucket_rise_kernel(const int32_t *__restrict__ freq,
                                                          const uint64_t *__restrict__ bucket_off,
                                                          uint64_t n_buckets, uint32_t fused_max,
                                                          uint32_t skip_from, uint32_t n_entries,
                                                          unsigned long long *__restrict__ counters)
{
    // (buckets of skip_from entries and more are segments whose count kernel tells a rise at the
    // bucket's start from one inside it by itself)
    unsigned int rises = 0;
    for (uint64_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets;
         b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = bucket_off[b], e = bucket_off[b + 1];
        if (s > 0 && s < e && e <= (uint64_t)n_entries && e - s > fused_max && e - s < skip_from)
            rises += freq[s] > freq[s - 1] ? 1u : 0u;
    }
    block_count_add(rises, &counters[CNT_START_RISES]);
}

// All pairs (row, col) of one task through the filter.  Each lane keeps RPT row
// keys in registers; column keys are staged in LDS and read back as wave-wide
// broadcasts (ds_read_b128, same address in every lane), so a pair costs one
// v_xor, one v_bcnt and half a v_min3.
template <typename KeyT, int THREADS, int RPT>
__global__ __launch_bounds__(THREADS) void pair_kernel(PairArgs a)
{
    constexpr int VEC = 16 / (int)sizeof(KeyT);
    __shared__ __attribute__((aligned(16))) KeyT cols[COL_TILE];
    __shared__ EdgeStage stage;
    const PairTask *__restrict__ tp = a.tasks + blockIdx.x;
    // wave-uniform task fields (SGPRs): every loop bound below is scalar
    const uint32_t row0 = __builtin_amdgcn_readfirstlane(tp->row0);
    const uint32_t row_end = __builtin_amdgcn_readfirstlane(tp->row_end);
    const uint32_t col0 = __builtin_amdgcn_readfirstlane(tp->col0);
    const uint32_t col1 = __builtin_amdgcn_readfirstlane(tp->col1);
    const KeyT *__restrict__ fkey = (const KeyT *)a.fkey;
    const int tid = threadIdx.x;
    const int lim = 2 * a.k;
    const bool with_dist = a.mode == MODE_NEIGHBOURS;

    if (tid == 0) {
        stage.count = 0;
        stage.candidates = 0;
    }

    KeyT rk[RPT];
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        const uint32_t gi = row0 + r * THREADS + tid;
        rk[r] = gi < row_end ? fkey[gi] : pad_row<KeyT>();
    }

    for (uint32_t c0 = col0; c0 < col1; c0 += COL_TILE) {
        const int nc = (int)min((uint32_t)COL_TILE, col1 - c0);
        const int nc_pad = (nc + CHECK_BLOCK - 1) & ~(CHECK_BLOCK - 1);
        __syncthreads();
        for (int s = tid; s < nc_pad; s += THREADS) {
            const uint32_t gj = c0 + s;
            cols[s] = gj < col1 ? fkey[gj] : pad_col<KeyT>();
        }
        __syncthreads();
        for (int cb = 0; cb < nc; cb += CHECK_BLOCK) {
            int m[RPT];
#pragma unroll
            for (int r = 0; r < RPT; r++) m[r] = 255;
#pragma unroll
            for (int c = 0; c < CHECK_BLOCK; c += VEC) {
                KeyT cv[VEC];
                *reinterpret_cast<uint4 *>(cv) = *reinterpret_cast<const uint4 *>(&cols[cb + c]);
#pragma unroll
                for (int r = 0; r < RPT; r++) {
#pragma unroll
                    for (int v = 0; v < VEC; v += 2) // -> v_min3_u32(m, p0, p1)
                        m[r] = min(min(m[r], popc(rk[r] ^ cv[v])), popc(rk[r] ^ cv[v + 1]));
                }
            }
            int mm = m[0];
#pragma unroll
            for (int r = 1; r < RPT; r++) mm = min(mm, m[r]);
            if (__any(mm <= lim)) {
                // rare: re-walk this block of columns, only for the row slots that hit
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    if (__any(m[r] <= lim)) {
                        const uint32_t gi = row0 + r * THREADS + tid;
                        for (int c = 0; c < CHECK_BLOCK; c++) {
                            if (popc(rk[r] ^ cols[cb + c]) <= lim)
                                verify_pair(a.keys, a.nmask, a.freq, a.thr, a.edges, a.edge_dist,
                                            a.counters, &stage, a.edge_cap, a.k, a.mode,
                                            a.adj_max_freq, row_end, col1, gi, c0 + cb + c,
                                            nullptr);
                        }
                    }
                }
            }
        }
        flush_edges<THREADS>(&stage, a.edges, a.edge_dist, a.counters, a.edge_cap, with_dist,
                             c0 + COL_TILE >= col1);
    }
    __syncthreads();
    if (tid == 0 && stage.candidates)
        atomicAdd(&a.counters[CNT_CANDIDATES], (unsigned long long)stage.candidates);
}

// ---- fused small-bucket kernel (n <= 64*RL): one wave per bucket ------------------
// A bucket's rows sit one (or two) per lane; column j's key, nmask and threshold are
// broadcast with v_readlane (j is wave-uniform), the exact reference distance
// (bitset.rs:85-87) decides each pair, the permitted in-edges of a row are kept as bit
// masks in registers, and the collapse runs in-wave: Gauss-Seidel min-label sweeps in
// rank order for directional (directional.rs:30-54,78-88), the sequential root loop for
// adjacency (adjacency.rs:52-60).  No LDS, no edge list.  The kernel also does for its buckets
// what prep_kernel and finalize_kernel do for the others (contract check, thresholds, kept
// mask, root, survivor count): a batch of small positions is this one launch.
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane)
{
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// one bucket [start, start + n), n <= 64 * RL, by the calling wave
// what one fused bucket hands back: contract violations seen, survivors written
struct FusedCounts {
    unsigned int bad, kept;
};

// contract check of a fused bucket (prep_kernel does it for the others): freq >= 1 and
// non-increasing from row to row; fr[s] of the rows past n is a sentinel and not looked at
template <int RL>
__device__ __forceinline__ unsigned int fused_bad_rows(const int32_t (&fr)[RL], int n)
{
    const int lane = threadIdx.x & 63;
    unsigned int bad = 0;
#pragma unroll
    for (int s = 0; s < RL; s++) {
        const int row = lane + 64 * s;
        int32_t prev = __shfl_up(fr[s], 1);
        if (lane == 0) prev = s == 0 ? 0x7FFFFFFF : __builtin_amdgcn_readlane(fr[0], 63);
        if (row < n) bad += (fr[s] < 1 ? 1u : 0u) + ((row > 0 && fr[s] > prev) ? 1u : 0u);
    }
    return bad;
}

// label, kept mask and root of a fused bucket's rows (what finalize_kernel / adj_finalize_kernel
// write for the others); returns the survivors among this lane's rows
template <int RL, int MODE>
__device__ __forceinline__ unsigned int fused_write_out(uint32_t start, int n, const uint32_t (&lab)[RL],
                                                        const uint32_t (&alive)[RL],
                                                        uint32_t *__restrict__ label,
                                                        uint8_t *__restrict__ kept,
                                                        uint32_t *__restrict__ root)
{
    const int lane = threadIdx.x & 63;
    unsigned int n_kept = 0;
#pragma unroll
    for (int s = 0; s < RL; s++) {
        const int row = lane + 64 * s;
        if (row < n) {
            const bool kp = MODE == MODE_DIRECTIONAL ? lab[s] == (uint32_t)row : alive[s] != 0u;
            if (label) label[start + row] = start + lab[s];
            kept[start + row] = kp ? 1 : 0;
            if (root) root[start + row] = (MODE != MODE_DIRECTIONAL && kp) ? start + row : start + lab[s];
            n_kept += kp ? 1u : 0u;
        }
    }
    return n_kept;
}

template <int RL, bool HAS_N, int MODE>
__device__ __forceinline__ FusedCounts small_bucket_body(const uint64_t *__restrict__ keys,
                                                         const uint64_t *__restrict__ nmask,
                                                         const int32_t *__restrict__ freq,
                                                         float percentage, uint32_t start, int n,
                                                         uint32_t *__restrict__ label,
                                                         uint8_t *__restrict__ kept,
                                                         uint32_t *__restrict__ root, int k,
                                                         int32_t adj_max_freq)
{
    const int lane = threadIdx.x & 63;
    constexpr int H = 2 * RL; // the bucket's entries in halves of 32: half h = entries 32h..32h+31
    uint64_t key[RL], nm[RL];
    int32_t fr[RL], th[RL];
#pragma unroll
    for (int s = 0; s < RL; s++) {
        const int r = lane + 64 * s;
        const bool in_range = r < n;
        key[s] = in_range ? keys[start + r] : 0ull;
        nm[s] = (HAS_N && in_range) ? nmask[start + r] : 0ull;
        fr[s] = in_range ? freq[start + r] : 0x7FFFFFFF;
        th[s] = in_range ? threshold_of(percentage, fr[s]) : (-0x7FFFFFFF - 1);
    }
    // in[s][h]: bit jj set <=> entry j = 32h+jj may remove row (lane + 64s)
    uint32_t in[RL][H];
#pragma unroll
    for (int s = 0; s < RL; s++)
#pragma unroll
        for (int h = 0; h < H; h++) in[s][h] = 0u;
    const int lim = 2 * k + 1; // bit_count_xor / 2 <= k  <=>  bit_count_xor <= 2k+1
#pragma unroll
    for (int h = 0; h < H; h++) {
        const int jn = min(32, n - 32 * h);
        for (int jj = 0; jj < jn; jj++) {
            const int src = 32 * (h & 1) + jj; // lane that holds entry j in slot h/2
            const uint64_t kj = readlane64(key[h >> 1], src);
            const uint64_t nj = HAS_N ? readlane64(nm[h >> 1], src) : 0ull;
            const int32_t thj = __builtin_amdgcn_readlane(th[h >> 1], src);
            const int j = 32 * h + jj;
            const uint32_t bit = 1u << jj; // wave-uniform
#pragma unroll
            for (int s = 0; s < RL; s++) {
                const uint64_t x = nm[s] ^ nj;
                const int bcx = __builtin_popcountll(x | (key[s] ^ kj)) -
                                (HAS_N ? __builtin_popcountll(x) / 3 : 0); // bitset.rs:85-87
                bool e = bcx <= lim && (lane + 64 * s) != j;
                if (MODE == MODE_DIRECTIONAL)
                    e = e && fr[s] <= thj; // naive.rs:31 with max_freq = threshold(start)
                else
                    e = e && fr[s] <= adj_max_freq && (lane + 64 * s) > j;
                in[s][h] |= e ? bit : 0u;
            }
        }
    }
    uint32_t lab[RL], alive[RL];
#pragma unroll
    for (int s = 0; s < RL; s++) {
        lab[s] = (uint32_t)(lane + 64 * s);
        alive[s] = 0u;
    }
    if (MODE == MODE_DIRECTIONAL) {
        bool changed;
        do { // Gauss-Seidel sweeps in rank order until no label moves
            uint32_t before[RL];
#pragma unroll
            for (int s = 0; s < RL; s++) before[s] = lab[s];
#pragma unroll
            for (int h = 0; h < H; h++) {
                const int jn = min(32, n - 32 * h);
                for (int jj = 0; jj < jn; jj++) {
                    const uint32_t lj = __builtin_amdgcn_readlane(lab[h >> 1], 32 * (h & 1) + jj);
                    const uint32_t bit = 1u << jj;
#pragma unroll
                    for (int s = 0; s < RL; s++) lab[s] = (in[s][h] & bit) ? min(lab[s], lj) : lab[s];
                }
            }
            changed = false;
#pragma unroll
            for (int s = 0; s < RL; s++) changed |= lab[s] != before[s];
        } while (__any(changed));
    } else {
#pragma unroll
        for (int s = 0; s < RL; s++) alive[s] = 1u;
#pragma unroll
        for (int h = 0; h < H; h++) {
            const int jn = min(32, n - 32 * h);
            for (int jj = 0; jj < jn; jj++) {
                if (__builtin_amdgcn_readlane(alive[h >> 1], 32 * (h & 1) + jj)) { // j is a root (adjacency.rs:54)
                    const uint32_t bit = 1u << jj;
#pragma unroll
                    for (int s = 0; s < RL; s++) {
                        if ((in[s][h] & bit) && alive[s]) {
                            alive[s] = 0u;
                            lab[s] = (uint32_t)(32 * h + jj);
                        }
                    }
                }
            }
        }
    }
    FusedCounts out;
    out.bad = fused_bad_rows<RL>(fr, n);
    out.kept = fused_write_out<RL, MODE>(start, n, lab, alive, label, kept, root);
    return out;
}

// ---- the same, with the distances bit-sliced over the bucket's columns -------------
// A lane still owns one row (two for 65..128 entries), but instead of walking the columns one
// by one it meets 64 of them at a time: for base i the two code bits of all entries are
// wave-wide ballots (P0, P1: one bit per column), the lane XORs them with its own two code
// bits and feeds a sticky counter of mismatching bases in v_bitop3 -- 8 wave-instructions per
// base for 64 pairs per lane instead of ~17 per column.  ~s[K+1] is the row's set of columns
// within distance K (exact on N-free keys; with N the hits are re-checked exactly through
// ds_bpermute).  The permitted in-edges follow from one binary search per row when thr[] is
// non-increasing inside the bucket (it is unless percentage < 0 or freq + 1 wrapped; the wave
// checks): {j : freq[i] <= thr[j]} is then a prefix.  The collapse then sweeps only over the entries that have an out-edge at all.
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= (uint32_t)__shfl_xor((int)v, off);
    return v;
}

template <int RL, bool HAS_N, int MODE, int K>
__device__ __forceinline__ FusedCounts small_bucket_body_bs(const uint64_t *__restrict__ keys,
                                                            const uint64_t *__restrict__ nmask,
                                                            const int32_t *__restrict__ freq,
                                                            float percentage, uint32_t start, int n,
                                                            int umi_len, uint32_t *__restrict__ label,
                                                            uint8_t *__restrict__ kept,
                                                            uint32_t *__restrict__ root,
                                                            int32_t adj_max_freq)
{
    const int lane = threadIdx.x & 63;
    constexpr int H = 2 * RL; // column halves of 32
    uint64_t key[RL], nm[RL], fold[RL];
    int32_t fr[RL], th[RL];
#pragma unroll
    for (int s = 0; s < RL; s++) {
        const int r = lane + 64 * s;
        const bool in_range = r < n;
        key[s] = in_range ? keys[start + r] : 0ull;
        nm[s] = (HAS_N && in_range) ? nmask[start + r] : 0ull;
        fold[s] = key[s] & ~nm[s]; // N folded onto A: the sliced distance never exceeds the exact one
        fr[s] = in_range ? freq[start + r] : 0x7FFFFFFF;
        th[s] = in_range ? threshold_of(percentage, fr[s]) : (-0x7FFFFFFF - 1);
    }
    // cnt[s][h][l]: columns of half h at which row (lane + 64 s) has more than l mismatches so far
    uint32_t cnt[RL][H][K + 1];
#pragma unroll
    for (int s = 0; s < RL; s++)
#pragma unroll
        for (int h = 0; h < H; h++)
#pragma unroll
            for (int l = 0; l <= K; l++) cnt[s][h][l] = 0u;
    uint64_t cur[RL]; // the folded key, shifted down a base per trip
#pragma unroll
    for (int s = 0; s < RL; s++) cur[s] = fold[s];
    for (int i = 0; i < umi_len; i++) {
        uint32_t r0[RL], r1[RL]; // this row's two code bits of base i, spread over a word (one v_bfe_i32 each)
#pragma unroll
        for (int s = 0; s < RL; s++) {
            r0[s] = (uint32_t)__builtin_amdgcn_sbfe((int)(uint32_t)cur[s], 0u, 1u);
            r1[s] = (uint32_t)__builtin_amdgcn_sbfe((int)(uint32_t)cur[s], 1u, 1u);
            cur[s] >>= 3;
        }
#pragma unroll
        for (int w = 0; w < RL; w++) {
            const unsigned long long p0 = __ballot(r0[w] != 0u);
            const unsigned long long p1 = __ballot(r1[w] != 0u);
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const uint32_t c0 = (uint32_t)(p0 >> (32 * hh)), c1 = (uint32_t)(p1 >> (32 * hh));
#pragma unroll
                for (int s = 0; s < RL; s++) {
                    const uint32_t m = BITOP3(c0 ^ r0[s], c1, r1[s], TT_A | (TT_B ^ TT_C));
#pragma unroll
                    for (int l = K; l >= 1; l--)
                        cnt[s][2 * w + hh][l] = BITOP3(cnt[s][2 * w + hh][l], cnt[s][2 * w + hh][l - 1], m,
                                                       TT_A | (TT_B & TT_C));
                    cnt[s][2 * w + hh][0] |= m;
                }
            }
        }
    }
    uint32_t in[RL][H];
#pragma unroll
    for (int s = 0; s < RL; s++)
#pragma unroll
        for (int h = 0; h < H; h++) {
            const int nv = n - 32 * h; // valid columns in this half
            const uint32_t colvalid = nv >= 32 ? 0xFFFFFFFFu : (nv <= 0 ? 0u : ((1u << nv) - 1u));
            in[s][h] = ~cnt[s][h][K] & colvalid;
        }
    if (HAS_N) { // exact re-check of the hits (bitset.rs:85-87), column values through ds_bpermute
#pragma unroll
        for (int s = 0; s < RL; s++)
#pragma unroll
            for (int h = 0; h < H; h++) {
                uint32_t todo = in[s][h];
                while (__any(todo != 0)) { // all lanes stay in the loop: bpermute needs its sources
                    const int jj = todo ? __builtin_ctz(todo) : 0;
                    const int src = 32 * (h & 1) + jj;
                    const uint64_t kj = ((uint64_t)(uint32_t)__shfl((int)(key[h >> 1] >> 32), src) << 32) |
                                        (uint32_t)__shfl((int)key[h >> 1], src);
                    const uint64_t nj = ((uint64_t)(uint32_t)__shfl((int)(nm[h >> 1] >> 32), src) << 32) |
                                        (uint32_t)__shfl((int)nm[h >> 1], src);
                    if (todo) {
                        const uint64_t x = nm[s] ^ nj;
                        const int bcx = __builtin_popcountll(x | (key[s] ^ kj)) - __builtin_popcountll(x) / 3;
                        if (bcx > 2 * K + 1) in[s][h] &= ~(1u << jj);
                        todo &= todo - 1;
                    }
                }
            }
    }
    // which sources may remove this row: freq condition + not itself.  freq is non-increasing
    // inside a bucket, so thr is too unless percentage < 0 or freq + 1 wrapped (directional.rs:
    // 100-102): then {j : freq[row] <= thr[j]} is a prefix and one binary search finds its end.
    bool monotone = true;
    if (MODE == MODE_DIRECTIONAL) {
        bool rise = false;
#pragma unroll
        for (int s = 0; s < RL; s++) {
            int32_t prev = __shfl_up(th[s], 1);
            if (lane == 0) prev = s == 0 ? 0x7FFFFFFF : __builtin_amdgcn_readlane(th[0], 63);
            rise |= th[s] > prev;
        }
        monotone = !__any(rise);
    }
#pragma unroll
    for (int s = 0; s < RL; s++) {
        const int row = lane + 64 * s;
        int q = 0; // sources 0..q-1 pass the freq test
        if (MODE == MODE_DIRECTIONAL) {
            if (monotone) {
                // first j with thr[j] < freq[row] (naive.rs:31 with max_freq = threshold(start)):
                // every source before it may remove this row
                int lo = 0, hi = n;
#pragma unroll
                for (int it = 0; it < (RL == 1 ? 7 : 8); it++) {
                    const int mid = min((lo + hi) >> 1, n - 1);
                    int t = __shfl(th[0], mid & 63);
                    if (RL == 2) {
                        const int t1 = __shfl(th[RL - 1], mid & 63);
                        t = mid >= 64 ? t1 : t;
                    }
                    if (lo < hi) {
                        if (t >= fr[s]) lo = mid + 1; else hi = mid;
                    }
                }
                q = lo;
            }
        } else {
            q = fr[s] <= adj_max_freq ? row : 0; // adjacency.rs:56: only earlier roots, freq <= max_freq
        }
#pragma unroll
        for (int h = 0; h < H; h++) {
            uint32_t pre;
            if (MODE == MODE_DIRECTIONAL && !monotone) {
                pre = 0u; // thresholds out of order: test every source (wave-uniform branch)
                const int jn = min(32, n - 32 * h);
                for (int jj = 0; jj < jn; jj++) {
                    const int32_t thj = __builtin_amdgcn_readlane(th[h >> 1], 32 * (h & 1) + jj);
                    pre |= fr[s] <= thj ? 1u << jj : 0u;
                }
            } else {
                const int nb = q - 32 * h;
                pre = nb >= 32 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
            }
            uint32_t self = 0u;
            if (row >= 32 * h && row < 32 * h + 32) self = 1u << (row - 32 * h);
            in[s][h] &= pre & ~self;
            if (row >= n) in[s][h] = 0u;
        }
    }
    // sources that have an out-edge at all (wave-uniform)
    uint32_t act[H];
#pragma unroll
    for (int h = 0; h < H; h++) {
        uint32_t v = 0;
#pragma unroll
        for (int s = 0; s < RL; s++) v |= in[s][h];
        act[h] = __builtin_amdgcn_readfirstlane(wave_or(v));
    }
    uint32_t lab[RL], alive[RL];
#pragma unroll
    for (int s = 0; s < RL; s++) {
        lab[s] = (uint32_t)(lane + 64 * s);
        alive[s] = 0u;
    }
    if (MODE == MODE_DIRECTIONAL) {
        // A sweep visits the sources in rank order.  If no row has an in-edge from a source of
        // larger rank, every source's label is final when its turn comes and one sweep is the
        // fixed point; such an edge needs freq[row] <= thr[source] with the source ranked behind
        // the row: at p = 0.5 two freq-1 UMIs next to each other, rare in a position.  Otherwise
        // the sweeps go on until one moves nothing.
        bool back = false;
#pragma unroll
        for (int s = 0; s < RL; s++) {
            const int row = lane + 64 * s;
#pragma unroll
            for (int h = 0; h < H; h++) {
                const int above = row - 32 * h + 1; // columns of this half from here on rank behind the row
                const uint32_t m = above <= 0 ? 0xFFFFFFFFu : (above >= 32 ? 0u : ~((1u << above) - 1u));
                back |= (in[s][h] & m) != 0u;
            }
        }
        const bool iterate = __any(back);
        bool changed;
        do { // Gauss-Seidel sweeps in rank order over the active sources until no label moves
            uint32_t before[RL];
#pragma unroll
            for (int s = 0; s < RL; s++) before[s] = lab[s];
#pragma unroll
            for (int h = 0; h < H; h++) {
                uint32_t todo = act[h];
                while (todo) {
                    const int jj = __builtin_ctz(todo);
                    todo &= todo - 1;
                    const uint32_t lj = __builtin_amdgcn_readlane(lab[h >> 1], 32 * (h & 1) + jj);
                    const uint32_t bit = 1u << jj;
#pragma unroll
                    for (int s = 0; s < RL; s++) lab[s] = (in[s][h] & bit) ? min(lab[s], lj) : lab[s];
                }
            }
            changed = false;
#pragma unroll
            for (int s = 0; s < RL; s++) changed |= lab[s] != before[s];
        } while (iterate && __any(changed));
    } else {
#pragma unroll
        for (int s = 0; s < RL; s++) alive[s] = 1u;
#pragma unroll
        for (int h = 0; h < H; h++) {
            uint32_t todo = act[h];
            while (todo) {
                const int jj = __builtin_ctz(todo);
                todo &= todo - 1;
                if (__builtin_amdgcn_readlane(alive[h >> 1], 32 * (h & 1) + jj)) { // j is a root
                    const uint32_t bit = 1u << jj;
#pragma unroll
                    for (int s = 0; s < RL; s++) {
                        if ((in[s][h] & bit) && alive[s]) {
                            alive[s] = 0u;
                            lab[s] = (uint32_t)(32 * h + jj);
                        }
                    }
                }
            }
        }
    }
    FusedCounts out;
    out.bad = fused_bad_rows<RL>(fr, n);
    out.kept = fused_write_out<RL, MODE>(start, n, lab, alive, label, kept, root);
    return out;
}

// ---- the directional body again, counted instruction by instruction ------------------------------
// A batch of small positions is this kernel and nothing else, and the kernel is bound by VALU issue
// (SQ counters: ~580 wave instructions per position of ~60 UMIs, 8 waves per SIMD): what it saves
// it saves in instructions.  Same algorithm as small_bucket_body_bs, with
//  * the rows' code bits taken four bases at a time out of one 12-bit window (one 64-bit shift per
//    four bases instead of one per base);
//  * q[row] = how many sources may remove the row (thr[j] >= freq[row]: a prefix, thresholds being in
//    rank order) and q'[j] = the first row source j may remove (freq[r] <= thr[j]: a suffix) found
//    per DISTINCT freq of the bucket -- a handful -- with two ballots each, instead of a seven-step
//    binary search over ds_bpermute per row;
//  * the sources that have an out-edge at all as ONE ballot: distances are symmetric, so lane j's own
//    hit mask, cut to the rows from q'[j] on, says whether j removes anything (was: an OR-reduction
//    of the in-edge masks over the wave, six ds_bpermute steps per half);
//  * neighbour values through DPP (wave_shr) instead of ds_bpermute; label[] not written (nothing
//    reads it for a finished bucket).
// Thresholds out of rank order (percentage < 0, freq + 1 wrapped) go to small_bucket_body_bs.
__device__ __forceinline__ int32_t lane_above(int32_t v, int32_t first)
{ // the value of the lane before (lane 0: `first`): DPP wave_shr:1, one instruction
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t low_bits(int nb)
{ // nb <= 0: none, nb >= 32: all
    return nb >= 32 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
}
__device__ __forceinline__ unsigned long long low_bits64(int nb)
{
    return nb >= 64 ? ~0ull : (nb <= 0 ? 0ull : ((1ull << nb) - 1ull));
}

// W = words per key (1: umi_len <= 21; 2..4: up to 85 bases, keys entry-major [i * W + w], the
// distance word by word as src/utils/bitset.rs:77-91 has it).
// twelve bits of a W-word string from bit 3 * g on (g wave-uniform); a window may run over a word's end
template <int W> __device__ __forceinline__ uint32_t window12(const uint64_t (&f)[W], int g)
{
    if (W == 1) return (uint32_t)(f[0] >> (3 * g));
    const int bit = 3 * g, wi = bit >> 6, sh = bit & 63;
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int w = 0; w < W; w++)
        if (wi == w) {
            lo = f[w];
            hi = w + 1 < W ? f[w + 1] : 0ull;
        }
    uint64_t v = lo >> sh;
    if (sh > 52) v |= hi << (64 - sh);
    return (uint32_t)v;
}

template <int RL, bool HAS_N, int K, int W>
__device__ __forceinline__ FusedCounts small_bucket_body_dir(const uint64_t *__restrict__ keys,
                                                             const uint64_t *__restrict__ nmask,
                                                             const int32_t *__restrict__ freq,
                                                             float percentage, uint32_t start, int n,
                                                             int umi_len, uint32_t *__restrict__ label,
                                                             uint8_t *__restrict__ kept,
                                                             uint32_t *__restrict__ root)
{
    const int lane = threadIdx.x & 63;
    constexpr int H = 2 * RL;
    uint64_t key[RL][W], nm[RL][W], fold[RL][W];
    int32_t fr[RL], th[RL];
    unsigned long long vmask[RL]; // lanes whose row of slot s exists (wave-uniform)
#pragma unroll
    for (int s = 0; s < RL; s++) {
        const int r = lane + 64 * s;
        const bool in_range = r < n;
#pragma unroll
        for (int w = 0; w < W; w++) {
            key[s][w] = in_range ? keys[(size_t)(start + r) * W + w] : 0ull;
            nm[s][w] = (HAS_N && in_range) ? nmask[(size_t)(start + r) * W + w] : 0ull;
            fold[s][w] = key[s][w] & ~nm[s][w]; // N folded onto A: the sliced distance never exceeds the exact one
        }
        fr[s] = in_range ? freq[start + r] : 0x7FFFFFFF;
        th[s] = in_range ? threshold_of(percentage, fr[s]) : (-0x7FFFFFFF - 1);
        const int left = n - 64 * s;
        vmask[s] = left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
    }
    // thresholds in rank order?  (and the contract: freq >= 1, not rising)
    bool rise_th = false;
    unsigned int bad = 0;
#pragma unroll
    for (int s = 0; s < RL; s++) {
        const int32_t th_first = s == 0 ? 0x7FFFFFFF : __builtin_amdgcn_readlane(th[0], 63);
        const int32_t fr_first = s == 0 ? 0x7FFFFFFF : __builtin_amdgcn_readlane(fr[0], 63);
        rise_th |= th[s] > lane_above(th[s], th_first);
        const bool row = lane + 64 * s < n;
        bad += (row && fr[s] < 1 ? 1u : 0u) + (row && fr[s] > lane_above(fr[s], fr_first) ? 1u : 0u);
    }
    const bool monotone = !__any(rise_th);
    // cnt[s][h][l]: columns of half h at which row (lane + 64 s) has more than l mismatches so far
    uint32_t cnt[RL][H][K + 1];
#pragma unroll
    for (int s = 0; s < RL; s++)
#pragma unroll
        for (int h = 0; h < H; h++)
#pragma unroll
            for (int l = 0; l <= K; l++) cnt[s][h][l] = 0u;
    uint32_t n_code[RL]; // (keys of several words without nmask: a base that holds the N code, 100)
#pragma unroll
    for (int s = 0; s < RL; s++) n_code[s] = 0u;
    auto one_base = [&](const uint32_t (&w)[RL], int j) { // base j (a constant) of the window
        uint32_t r0[RL], r1[RL];
#pragma unroll
        for (int s = 0; s < RL; s++) {
            r0[s] = (uint32_t)__builtin_amdgcn_sbfe((int)w[s], 3u * (uint32_t)j, 1u);
            r1[s] = (uint32_t)__builtin_amdgcn_sbfe((int)w[s], 3u * (uint32_t)j + 1u, 1u);
            if (W > 1 && !HAS_N) n_code[s] |= (w[s] >> (3 * j + 2)) & ~r0[s] & ~r1[s] & 1u;
        }
#pragma unroll
        for (int c = 0; c < RL; c++) {
            const unsigned long long p0 = __builtin_amdgcn_uicmp(r0[c], 0u, 33 /* ICMP_NE */);
            const unsigned long long p1 = __builtin_amdgcn_uicmp(r1[c], 0u, 33);
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const uint32_t c0 = (uint32_t)(p0 >> (32 * hh)), c1 = (uint32_t)(p1 >> (32 * hh));
#pragma unroll
                for (int s = 0; s < RL; s++) {
                    const uint32_t m = BITOP3(c0 ^ r0[s], c1, r1[s], TT_A | (TT_B ^ TT_C));
#pragma unroll
                    for (int l = K; l >= 1; l--)
                        cnt[s][2 * c + hh][l] = BITOP3(cnt[s][2 * c + hh][l], cnt[s][2 * c + hh][l - 1], m,
                                                       TT_A | (TT_B & TT_C));
                    cnt[s][2 * c + hh][0] |= m;
                }
            }
        }
    };
    int g = 0;
    for (; g + 4 <= umi_len; g += 4) { // four bases of every row: 12 bits
        uint32_t w[RL];
#pragma unroll
        for (int s = 0; s < RL; s++) w[s] = window12<W>(fold[s], g);
        one_base(w, 0);
        one_base(w, 1);
        one_base(w, 2);
        one_base(w, 3);
    }
    for (; g < umi_len; g++) {
        uint32_t w[RL];
#pragma unroll
        for (int s = 0; s < RL; s++) w[s] = window12<W>(fold[s], g);
        one_base(w, 0);
    }
    // hits[s][h]: columns of half h within K of row (lane + 64 s), the row itself included
    uint32_t hits[RL][H];
#pragma unroll
    for (int s = 0; s < RL; s++)
#pragma unroll
        for (int h = 0; h < H; h++) hits[s][h] = ~cnt[s][h][K] & low_bits(n - 32 * h);
    if (HAS_N) { // exact re-check of the hits (bitset.rs:77-91), column values through ds_bpermute
#pragma unroll
        for (int s = 0; s < RL; s++)
#pragma unroll
            for (int h = 0; h < H; h++) {
                uint32_t todo = hits[s][h];
                while (__any(todo != 0)) { // all lanes stay in the loop: bpermute needs its sources
                    const int jj = todo ? __builtin_ctz(todo) : 0;
                    const int src = 32 * (h & 1) + jj;
                    int bcx = 0;
#pragma unroll
                    for (int w = 0; w < W; w++) {
                        const uint64_t kj = ((uint64_t)(uint32_t)__shfl((int)(key[h >> 1][w] >> 32), src) << 32) |
                                            (uint32_t)__shfl((int)key[h >> 1][w], src);
                        const uint64_t nj = ((uint64_t)(uint32_t)__shfl((int)(nm[h >> 1][w] >> 32), src) << 32) |
                                            (uint32_t)__shfl((int)nm[h >> 1][w], src);
                        const uint64_t x = nm[s][w] ^ nj;
                        bcx += __builtin_popcountll(x | (key[s][w] ^ kj)) - __builtin_popcountll(x) / 3;
                    }
                    if (todo) {
                        if (bcx > 2 * K + 1) hits[s][h] &= ~(1u << jj);
                        todo &= todo - 1;
                    }
                }
            }
    }
    // in-edges of the rows, and the sources that have an out-edge at all; columns 64 at a time
    unsigned long long in[RL][RL], act64[RL];
    if (monotone) {
        // q / q' per distinct freq of the bucket (a wave-uniform loop; the lanes of one freq leave together)
        int q[RL], qp[RL];
        unsigned long long todo[RL];
#pragma unroll
        for (int s = 0; s < RL; s++) {
            q[s] = 0;
            qp[s] = 0;
            todo[s] = vmask[s];
        }
        for (;;) {
            int32_t v, tv;
            if (todo[0]) {
                const int l = __builtin_ctzll(todo[0]);
                v = __builtin_amdgcn_readlane(fr[0], l);
                tv = __builtin_amdgcn_readlane(th[0], l);
            } else if (RL == 2 && todo[RL - 1]) {
                const int l = __builtin_ctzll(todo[RL - 1]);
                v = __builtin_amdgcn_readlane(fr[RL - 1], l);
                tv = __builtin_amdgcn_readlane(th[RL - 1], l);
            } else {
                break;
            }
            int n_src = 0, n_above = 0;
#pragma unroll
            for (int s = 0; s < RL; s++) {
                n_src += __builtin_popcountll(__ballot(th[s] >= v) & vmask[s]);   // sources that may remove a row of freq v
                n_above += __builtin_popcountll(__ballot(fr[s] > tv) & vmask[s]); // rows a source of this freq may not remove
            }
#pragma unroll
            for (int s = 0; s < RL; s++) {
                const bool mine = fr[s] == v;
                q[s] = mine ? n_src : q[s];
                qp[s] = mine ? n_above : qp[s];
                todo[s] &= ~__ballot(mine);
            }
        }
#pragma unroll
        for (int s = 0; s < RL; s++) {
            unsigned long long out_any = 0;
#pragma unroll
            for (int c = 0; c < RL; c++) {
                const unsigned long long hit = ((unsigned long long)hits[s][2 * c + 1] << 32) | hits[s][2 * c];
                const unsigned long long h_ns = s == c ? hit & ~(1ull << lane) : hit; // not the row itself
                in[s][c] = h_ns & low_bits64(q[s] - 64 * c);
                out_any |= h_ns & ~low_bits64(qp[s] - 64 * c);
            }
            act64[s] = __ballot(out_any != 0ull) & vmask[s];
        }
    } else {
        // thresholds out of rank order (percentage < 0, freq + 1 wrapped): every (row, source) pair is
        // tested by itself, a source per trip (rare, so it only has to be right)
        unsigned long long may[RL][RL], gives[RL][RL]; // may[s][c]: sources of slot c that pass row s's freq test;
                                                       // gives[s][c]: rows of slot c the source in slot s may remove
#pragma unroll
        for (int s = 0; s < RL; s++)
#pragma unroll
            for (int c = 0; c < RL; c++) may[s][c] = gives[s][c] = 0ull;
#pragma unroll
        for (int c = 0; c < RL; c++) {
            const int jn = min(64, n - 64 * c);
            for (int j = 0; j < jn; j++) {
                const int32_t thj = __builtin_amdgcn_readlane(th[c], j), frj = __builtin_amdgcn_readlane(fr[c], j);
#pragma unroll
                for (int s = 0; s < RL; s++) {
                    may[s][c] |= fr[s] <= thj ? 1ull << j : 0ull;   // naive.rs:31 with max_freq = threshold(source j)
                    gives[s][c] |= frj <= th[s] ? 1ull << j : 0ull; // ... and this lane as the source, row j
                }
            }
        }
#pragma unroll
        for (int s = 0; s < RL; s++) {
            unsigned long long out_any = 0;
#pragma unroll
            for (int c = 0; c < RL; c++) {
                const unsigned long long hit = ((unsigned long long)hits[s][2 * c + 1] << 32) | hits[s][2 * c];
                const unsigned long long h_ns = s == c ? hit & ~(1ull << lane) : hit;
                in[s][c] = h_ns & may[s][c];
                out_any |= h_ns & gives[s][c];
            }
            act64[s] = __ballot(out_any != 0ull) & vmask[s];
        }
    }
    bool back = false;
#pragma unroll
    for (int s = 0; s < RL; s++) {
        const int row = lane + 64 * s;
#pragma unroll
        for (int c = 0; c < RL; c++) {
            if (row >= n) in[s][c] = 0ull;
            // an in-edge from a source ranked behind the row: one sweep is then not the fixed point
            back |= (in[s][c] & ~low_bits64(row - 64 * c + 1)) != 0ull;
        }
    }
    uint32_t lab[RL];
#pragma unroll
    for (int s = 0; s < RL; s++) lab[s] = (uint32_t)(lane + 64 * s);
    const bool iterate = __any(back);
    bool changed;
    do { // Gauss-Seidel sweeps in rank order over the active sources until no label moves
        uint32_t before[RL];
#pragma unroll
        for (int s = 0; s < RL; s++) before[s] = lab[s];
#pragma unroll
        for (int h = 0; h < H; h++) {
            uint32_t t = (uint32_t)(act64[h >> 1] >> (32 * (h & 1)));
            while (t) {
                const int jj = __builtin_ctz(t);
                t &= t - 1;
                const uint32_t lj = __builtin_amdgcn_readlane(lab[h >> 1], 32 * (h & 1) + jj);
#pragma unroll
                for (int s = 0; s < RL; s++) {
                    // lj where the row has the edge, all ones where not: one min either way
                    const uint32_t half = (uint32_t)(in[s][h >> 1] >> (32 * (h & 1)));
                    const uint32_t e = (uint32_t)__builtin_amdgcn_sbfe((int)half, (uint32_t)jj, 1u);
                    lab[s] = min(lab[s], lj | ~e);
                }
            }
        }
        changed = false;
#pragma unroll
        for (int s = 0; s < RL; s++) changed |= lab[s] != before[s];
    } while (iterate && __any(changed));
    FusedCounts out;
    out.bad = bad;
    out.kept = 0;
#pragma unroll
    for (int s = 0; s < RL; s++) {
        const int row = lane + 64 * s;
        if (W > 1 && !HAS_N && row < n) out.bad += n_code[s];
        if (row < n) {
            const bool kp = lab[s] == (uint32_t)row;
            if (label) label[start + row] = start + lab[s];
            kept[start + row] = kp ? 1 : 0;
            if (root) root[start + row] = start + lab[s];
            out.kept += kp ? 1u : 0u;
        }
    }
    return out;
}

// Walks the bucket table itself (no task list to build or upload): wave w takes buckets
// w, w + n_waves, ...; buckets with fewer than 2 or more than fused_max entries belong
// to other kernels and are skipped.
// KB >= 0: bit-sliced body with K = KB (k <= 3); KB < 0: the column-walking body (any k).
// W > 1: keys of W words (directional, KB >= 0 only: the host sends nothing else here).
template <bool HAS_N, int MODE, int KB, int W = 1>
__global__ __launch_bounds__(256) void small_bucket_kernel(const uint64_t *__restrict__ keys,
                                                           const uint64_t *__restrict__ nmask,
                                                           const int32_t *__restrict__ freq,
                                                           float percentage,
                                                           const uint64_t *__restrict__ bucket_off,
                                                           uint32_t n_buckets, uint32_t fused_max,
                                                           uint32_t n_entries,
                                                           uint32_t *__restrict__ label,
                                                           uint8_t *__restrict__ kept,
                                                           uint32_t *__restrict__ root, int k,
                                                           int umi_len, int32_t adj_max_freq,
                                                           unsigned long long *__restrict__ counters)
{
    // Positions are dealt statically to the BLOCKS (position b to block b mod the grid) and taken one by
    // one by the block's four waves from a counter in LDS: a position of 65..128 entries costs 2.2x one
    // of up to 64, and with positions dealt to the waves themselves the busiest SIMD of config 3 carried
    // 13 % more than the mean (a counter in global memory instead: DESIGN.md section 5 -- 5-10x slower).
    __shared__ uint32_t next_of_block;
    if (threadIdx.x == 0) next_of_block = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned int bad = 0, n_kept = 0; // per lane; summed over the block at the end
    auto draw = [&]() -> uint32_t {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(&next_of_block, 1u);
        return blockIdx.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)i) * gridDim.x;
    };
    uint32_t b_next = draw();
    while (b_next < n_buckets) {
        const uint32_t b = b_next;
        b_next = draw(); // (asked for before the work: the answer is there when it is wanted)
        const uint64_t s64 = bucket_off[b], e64 = bucket_off[b + 1];
        const uint32_t start = __builtin_amdgcn_readfirstlane((uint32_t)s64);
        const uint32_t end = __builtin_amdgcn_readfirstlane((uint32_t)e64);
        const uint32_t n = end - start;
        // (a table in device memory is the caller's: one that does not match the host's copy -- which
        // is the one that was checked -- must not lead outside the arrays)
        if (e64 < s64 || e64 > (uint64_t)n_entries) {
            bad += lane == 0 ? 1u : 0u;
            continue;
        }
        if (n == 0 || n > fused_max) continue;
        if (n == 1) { // a position with one UMI: it survives
            if (lane == 0) {
                if (label) label[start] = start;
                kept[start] = 1;
                if (root) root[start] = start;
                bad += freq[start] < 1 ? 1u : 0u;
                n_kept += 1;
            }
            continue;
        }
        FusedCounts c;
        if ((KB >= 0 && MODE == MODE_DIRECTIONAL) || W > 1) {
            constexpr int K = KB >= 0 ? KB : 0;
            if (n <= 64)
                c = small_bucket_body_dir<1, HAS_N, K, W>(keys, nmask, freq, percentage, start, (int)n, umi_len, label,
                                                          kept, root);
            else
                c = small_bucket_body_dir<2, HAS_N, K, W>(keys, nmask, freq, percentage, start, (int)n, umi_len, label,
                                                          kept, root);
        } else if (KB >= 0) {
            constexpr int K = KB >= 0 ? KB : 0;
            if (n <= 64)
                c = small_bucket_body_bs<1, HAS_N, MODE, K>(keys, nmask, freq, percentage, start, (int)n,
                                                            umi_len, label, kept, root, adj_max_freq);
            else
                c = small_bucket_body_bs<2, HAS_N, MODE, K>(keys, nmask, freq, percentage, start, (int)n,
                                                            umi_len, label, kept, root, adj_max_freq);
        } else {
            if (n <= 64)
                c = small_bucket_body<1, HAS_N, MODE>(keys, nmask, freq, percentage, start, (int)n, label,
                                                      kept, root, k, adj_max_freq);
            else
                c = small_bucket_body<2, HAS_N, MODE>(keys, nmask, freq, percentage, start, (int)n, label,
                                                      kept, root, k, adj_max_freq);
        }
        bad += c.bad;
        n_kept += c.kept;
    }
    block_count_add(bad, &counters[CNT_ERROR]);
    block_count_add(n_kept, &counters[CNT_KEPT_FUSED]);
}

// ---- collapse: pointer jumps of the one-way rounds --------------------------------
__global__ __launch_bounds__(256) void jump_kernel(uint32_t *label, uint32_t n, uint32_t *changed,
                                                   int round)
{
    if (round > 0 && changed[round - 1] == 0) return;
    bool any = false;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        const uint32_t l = label[v];
        if (l != v) {
            // label[l] reaches l and l reaches v: transitivity keeps the invariant
            const uint32_t ll = label[l];
            if (ll < l) {
                atomicMin(&label[v], ll);
                any = true;
            }
        }
    }
    if (any) changed[round] = 1;
}

// ---- the one-way pairs of the directional collapse ---------------------------------------
// Reachability inside a set of entries joined by symmetric pairs (both directions permitted)
// is symmetric, and a one-way pair always leads to a strictly lower freq (thr is monotone in
// freq), so the one-way pairs form a DAG over those sets.  The sets come from the union-find
// of umihip_collapse.hip (comp[v] = smallest index of v's set); here lab[c] = smallest set index
// that reaches set c is propagated along the one-way pairs (rounds <= depth of the DAG).  Then
// label[v] = lab[comp[v]], the smallest rank that reaches v: what directional.rs:30-54,78-88
// removes v under.
__global__ __launch_bounds__(256) void dag_hook_kernel(const uint2 *__restrict__ edges,
                                                       const unsigned long long *counters,
                                                       uint32_t edge_cap,
                                                       const uint32_t *__restrict__ comp,
                                                       uint32_t *lab, uint32_t *changed, int round)
{
    if (round > 0 && changed[round - 1] == 0) return;
    unsigned long long ne = counters[CNT_EDGES];
    const uint32_t E = ne < edge_cap ? (uint32_t)ne : edge_cap;
    bool any = false;
    HotMin hot;
    // (every lane of a wave makes the same number of trips: the shuffles below need them all)
    for (uint32_t e0 = blockIdx.x * blockDim.x; e0 < E; e0 += gridDim.x * blockDim.x) {
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
        wave_atomic_min(lab, cv, lu, todo, hot); // many one-way pairs point into the giant component
    }
    hot_flush(lab, hot);
    if (any) changed[round] = 1;
}

__global__ __launch_bounds__(256) void map_label_kernel(uint32_t *comp, const uint32_t *__restrict__ lab,
                                                        uint32_t n)
{
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        comp[v] = lab[comp[v]];
}

__global__ __launch_bounds__(256) void finalize_kernel(const uint32_t *__restrict__ label,
                                                       const RangeTask *__restrict__ ranges, uint32_t n,
                                                       uint8_t *__restrict__ kept,
                                                       uint32_t *__restrict__ root,
                                                       unsigned long long *counters)
{
    unsigned int cnt = 0;
    for_entries(ranges, n, [&](uint32_t i) {
        const uint32_t l = label[i];
        const bool kp = l == i;
        kept[i] = kp ? 1 : 0;
        if (root) root[i] = l;
        cnt += kp ? 1u : 0u;
    });
    block_count_add(cnt, &counters[CNT_KEPT]);
}

// ---- adjacency with max_freq > 0: greedy independent set in rank order --------
constexpr uint8_t ST_UNKNOWN = 0, ST_ROOT = 1, ST_REMOVED = 2;

__global__ __launch_bounds__(256) void adj_mark_kernel(const uint2 *__restrict__ edges,
                                                       const unsigned long long *counters,
                                                       uint32_t edge_cap, uint8_t *status,
                                                       uint8_t *blocked, uint32_t *label)
{
    unsigned long long ne = counters[CNT_EDGES];
    const uint32_t E = ne < edge_cap ? (uint32_t)ne : edge_cap;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const uint2 uv = edges[e]; // u < v
        const uint8_t su = status[uv.x];
        if (su == ST_ROOT) {
            status[uv.y] = ST_REMOVED; // the first root in rank order is the one that removes it
            atomicMin(&label[uv.y], uv.x);
        } else if (su == ST_UNKNOWN) {
            blocked[uv.y] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void adj_promote_kernel(uint8_t *status, uint8_t *blocked,
                                                          uint32_t n,
                                                          unsigned long long *counters)
{
    unsigned int unk = 0;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        if (status[v] == ST_UNKNOWN) {
            if (!blocked[v]) status[v] = ST_ROOT; else unk++;
        }
        blocked[v] = 0;
    }
    block_count_add(unk, &counters[CNT_UNKNOWN]);
}

__global__ __launch_bounds__(256) void adj_finalize_kernel(const uint8_t *__restrict__ status,
                                                           const uint32_t *__restrict__ label,
                                                           const RangeTask *__restrict__ ranges, uint32_t n,
                                                           uint8_t *__restrict__ kept,
                                                           uint32_t *__restrict__ root,
                                                           unsigned long long *counters)
{
    unsigned int cnt = 0;
    for_entries(ranges, n, [&](uint32_t i) {
        const bool kp = status[i] == ST_ROOT;
        kept[i] = kp ? 1 : 0;
        if (root) root[i] = kp ? i : label[i];
        cnt += kp ? 1u : 0u;
    });
    block_count_add(cnt, &counters[CNT_KEPT]);
}

inline uint32_t grid_for(uint64_t work, int block, uint32_t cap = 2048)
{
    uint64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

} // namespace

// ranges == nullptr: all n entries; else n_ranges chunks of entries (one block each)
hipError_t launch_prep(const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                       const uint64_t *bucket_off, uint64_t n_buckets, const RangeTask *ranges,
                       uint32_t n_ranges, uint32_t n, uint32_t fused_max, int umi_len, float percentage,
                       bool key32, void *fkey, int32_t *thr, uint32_t *label, uint32_t *lab,
                       unsigned long long *counters, const SegDesc *segs, int n_seg_parts,
                       uint32_t *bin_cnt, hipStream_t s, bool skip_seg, bool entries_too, uint32_t seg_from,
                       int key_words, int full_umi_len)
{
    if (n == 0 || (ranges && n_ranges == 0)) return hipSuccess;
    if (!entries_too) return hipSuccess; // every range is a segment's: its count kernel does it all
    prep_kernel<<<ranges ? n_ranges : grid_for(n, 256), 256, 0, s>>>(keys, nmask, freq, ranges, n, umi_len,
                                                                     percentage, key32 ? 1 : 0, fkey, thr,
                                                                     label, lab, counters, segs, n_seg_parts,
                                                                     bin_cnt, skip_seg ? 1 : 0, key_words, full_umi_len);
    bucket_rise_kernel<<<grid_for(n_buckets, 256, 512), 256, 0, s>>>(freq, bucket_off, n_buckets,
                                                                     ranges ? fused_max : 0u,
                                                                     skip_seg ? seg_from : 0xFFFFFFFFu, n, counters);
    return hipGetLastError();
}

hipError_t launch_pairs(const PairArgs &a, uint32_t n_tasks, bool big, bool key32, hipStream_t s)
{
    if (n_tasks == 0) return hipSuccess;
    if (big) {
        if (key32)
            pair_kernel<uint32_t, BIG_THREADS, BIG_RPT><<<n_tasks, BIG_THREADS, 0, s>>>(a);
        else
            pair_kernel<uint64_t, BIG_THREADS, BIG_RPT><<<n_tasks, BIG_THREADS, 0, s>>>(a);
    } else {
        if (key32)
            pair_kernel<uint32_t, SMALL_THREADS, SMALL_RPT><<<n_tasks, SMALL_THREADS, 0, s>>>(a);
        else
            pair_kernel<uint64_t, SMALL_THREADS, SMALL_RPT><<<n_tasks, SMALL_THREADS, 0, s>>>(a);
    }
    return hipGetLastError();
}

namespace {
template <bool HAS_N, int MODE>
void launch_small_k(int kb, uint32_t blocks, const uint64_t *keys, const uint64_t *nmask,
                    const int32_t *freq, float percentage, const uint64_t *bucket_off,
                    uint32_t n_buckets, uint32_t fused_max, uint32_t n_entries, uint32_t *label, uint8_t *kept,
                    uint32_t *root, int k, int umi_len, int32_t adj_max_freq,
                    unsigned long long *counters, hipStream_t s)
{
#define UMI_LAUNCH_SMALL(KB)                                                                          \
    small_bucket_kernel<HAS_N, MODE, KB><<<blocks, 256, 0, s>>>(keys, nmask, freq, percentage, bucket_off, \
                                                                n_buckets, fused_max, n_entries, label, kept, root, k, \
                                                                umi_len, adj_max_freq, counters)
    switch (kb) {
    case 0: UMI_LAUNCH_SMALL(0); break;
    case 1: UMI_LAUNCH_SMALL(1); break;
    case 2: UMI_LAUNCH_SMALL(2); break;
    case 3: UMI_LAUNCH_SMALL(3); break;
    default: UMI_LAUNCH_SMALL(-1); break;
    }
#undef UMI_LAUNCH_SMALL
}
} // namespace

// The same for keys of n_words = 2..4 words (umi_len 22..85), directional, k <= 3.
hipError_t launch_small_buckets_wide(const uint64_t *keys, const uint64_t *nmask, int n_words, const int32_t *freq,
                                     float percentage, const uint64_t *bucket_off, uint32_t n_buckets,
                                     uint32_t fused_max, uint32_t n_entries, uint8_t *kept, uint32_t *root, int k,
                                     int umi_len, unsigned long long *counters, uint32_t max_blocks, hipStream_t s)
{
    if (n_buckets == 0 || fused_max < 1) return hipSuccess;
    if (k < 0 || k > 3 || n_words < 2 || n_words > 4) return hipErrorInvalidValue;
    const uint32_t blocks = grid_for((uint64_t)n_buckets * 64, 256, max_blocks);
#define UMI_LAUNCH_WIDE(HN, KB, WN)                                                                                \
    small_bucket_kernel<HN, MODE_DIRECTIONAL, KB, WN><<<blocks, 256, 0, s>>>(keys, nmask, freq, percentage, bucket_off, \
                                                                             n_buckets, fused_max, n_entries, nullptr,  \
                                                                             kept, root, k, umi_len, 0, counters)
#define UMI_LAUNCH_WIDE_K(HN, WN)                   \
    switch (k) {                                    \
    case 0: UMI_LAUNCH_WIDE(HN, 0, WN); break;      \
    case 1: UMI_LAUNCH_WIDE(HN, 1, WN); break;      \
    case 2: UMI_LAUNCH_WIDE(HN, 2, WN); break;      \
    default: UMI_LAUNCH_WIDE(HN, 3, WN); break;     \
    }
#define UMI_LAUNCH_WIDE_W(HN)                       \
    switch (n_words) {                              \
    case 2: UMI_LAUNCH_WIDE_K(HN, 2); break;        \
    case 3: UMI_LAUNCH_WIDE_K(HN, 3); break;        \
    default: UMI_LAUNCH_WIDE_K(HN, 4); break;       \
    }
    if (nmask) { UMI_LAUNCH_WIDE_W(true) } else { UMI_LAUNCH_WIDE_W(false) }
#undef UMI_LAUNCH_WIDE_W
#undef UMI_LAUNCH_WIDE_K
#undef UMI_LAUNCH_WIDE
    return hipGetLastError();
}

// Every bucket of at most fused_max entries, start to finish: contract check, thresholds,
// label, kept mask, root, survivor count.
hipError_t launch_small_buckets(const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                                float percentage, const uint64_t *bucket_off, uint32_t n_buckets,
                                uint32_t fused_max, uint32_t n_entries, uint32_t *label, uint8_t *kept, uint32_t *root,
                                int k, int umi_len, bool sliced, int mode, int32_t adj_max_freq,
                                unsigned long long *counters, uint32_t max_blocks, hipStream_t s)
{
    if (n_buckets == 0 || fused_max < 1) return hipSuccess;
    // The blocks walk the bucket table with the stride of the grid, a block's waves take its positions
    // one by one.  Measured on 10^5 positions of ~60 UMIs (config 3), blocks per CU: kernel us -- 8: 103,
    // 12: 100, 16: 96.5, 20: 96.4, 24: 98, 32: 125, 48: 172 (a CU holds 8 at once: more, smaller shares
    // even out what the positions' unequal costs leave, many more are launches of their own); with
    // positions dealt to the waves themselves it was 8: 104, 12: 98, 16: 98, 24: 123.  max_blocks comes
    // from the context ("fused_blocks" per CU, default 20).
    const uint32_t blocks = grid_for((uint64_t)n_buckets * 64, 256, max_blocks);
    const int kb = (sliced && k >= 0 && k <= 3) ? k : -1;
    if (mode == MODE_DIRECTIONAL) {
        if (nmask) launch_small_k<true, MODE_DIRECTIONAL>(kb, blocks, keys, nmask, freq, percentage, bucket_off, n_buckets, fused_max, n_entries, label, kept, root, k, umi_len, adj_max_freq, counters, s);
        else launch_small_k<false, MODE_DIRECTIONAL>(kb, blocks, keys, nmask, freq, percentage, bucket_off, n_buckets, fused_max, n_entries, label, kept, root, k, umi_len, adj_max_freq, counters, s);
    } else {
        if (nmask) launch_small_k<true, MODE_ADJACENCY>(kb, blocks, keys, nmask, freq, percentage, bucket_off, n_buckets, fused_max, n_entries, label, kept, root, k, umi_len, adj_max_freq, counters, s);
        else launch_small_k<false, MODE_ADJACENCY>(kb, blocks, keys, nmask, freq, percentage, bucket_off, n_buckets, fused_max, n_entries, label, kept, root, k, umi_len, adj_max_freq, counters, s);
    }
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void iota_label_kernel(uint32_t *label, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        label[i] = i;
}
} // namespace

hipError_t launch_iota(uint32_t *label, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    iota_label_kernel<<<grid_for(n, 256), 256, 0, s>>>(label, n);
    return hipGetLastError();
}

hipError_t launch_dag_round(const uint2 *edges, const unsigned long long *counters, uint32_t edge_cap,
                            const uint32_t *comp, uint32_t *lab, uint32_t n, uint32_t *changed,
                            int round, uint32_t n_edges_hint, hipStream_t s)
{
    dag_hook_kernel<<<grid_for(n_edges_hint, 256, 512), 256, 0, s>>>(edges, counters, edge_cap, comp, lab,
                                                                changed, round);
    jump_kernel<<<grid_for(n, 256), 256, 0, s>>>(lab, n, changed, round);
    return hipGetLastError();
}

hipError_t launch_map_labels(uint32_t *comp, const uint32_t *lab, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    map_label_kernel<<<grid_for(n, 256), 256, 0, s>>>(comp, lab, n);
    return hipGetLastError();
}

hipError_t launch_finalize(const uint32_t *label, const RangeTask *ranges, uint32_t n_ranges, uint32_t n,
                           uint8_t *kept, uint32_t *root, unsigned long long *counters, hipStream_t s)
{
    if (n == 0 || (ranges && n_ranges == 0)) return hipSuccess;
    finalize_kernel<<<ranges ? n_ranges : grid_for(n, 1024, 1024), 256, 0, s>>>(label, ranges, n, kept, root,
                                                                                counters);
    return hipGetLastError();
}

hipError_t launch_adj_iter(const uint2 *edges, const unsigned long long *counters,
                           uint32_t edge_cap, uint8_t *status, uint8_t *blocked, uint32_t *label,
                           uint32_t n, unsigned long long *counters_rw, uint32_t n_edges_hint,
                           hipStream_t s)
{
    adj_mark_kernel<<<grid_for(n_edges_hint, 256), 256, 0, s>>>(edges, counters, edge_cap, status,
                                                                blocked, label);
    adj_promote_kernel<<<grid_for(n, 1024, 1024), 256, 0, s>>>(status, blocked, n, counters_rw);
    return hipGetLastError();
}

hipError_t launch_adj_finalize(const uint8_t *status, const uint32_t *label, const RangeTask *ranges,
                               uint32_t n_ranges, uint32_t n, uint8_t *kept, uint32_t *root,
                               unsigned long long *counters, hipStream_t s)
{
    if (n == 0 || (ranges && n_ranges == 0)) return hipSuccess;
    adj_finalize_kernel<<<ranges ? n_ranges : grid_for(n, 1024, 1024), 256, 0, s>>>(status, label, ranges, n,
                                                                                    kept, root, counters);
    return hipGetLastError();
}

} // namespace umihip
