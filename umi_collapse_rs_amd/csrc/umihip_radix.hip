// Sort and scan primitives of the read staging (gfx950): a stable least-significant-digit radix
// sort of (64-bit key, 32-bit value) pairs with one read and one write of the data per 8-bit digit
// -- tiles chained by a decoupled look-back, as in the segment index's bin scan -- and an inclusive
// scan of 64-bit words (two 32-bit counters ride in one word).
//
// What they serve in the reference (tkob-vh/umi-collapse-rs): the two HashMaps of
// DeduplicateSAM::deduplicate_and_merge (src/deduplicate_sam.rs:88-89,148-176) group the reads by
// alignment and by UMI; here the reads are sorted by (alignment key, UMI) instead and equal
// neighbours are one entry (umihip_stage.hip).  Integer / byte work on HBM streams, no MFMA.
//
// One pass over a tile of 4,096 pairs (256 threads, 16 rounds of 64 pairs per wave):
//   rank   every key's position among the keys of its digit inside its wave's 1,024 pairs, in
//          order: the lanes of a round that share a digit find each other with eight ballots (one
//          per bit of the digit), the wave keeps a running count per digit in LDS;
//   chain  the tile's 256 counts are published, thread d adds up digit d's counts of the tiles before
//          (stopping at the first tile that knows its own prefix) and publishes the tile's prefix --
//          tiles draw their number from a ticket counter, so everything before a tile is running;
//   move   the pairs go to LDS in digit order and from there to their places: a digit's pairs of
//          one tile are neighbours in the output, the stores of a wave fill whole lines.
// The digit histograms of all passes -- what tile 0's prefix starts from -- come from one sweep over
// the keys in front of the first pass.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "umihip_internal.h"

namespace umihip {

namespace {

#ifndef RS_THREADS_CFG
#define RS_THREADS_CFG 1024
#define RS_ROUNDS_CFG 4
#endif
constexpr int RS_THREADS = RS_THREADS_CFG, RS_ROUNDS = RS_ROUNDS_CFG, RS_TILE = RS_THREADS * RS_ROUNDS, RS_RADIX = 256,
              RS_WAVES = RS_THREADS / 64;
constexpr int RS_MAX_PASSES = RADIX_MAX_PASSES;
static_assert(RS_RADIX == RADIX_BINS, "8-bit digits");
// a tile's word per digit: two flag bits over a 30-bit count (n < 2^30, checked by the caller)
constexpr uint32_t RS_HAVE = 1u << 30, RS_PREFIX = 1u << 31, RS_COUNT = (1u << 30) - 1u;
static_assert(RS_THREADS >= RS_RADIX && RS_THREADS % 64 == 0, "a thread per digit");

template <typename KeyT> struct RadixPass {
    const KeyT *kin;
    KeyT *kout;
    const uint32_t *vin;
    uint32_t *vout;
    uint32_t n;
    int shift;
    uint32_t mask;
    const uint32_t *digit_base; // [256] keys of the whole input with a smaller digit
    uint32_t *status;           // [tiles][256], zero before the pass
    uint32_t *ticket;           // zero before the pass
};

// digit histograms of all passes in one sweep: every block writes its own counts, parts[block][pass][digit]
// (a producer of the keys may do this itself, umihip_internal.h).  No atomics: 2,048 blocks adding
// their 1,536 counts to one table were 3 million memory-side atomics, 0.08 ms at 10^7 keys.
template <typename KeyT>
__global__ __launch_bounds__(256) void radix_hist_kernel(const KeyT *__restrict__ keys, uint32_t n, int begin_bit,
                                                         int n_passes, int end_bit, uint32_t *__restrict__ parts)
{
    __shared__ uint32_t h[RS_MAX_PASSES][RS_RADIX];
    for (int i = threadIdx.x; i < n_passes * RS_RADIX; i += blockDim.x) (&h[0][0])[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        for (int p = 0; p < n_passes; p++) {
            const int sh = begin_bit + 8 * p, bits = std::min(8, end_bit - sh);
            atomicAdd(&h[p][(uint32_t)(k >> sh) & ((1u << bits) - 1u)], 1u);
        }
    }
    __syncthreads();
    uint32_t *mine = parts + (size_t)blockIdx.x * n_passes * RS_RADIX;
    for (int i = threadIdx.x; i < n_passes * RS_RADIX; i += blockDim.x) mine[i] = (&h[0][0])[i];
}

// the parts added up: block (pass, j) takes every 16th part from j on, four at a time, and adds its
// 256 sums to hist[pass][] (zero before).  (One block per pass walking all 2,048 parts took 86 us:
// 512 loads in a row per thread.)
constexpr int RS_SUM_BLOCKS = 16;
__global__ __launch_bounds__(1024) void radix_sum_kernel(const uint32_t *__restrict__ parts, uint32_t n_parts, int n_passes,
                                                         uint32_t *__restrict__ hist)
{
    __shared__ uint32_t grp[4][RS_RADIX];
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8, pass = blockIdx.x / RS_SUM_BLOCKS, j = blockIdx.x % RS_SUM_BLOCKS;
    uint32_t c = 0;
#pragma unroll 4
    for (uint32_t part = j * 4 + q; part < n_parts; part += 4 * RS_SUM_BLOCKS)
        c += parts[((size_t)part * n_passes + pass) * RS_RADIX + d];
    grp[q][d] = c;
    __syncthreads();
    if (threadIdx.x < RS_RADIX) {
        const uint32_t v = grp[0][d] + grp[1][d] + grp[2][d] + grp[3][d];
        if (v) atomicAdd(&hist[(size_t)pass * RS_RADIX + d], v);
    }
}

// exclusive scan of a pass's 256 bins, in place (one block per pass)
__global__ __launch_bounds__(RS_RADIX) void radix_base_kernel(uint32_t *__restrict__ hist)
{
    __shared__ uint32_t wsum[RS_RADIX / 64];
    uint32_t *h = hist + (size_t)blockIdx.x * RS_RADIX;
    const uint32_t v = h[threadIdx.x];
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += wsum[w];
    h[threadIdx.x] = off + incl - v;
}

template <typename KeyT> __global__ __launch_bounds__(RS_THREADS) void radix_onesweep_kernel(RadixPass<KeyT> a)
{
    __shared__ uint32_t s_tile;
    __shared__ uint32_t cnt[RS_WAVES][RS_RADIX]; // per wave: running count of every digit; then the place of the wave's
                                                 // first pair of the digit in the tile's digit order
    __shared__ uint32_t gbase[RS_RADIX];         // output position of the tile's first pair of digit d, less its place in the tile
    __shared__ uint32_t wtot[RS_RADIX / 64];
    __shared__ KeyT skeys[RS_TILE];
    __shared__ uint32_t svals[RS_TILE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
    for (int i = tid; i < RS_WAVES * RS_RADIX; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t base = (uint64_t)tile * RS_TILE;
    const uint32_t tile_n = (uint32_t)std::min<uint64_t>(RS_TILE, a.n - base);
    // ---- rank: the wave's pairs in rounds of 64 consecutive ones
    KeyT key[RS_ROUNDS];
    uint32_t val[RS_ROUNDS], rank[RS_ROUNDS];
    volatile uint32_t *my_cnt = cnt[wave];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const uint32_t at = (uint32_t)wave * (RS_ROUNDS * 64) + (uint32_t)r * 64 + (uint32_t)lane;
        const bool valid = at < tile_n;
        key[r] = valid ? a.kin[base + at] : (KeyT)~(KeyT)0;
        val[r] = valid ? a.vin[base + at] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const uint32_t at = (uint32_t)wave * (RS_ROUNDS * 64) + (uint32_t)r * 64 + (uint32_t)lane;
        const bool valid = at < tile_n;
        const uint32_t d = (uint32_t)(key[r] >> a.shift) & a.mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t pre = my_cnt[d];
        rank[r] = pre + (uint32_t)__builtin_popcountll(peers & lt);
        if (valid && (peers & lt) == 0ull) my_cnt[d] = pre + (uint32_t)__builtin_popcountll(peers); // (the first of its peers)
    }
    __syncthreads();
    // ---- chain: thread d owns digit d
    uint32_t tot = 0, incl = 0;
    if (tid < RS_RADIX) {
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) {
            const uint32_t c = cnt[w][tid];
            cnt[w][tid] = tot; // the wave's offset inside the tile's run of digit d
            tot += c;
        }
        incl = tot; // exclusive scan of tot over the digits: where the tile's run of digit d starts
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wtot[wave] = incl;
    }
    __syncthreads();
    if (tid < RS_RADIX) {
        const int d = tid;
        uint32_t off = 0;
        for (int w = 0; w < wave; w++) off += wtot[w];
        const uint32_t start = off + incl - tot;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) cnt[w][d] += start;
        uint32_t *st = a.status + (size_t)tile * RS_RADIX + d;
        uint32_t before = 0;
        if (tile == 0) {
            __hip_atomic_store(st, RS_HAVE | RS_PREFIX | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(st, RS_HAVE | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (tile t drew its ticket before this one: it is running.)  Eight tiles' words are asked for
            // at once: a trip to the memory side per tile would make the tiles that start together wait
            // for one another in a row
#ifndef RS_LB_CFG
#define RS_LB_CFG 8
#endif
            constexpr int LB = RS_LB_CFG;
            bool done = false;
            for (int64_t t = (int64_t)tile - 1; t >= 0 && !done; t -= LB) {
                uint32_t v[LB];
#pragma unroll
                for (int q = 0; q < LB; q++)
                    v[q] = t - q >= 0 ? __hip_atomic_load(a.status + (size_t)(t - q) * RS_RADIX + d, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT)
                                      : (RS_HAVE | RS_PREFIX);
#pragma unroll
                for (int q = 0; q < LB; q++) {
                    if (!done) {
                        while (!(v[q] & RS_HAVE))
                            v[q] = __hip_atomic_load(a.status + (size_t)(t - q) * RS_RADIX + d, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                        before += v[q] & RS_COUNT;
                        done = (v[q] & RS_PREFIX) != 0;
                    }
                }
            }
            __hip_atomic_store(st, RS_HAVE | RS_PREFIX | (before + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gbase[d] = a.digit_base[d] + before - start;
    }
    __syncthreads();
    // ---- move: to LDS in digit order, then out in runs
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const uint32_t at = (uint32_t)wave * (RS_ROUNDS * 64) + (uint32_t)r * 64 + (uint32_t)lane;
        if (at < tile_n) {
            const uint32_t d = (uint32_t)(key[r] >> a.shift) & a.mask;
            const uint32_t pos = cnt[wave][d] + rank[r];
            skeys[pos] = key[r];
            svals[pos] = val[r];
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < tile_n; i += RS_THREADS) {
        const KeyT k = skeys[i];
        const uint32_t d = (uint32_t)(k >> a.shift) & a.mask;
        const uint32_t pos = gbase[d] + i;
        a.kout[pos] = k;
        a.vout[pos] = svals[i];
    }
}

// ---- inclusive scan of 64-bit words: tile sums | scan of the sums (one block) | apply.  Three
// launches and two reads of the data, but no chain: a single-pass scan with a decoupled look-back was
// built first and took 2.3 ms for 10^7 words -- 4,883 tiles start together, and the walk back over
// tiles whose prefix is not there yet costs a trip to the memory side per tile.
constexpr int SC_THREADS = 256, SC_ITEMS = 8, SC_TILE = SC_THREADS * SC_ITEMS;
__global__ __launch_bounds__(SC_THREADS) void scan_sums_kernel(const uint64_t *__restrict__ in, uint32_t n,
                                                               unsigned long long *__restrict__ sums)
{
    __shared__ uint64_t wsum[SC_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)threadIdx.x * SC_ITEMS;
    uint64_t mine = 0;
#pragma unroll
    for (int q = 0; q < SC_ITEMS; q++) mine += base + q < n ? in[base + q] : 0ull;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < SC_THREADS / 64; w++) t += wsum[w];
        sums[blockIdx.x] = t;
    }
}
// exclusive scan of the tile sums, in place (one block of 1024 threads walks them)
__global__ __launch_bounds__(1024) void scan_spine_kernel(unsigned long long *sums, uint32_t n_tiles)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n_tiles ? sums[i] : 0ull;
        uint64_t incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t up = __shfl_up(incl, o);
            if ((int)(threadIdx.x & 63) >= o) incl += up;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t off = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += wsum[w];
        if (i < n_tiles) sums[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = off + incl;
        __syncthreads();
    }
}
__global__ __launch_bounds__(SC_THREADS) void scan_apply_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                                uint32_t n, const unsigned long long *__restrict__ sums)
{
    __shared__ uint64_t wsum[SC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)tid * SC_ITEMS;
    uint64_t v[SC_ITEMS], mine = 0;
#pragma unroll
    for (int q = 0; q < SC_ITEMS; q++) {
        v[q] = base + q < n ? in[base + q] : 0ull;
        mine += v[q];
    }
    uint64_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t off = sums[blockIdx.x];
    for (int w = 0; w < wave; w++) off += wsum[w];
    uint64_t run = off + incl - mine;
#pragma unroll
    for (int q = 0; q < SC_ITEMS; q++) {
        run += v[q];
        if (base + q < n) out[base + q] = run;
    }
}

inline uint32_t tiles_of(uint32_t n, uint32_t tile) { return std::max(1u, (uint32_t)(((uint64_t)n + tile - 1) / tile)); }

struct RadixTemp {
    uint32_t *hist;    // [RS_MAX_PASSES][256]
    uint32_t *tickets; // [RS_MAX_PASSES + 1]
    uint32_t *status;  // [passes][tiles][256]
    uint32_t *parts;   // [RADIX_HIST_PARTS][passes][256]: written whole by whoever counts, not zeroed
    size_t zero_bytes, total;
};
RadixTemp radix_carve(void *temp, uint32_t n, int passes)
{
    RadixTemp t;
    char *p = (char *)temp;
    size_t off = 0;
    t.hist = (uint32_t *)(p + off);
    off += (size_t)RS_MAX_PASSES * RS_RADIX * 4;
    t.tickets = (uint32_t *)(p + off);
    off += 64;
    t.status = (uint32_t *)(p + off);
    off += (size_t)passes * tiles_of(n, RS_TILE) * RS_RADIX * 4;
    t.zero_bytes = off = (off + 255) & ~(size_t)255;
    t.parts = (uint32_t *)(p + off);
    off += (size_t)RADIX_HIST_PARTS * RS_MAX_PASSES * RS_RADIX * 4;
    t.total = off;
    return t;
}

template <typename KeyT>
hipError_t radix_sort_impl(KeyT *keys_a, KeyT *keys_b, uint32_t *vals_a, uint32_t *vals_b, uint32_t n, int begin_bit,
                           int end_bit, void *temp, size_t temp_bytes, uint32_t hist_parts, bool *result_in_b, hipStream_t s)
{
    *result_in_b = false;
    if (n == 0 || end_bit <= begin_bit) return hipSuccess;
    if (begin_bit < 0 || end_bit > (int)sizeof(KeyT) * 8 || n > RS_COUNT) return hipErrorInvalidValue;
    const int passes = (end_bit - begin_bit + 7) / 8;
    const RadixTemp t = radix_carve(temp, n, passes);
    if (t.total > temp_bytes) return hipErrorInvalidValue;
    const uint32_t tiles = tiles_of(n, RS_TILE);
    if (hist_parts > (uint32_t)RADIX_HIST_PARTS) return hipErrorInvalidValue;
    if (!hist_parts) {
        const hipError_t e = hipMemsetAsync(temp, 0, t.zero_bytes, s);
        if (e != hipSuccess) return e;
        hist_parts = std::min(tiles_of(n, 256 * 16), (uint32_t)RADIX_HIST_PARTS);
        radix_hist_kernel<KeyT><<<hist_parts, 256, 0, s>>>(keys_a, n, begin_bit, passes, end_bit, t.parts);
    }
    radix_sum_kernel<<<passes * RS_SUM_BLOCKS, 1024, 0, s>>>(t.parts, hist_parts, passes, t.hist);
    radix_base_kernel<<<passes, RS_RADIX, 0, s>>>(t.hist);
    KeyT *kin = keys_a, *kout = keys_b;
    uint32_t *vin = vals_a, *vout = vals_b;
    for (int p = 0; p < passes; p++) {
        RadixPass<KeyT> a;
        a.kin = kin;
        a.kout = kout;
        a.vin = vin;
        a.vout = vout;
        a.n = n;
        a.shift = begin_bit + 8 * p;
        a.mask = (1u << std::min(8, end_bit - a.shift)) - 1u;
        a.digit_base = t.hist + (size_t)p * RS_RADIX;
        a.status = t.status + (size_t)p * tiles * RS_RADIX;
        a.ticket = t.tickets + p;
        radix_onesweep_kernel<KeyT><<<tiles, RS_THREADS, 0, s>>>(a);
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    *result_in_b = (passes & 1) != 0;
    return hipGetLastError();
}

} // namespace

size_t radix_sort_temp_bytes(uint32_t n) { return radix_carve(nullptr, n, RS_MAX_PASSES).total; }

// The digit counts may come from the kernel that produces the keys: radix_sort_prepare zeroes the
// temporaries and returns where that kernel's blocks put their counts (umihip_internal.h).
hipError_t radix_sort_prepare(void *temp, size_t temp_bytes, uint32_t n, int begin_bit, int end_bit, uint32_t **hist_parts,
                              hipStream_t s)
{
    *hist_parts = nullptr;
    if (n == 0 || end_bit <= begin_bit) return hipSuccess;
    const RadixTemp t = radix_carve(temp, n, (end_bit - begin_bit + 7) / 8);
    if (t.total > temp_bytes) return hipErrorInvalidValue;
    *hist_parts = t.parts;
    return hipMemsetAsync(temp, 0, t.zero_bytes, s);
}

hipError_t radix_sort_pairs_u64(uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, uint32_t n,
                                int begin_bit, int end_bit, void *temp, size_t temp_bytes, bool *result_in_b, hipStream_t s,
                                uint32_t hist_parts)
{
    return radix_sort_impl<uint64_t>(keys_a, keys_b, vals_a, vals_b, n, begin_bit, end_bit, temp, temp_bytes, hist_parts,
                                     result_in_b, s);
}
hipError_t radix_sort_pairs_u32(uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, uint32_t n,
                                int begin_bit, int end_bit, void *temp, size_t temp_bytes, bool *result_in_b, hipStream_t s,
                                uint32_t hist_parts)
{
    return radix_sort_impl<uint32_t>(keys_a, keys_b, vals_a, vals_b, n, begin_bit, end_bit, temp, temp_bytes, hist_parts,
                                     result_in_b, s);
}

size_t scan_temp_bytes(uint32_t n) { return ((size_t)tiles_of(n, SC_TILE) * 8 + 255) & ~(size_t)255; }

hipError_t scan_spine_u64(unsigned long long *sums, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    scan_spine_kernel<<<1, 1024, 0, s>>>(sums, n);
    return hipGetLastError();
}

hipError_t scan_inclusive_u64(const uint64_t *in, uint64_t *out, uint32_t n, void *temp, size_t temp_bytes, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint32_t tiles = tiles_of(n, SC_TILE);
    if (scan_temp_bytes(n) > temp_bytes) return hipErrorInvalidValue;
    unsigned long long *sums = (unsigned long long *)temp;
    scan_sums_kernel<<<tiles, SC_THREADS, 0, s>>>(in, n, sums);
    scan_spine_kernel<<<1, 1024, 0, s>>>(sums, tiles);
    scan_apply_kernel<<<tiles, SC_THREADS, 0, s>>>(in, out, n, sums);
    return hipGetLastError();
}

} // namespace umihip
