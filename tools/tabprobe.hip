// Which part of the table kernel's column step makes its waves wait?  The fast path of
// bs_tab_kernel<12,1,2,2> (keys from LDS -> v_readfirstlane -> index window -> 4 bitop3, one
// hit test per 4 columns) on synthetic data, with pieces removed one at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
#define BITOP3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
constexpr int NCOLS = 1 << 15;
constexpr int TILE = 2048;

// MODE 0 full; 1 no hit test; 2 indices from the loop counter (no LDS key, no readfirstlane);
// 3 no index window (plain moves from fixed table entries); 4 window only, no bitop3;
// 5 full but keys fetched two groups ahead
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t *in, uint32_t *out, unsigned long long *clk, int waves_pad)
{
    __shared__ uint32_t ckey[TILE + 16];
    __shared__ uint32_t pad[8192]; // 32 KB: with waves_pad the block count per CU is set from the host
    if (waves_pad == 12345) pad[threadIdx.x] = 1;
    const int tid = threadIdx.x;
    u32x16 t00, t01, t10, t11, t02, t12;
    for (int i = 0; i < 16; i++) {
        t00[i] = in[(tid + i * 3) & 8191];
        t01[i] = in[(tid + i * 5 + 1) & 8191];
        t10[i] = in[(tid + i * 7 + 2) & 8191];
        t11[i] = in[(tid + i * 11 + 3) & 8191];
        t02[i] = in[(tid + i * 13 + 4) & 8191];
        t12[i] = in[(tid + i * 17 + 5) & 8191];
    }
    uint32_t anyP0 = in[tid & 8191], twoP0 = in[(tid + 9) & 8191], anyP1 = in[(tid + 17) & 8191], twoP1 = in[(tid + 31) & 8191];
    const uint32_t valid = 0xFFFFFFFFu;
    uint32_t acc = 0, hits = 0;
    unsigned long long t0 = 0, r0 = 0;
    for (int c0 = 0; c0 < NCOLS; c0 += TILE) {
        __syncthreads();
        for (int cc = tid; cc < TILE + 16; cc += 256) ckey[cc] = in[(c0 + cc) & 8191];
        __syncthreads();
        if (c0 == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
        uint32_t next[4];
        for (int i = 0; i < 4; i++) next[i] = ckey[i];
        for (int c = 0; c < TILE; c += 4) {
            uint32_t key[4], h[4][2];
            for (int i = 0; i < 4; i++) key[i] = MODE == 2 ? (uint32_t)(c * 37 + i * 11) : __builtin_amdgcn_readfirstlane(next[i]);
            if (MODE != 2)
                for (int i = 0; i < 4; i++) next[i] = ckey[c + 4 + i];
            uint32_t anyhit = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t e0, e1, f0, f1;
                const uint32_t i0 = key[i] & 15u, i1 = (key[i] >> 4) & 15u;
                if (MODE == 8) {
                    uint32_t ta, tb;
                    asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\t"
                                 "v_mov_b32 %0, v64\n\t"
                                 "v_mov_b32 %1, v96\n\t"
                                 "s_set_gpr_idx_idx %5\n\t"
                                 "v_bitop3_b32 %2, v80, %6, %0 bitop3:0xe8\n\t"
                                 "v_bitop3_b32 %3, v112, %7, %1 bitop3:0xe8\n\t"
                                 "s_set_gpr_idx_off"
                                 : "=&v"(e0), "=&v"(e1), "=&v"(ta), "=&v"(tb)
                                 : "s"(i0), "s"(i1), "v"(anyP0), "v"(anyP1), "{v[64:79]}"(t00), "{v[80:95]}"(t01),
                                   "{v[96:111]}"(t10), "{v[112:127]}"(t11));
                    h[i][0] = BITOP3(twoP0, ta, valid, 0x02);
                    h[i][1] = BITOP3(twoP1, tb, valid, 0x02);
                    anyhit |= h[i][0] | h[i][1];
                    continue;
                }
                if (MODE == 6) {
                    uint32_t g0, g1;
                    const uint32_t i2 = (key[i] >> 8) & 15u;
                    asm volatile("s_set_gpr_idx_on %6, gpr_idx(SRC0)\n\t"
                                 "v_mov_b32 %0, v64\n\t"
                                 "v_mov_b32 %1, v112\n\t"
                                 "s_set_gpr_idx_idx %7\n\t"
                                 "v_mov_b32 %2, v80\n\t"
                                 "v_mov_b32 %3, v128\n\t"
                                 "s_set_gpr_idx_idx %8\n\t"
                                 "v_mov_b32 %4, v96\n\t"
                                 "v_mov_b32 %5, v144\n\t"
                                 "s_set_gpr_idx_off"
                                 : "=&v"(e0), "=&v"(e1), "=&v"(f0), "=&v"(f1), "=&v"(g0), "=&v"(g1)
                                 : "s"(i0), "s"(i1), "s"(i2), "{v[64:79]}"(t00), "{v[80:95]}"(t01), "{v[96:111]}"(t02),
                                   "{v[112:127]}"(t10), "{v[128:143]}"(t11), "{v[144:159]}"(t12));
                    const uint32_t anyA = BITOP3(e0, f0, g0, 0xfe), twoA = BITOP3(e0, f0, g0, 0xe8);
                    const uint32_t ta = BITOP3(twoA, anyP0, anyA, 0xf8);
                    h[i][0] = BITOP3(twoP0, ta, valid, 0x02);
                    const uint32_t anyB = BITOP3(e1, f1, g1, 0xfe), twoB = BITOP3(e1, f1, g1, 0xe8);
                    const uint32_t tb = BITOP3(twoB, anyP1, anyB, 0xf8);
                    h[i][1] = BITOP3(twoP1, tb, valid, 0x02);
                    anyhit |= h[i][0] | h[i][1];
                    continue;
                }
                if (MODE == 3) {
                    e0 = t00[3] ^ i0; e1 = t10[5]; f0 = t01[7]; f1 = t11[9] ^ i1;
                } else {
                    asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\t"
                                 "v_mov_b32 %0, v64\n\t"
                                 "v_mov_b32 %1, v96\n\t"
                                 "s_set_gpr_idx_idx %5\n\t"
                                 "v_mov_b32 %2, v80\n\t"
                                 "v_mov_b32 %3, v112\n\t"
                                 "s_set_gpr_idx_off"
                                 : "=&v"(e0), "=&v"(e1), "=&v"(f0), "=&v"(f1)
                                 : "s"(i0), "s"(i1), "{v[64:79]}"(t00), "{v[80:95]}"(t01), "{v[96:111]}"(t10), "{v[112:127]}"(t11));
                }
                if (MODE == 4) {
                    h[i][0] = e0 ^ f0; h[i][1] = e1 ^ f1;
                } else {
                    const uint32_t ta = BITOP3(anyP0, e0, f0, 0xe8);
                    h[i][0] = BITOP3(twoP0, ta, valid, 0x02);
                    const uint32_t tb = BITOP3(anyP1, e1, f1, 0xe8);
                    h[i][1] = BITOP3(twoP1, tb, valid, 0x02);
                }
                anyhit |= h[i][0] | h[i][1];
            }
            if (MODE == 1 || MODE == 4) {
                acc ^= anyhit;
            } else if (__any(anyhit == 0x12345678u)) { // never true on this data
                hits += anyhit;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = acc + hits;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
// Variant with the tables in LDS (one row group per lane): tab[u][v][lane], 32 KB per block.
// A column's key stays in a VGPR; two v_bfe/v_lshl_add give the lane's addresses, two
// ds_read_b32 fetch the entries.  No scalar instruction, no index mode.
__global__ __launch_bounds__(256) void klds(const uint32_t *in, uint32_t *out, unsigned long long *clk)
{
    __shared__ uint32_t ckey[TILE + 16];
    __shared__ uint32_t tab[2 * 16 * 256];
    const int tid = threadIdx.x;
    for (int i = 0; i < 32; i++) tab[i * 256 + tid] = in[(tid * 3 + i * 7) & 8191];
    uint32_t anyP0 = in[tid & 8191], twoP0 = in[(tid + 9) & 8191];
    const uint32_t valid = 0xFFFFFFFFu;
    const uint32_t lane_off = tid * 4;
    uint32_t acc = 0, hits = 0;
    unsigned long long t0 = 0, r0 = 0;
    for (int c0 = 0; c0 < NCOLS; c0 += TILE) {
        __syncthreads();
        for (int cc = tid; cc < TILE + 16; cc += 256) ckey[cc] = in[(c0 + cc) & 8191];
        __syncthreads();
        if (c0 == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
        uint32_t next[4];
        for (int i = 0; i < 4; i++) next[i] = ckey[i];
        for (int c = 0; c < TILE; c += 4) {
            uint32_t key[4], e[4], f[4];
            for (int i = 0; i < 4; i++) key[i] = next[i];
            for (int i = 0; i < 4; i++) next[i] = ckey[c + 4 + i];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t a0 = ((key[i] & 15u) << 10) + lane_off;
                const uint32_t a1 = (((key[i] >> 4) & 15u) << 10) + lane_off + 16 * 1024;
                e[i] = *(const uint32_t *)((const char *)tab + a0);
                f[i] = *(const uint32_t *)((const char *)tab + a1);
            }
            uint32_t anyhit = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t ta = BITOP3(anyP0, e[i], f[i], 0xe8);
                anyhit |= BITOP3(twoP0, ta, valid, 0x02);
            }
            if (__any(anyhit == 0x12345678u)) hits += anyhit;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = acc + hits;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
void run_lds(uint32_t *in, uint32_t *out, unsigned long long *clk, int blocks_per_cu)
{
    const int nblk = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); klds<<<nblk, 256>>>(in, out, clk); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> hc(2 * nblk);
    CK(hipMemcpy(hc.data(), clk, nblk * 16, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0; for (int b = 0; b < nblk; b++) { cyc += hc[2 * b]; real += hc[2 * b + 1]; }
    const double ghz = cyc / real * 0.1;
    const double cols_per_simd = (double)blocks_per_cu * NCOLS;
    printf("%-44s %d waves/SIMD %7.3f ms %.2f GHz -> %6.1f SIMD-cycles per column of 2048 rows (x2 = %6.1f per 4096)\n",
           "LDS tables, one group per lane", blocks_per_cu, best, ghz, best * 1e-3 * ghz * 1e9 / cols_per_simd,
           2 * best * 1e-3 * ghz * 1e9 / cols_per_simd);
}

template <int MODE> void run(const char *name, uint32_t *in, uint32_t *out, unsigned long long *clk, int blocks_per_cu)
{
    const int nblk = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); k<MODE><<<nblk, 256>>>(in, out, clk, 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> hc(2 * nblk);
    CK(hipMemcpy(hc.data(), clk, nblk * 16, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0; for (int b = 0; b < nblk; b++) { cyc += hc[2 * b]; real += hc[2 * b + 1]; }
    const double ghz = cyc / real * 0.1;
    const double cols_per_simd = (double)blocks_per_cu * NCOLS; // one wave of each block per SIMD
    printf("%-44s %d waves/SIMD %7.3f ms %.2f GHz -> %6.1f SIMD-cycles per column (%6.1f wave-cycles)\n", name, blocks_per_cu,
           best, ghz, best * 1e-3 * ghz * 1e9 / cols_per_simd, best * 1e-3 * ghz * 1e9 / NCOLS);
}
int main()
{
    uint32_t *in, *out; unsigned long long *clk;
    CK(hipMalloc(&in, 8192 * 4)); CK(hipMalloc(&out, (size_t)256 * 4 * 256 * 4)); CK(hipMalloc(&clk, 256 * 4 * 16));
    std::vector<uint32_t> h(8192); for (auto &x : h) x = rand();
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int w = 1; w <= 4; w++) {
        run<0>("full", in, out, clk, w);
        run<1>("no hit test", in, out, clk, w);
        run<2>("indices from the loop counter", in, out, clk, w);
        run<3>("no index window", in, out, clk, w);
        run<4>("window only", in, out, clk, w);
        run<6>("three live units (6 tables), tree + merge", in, out, clk, w);
        run<8>("second lookup folded into the majority op", in, out, clk, w);
        run_lds(in, out, clk, w);
    }
    return 0;
}
