// Prototype + micro-benchmark of the bit-sliced all-pairs filter (k = 1):
// rows live as 2-bit-code bit planes (32 rows per VGPR word), one column at a time is
// broadcast as 2L uniform masks, v_bitop3_b32 does the mismatch/sticky-counter logic.
// Build: hipcc -O3 --offload-arch=gfx950 tools/bsbench.hip -o build/bsbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int LP = 12;           // bases
constexpr int NP = 2 * LP;       // planes
constexpr int CT = 128;          // columns per LDS tile

// planesT[b * ngroups + g]: bit j = bit b of key[32 g + j]
__global__ void build_planes(const uint32_t *keys, uint32_t n, uint32_t ngroups, uint32_t *planesT)
{
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t key = row < n ? keys[row] : 0u;
    const uint32_t g = row >> 5;
    for (int b = 0; b < NP; b++) {
        const unsigned long long bal = __ballot((key >> b) & 1u);
        if ((threadIdx.x & 63) == 0) {
            if (g < ngroups) planesT[(size_t)b * ngroups + g] = (uint32_t)bal;
            if (g + 1 < ngroups) planesT[(size_t)b * ngroups + g + 1] = (uint32_t)(bal >> 32);
        }
    }
}

// MASKS: 0 = expanded masks staged in LDS (VGPR operands), 1 = masks from SGPRs (SALU expansion)
template <int G, int MASKS>
__global__ __launch_bounds__(256) void bs_pairs(const uint32_t *__restrict__ planesT, const uint32_t *__restrict__ keys,
                                                uint32_t ngroups, uint32_t n, uint32_t col_chunk,
                                                unsigned long long *hits, uint2 *hitlist, uint32_t hitcap)
{
    __shared__ __attribute__((aligned(16))) uint32_t cmask[CT * NP];
    __shared__ uint32_t ckeys[CT];
    const int tid = threadIdx.x;
    const uint32_t tiles_per_col = gridDim.y;
    (void)tiles_per_col;
    const uint32_t g0 = blockIdx.x * 256 * G; // first group of this row tile
    const uint32_t col_begin = blockIdx.y * col_chunk;
    const uint32_t col_end = min(n, col_begin + col_chunk);
    uint32_t p[G][NP];
    uint32_t valid[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const uint32_t grp = g0 + g * 256 + tid;
#pragma unroll
        for (int b = 0; b < NP; b++) p[g][b] = grp < ngroups ? planesT[(size_t)b * ngroups + grp] : 0u;
        const uint32_t r0 = grp * 32;
        valid[g] = r0 >= n ? 0u : (n - r0 >= 32 ? 0xFFFFFFFFu : ((1u << (n - r0)) - 1u));
    }
    unsigned long long local_hits = 0;
    for (uint32_t c0 = col_begin; c0 < col_end; c0 += CT) {
        const uint32_t nc = min((uint32_t)CT, col_end - c0);
        __syncthreads();
        if (MASKS == 0) {
            for (uint32_t w = tid; w < nc * NP; w += 256) {
                const uint32_t c = w / NP, b = w % NP;
                cmask[w] = ((keys[c0 + c] >> b) & 1u) ? 0xFFFFFFFFu : 0u;
            }
        } else {
            for (uint32_t w = tid; w < nc; w += 256) ckeys[w] = keys[c0 + w];
        }
        __syncthreads();
        for (uint32_t c = 0; c < nc; c++) {
            uint32_t cm[NP];
            if (MASKS == 0) {
#pragma unroll
                for (int q = 0; q < NP / 4; q++)
                    *reinterpret_cast<uint4 *>(&cm[4 * q]) = *reinterpret_cast<const uint4 *>(&cmask[c * NP + 4 * q]);
            } else {
                const uint32_t ck = __builtin_amdgcn_readfirstlane(ckeys[c]);
#pragma unroll
                for (int b = 0; b < NP; b++) cm[b] = (uint32_t)(((int32_t)(ck << (31 - b))) >> 31);
            }
            uint32_t anyhit = 0;
            uint32_t h[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                uint32_t s1 = 0, s2 = 0;
#pragma unroll
                for (int i = 0; i < LP; i += 2) {
                    // A=0xF0 B=0xCC C=0xAA truth-table convention
                    const uint32_t ta = p[g][2 * i] ^ cm[2 * i];
                    const uint32_t ma = __builtin_amdgcn_bitop3_b32(ta, p[g][2 * i + 1], cm[2 * i + 1], 0xF0 | (0xCC ^ 0xAA));
                    const uint32_t tb = p[g][2 * i + 2] ^ cm[2 * i + 2];
                    const uint32_t mb = __builtin_amdgcn_bitop3_b32(tb, p[g][2 * i + 3], cm[2 * i + 3], 0xF0 | (0xCC ^ 0xAA));
                    const uint32_t two = __builtin_amdgcn_bitop3_b32(s1, ma, mb, (0xF0 & 0xCC) | (0xF0 & 0xAA) | (0xCC & 0xAA));
                    s2 |= two;
                    s1 = __builtin_amdgcn_bitop3_b32(s1, ma, mb, 0xF0 | 0xCC | 0xAA);
                }
                h[g] = ~s2 & valid[g];
                anyhit |= h[g];
            }
            if (__any(anyhit != 0)) {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    uint32_t hh = h[g];
                    const uint32_t grp = g0 + g * 256 + tid;
                    while (hh) {
                        const int j = __builtin_ctz(hh);
                        hh &= hh - 1;
                        const uint32_t gi = grp * 32 + j, gj = c0 + c;
                        if (gi < gj) {
                            local_hits++;
                            if (hitlist) {
                                unsigned long long pos = atomicAdd(&hits[1], 1ull);
                                if (pos < hitcap) hitlist[pos] = make_uint2(gi, gj);
                            }
                        }
                    }
                }
            }
        }
    }
    if (local_hits) atomicAdd(&hits[0], local_hits);
}

static uint64_t sm(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int G, int MASKS>
double run(const char *name, const uint32_t *d_planes, const uint32_t *d_keys, uint32_t ngroups, uint32_t n,
           unsigned long long *d_hits, uint32_t col_chunk, unsigned long long *hits_out)
{
    dim3 grid((ngroups + 256 * G - 1) / (256 * G), (n + col_chunk - 1) / col_chunk);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(d_hits, 0, 16));
        CK(hipEventRecord(e0));
        bs_pairs<G, MASKS><<<grid, 256>>>(d_planes, d_keys, ngroups, n, col_chunk, d_hits, nullptr, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipMemcpy(hits_out, d_hits, 8, hipMemcpyDeviceToHost));
    double pairs_eval = (double)grid.x * 256 * G * 32 * (double)n; // full square incl. padding
    double w = (double)n * (n - 1) / 2;
    printf("%-26s n=%u grid=%ux%u  %8.3f ms  %6.2f Tpairs/s evaluated (full square)  hits(i<j)=%llu  [W=%.3g]\n", name, n, grid.x,
           grid.y, best, pairs_eval / (best * 1e-3) / 1e12, *hits_out, w);
    return best;
}

int main(int argc, char **argv)
{
    uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144;
    uint32_t col_chunk = argc > 2 ? (uint32_t)atoi(argv[2]) : 4096;
    std::vector<uint32_t> keys(n);
    uint64_t s = 42;
    for (auto &k : keys) k = (uint32_t)(sm(s) & 0xFFFFFFu);
    // CPU reference count on a prefix (exact 2-bit base mismatch count <= 1, i<j)
    uint32_t nref = n < 20000 ? n : 20000;
    unsigned long long ref = 0;
    for (uint32_t i = 0; i < nref; i++)
        for (uint32_t j = i + 1; j < nref; j++) {
            uint32_t x = keys[i] ^ keys[j];
            uint32_t y = (x | (x >> 1)) & 0x555555u;
            if (__builtin_popcount(y) <= 1) ref++;
        }
    uint32_t *d_keys, *d_planes; unsigned long long *d_hits;
    uint32_t ngroups = (n + 31) / 32;
    CK(hipMalloc(&d_keys, (size_t)n * 4)); CK(hipMalloc(&d_planes, (size_t)NP * ngroups * 4)); CK(hipMalloc(&d_hits, 16));
    CK(hipMemcpy(d_keys, keys.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    build_planes<<<(n + 255) / 256, 256>>>(d_keys, n, ngroups, d_planes);
    CK(hipDeviceSynchronize());
    unsigned long long h;
    // correctness on the prefix
    {
        uint32_t ng = (nref + 31) / 32;
        uint32_t *dp; CK(hipMalloc(&dp, (size_t)NP * ng * 4));
        build_planes<<<(nref + 255) / 256, 256>>>(d_keys, nref, ng, dp);
        run<2, 0>("check G=2 LDS masks", dp, d_keys, ng, nref, d_hits, col_chunk, &h);
        printf("  reference hits on %u-prefix: %llu -> %s\n", nref, ref, h == ref ? "MATCH" : "MISMATCH");
        run<2, 1>("check G=2 SGPR masks", dp, d_keys, ng, nref, d_hits, col_chunk, &h);
        printf("  -> %s\n", h == ref ? "MATCH" : "MISMATCH");
    }
    run<1, 0>("G=1 LDS masks", d_planes, d_keys, ngroups, n, d_hits, col_chunk, &h);
    run<2, 0>("G=2 LDS masks", d_planes, d_keys, ngroups, n, d_hits, col_chunk, &h);
    run<4, 0>("G=4 LDS masks", d_planes, d_keys, ngroups, n, d_hits, col_chunk, &h);
    run<1, 1>("G=1 SGPR masks", d_planes, d_keys, ngroups, n, d_hits, col_chunk, &h);
    run<2, 1>("G=2 SGPR masks", d_planes, d_keys, ngroups, n, d_hits, col_chunk, &h);
    run<4, 1>("G=4 SGPR masks", d_planes, d_keys, ngroups, n, d_hits, col_chunk, &h);
    return 0;
}
