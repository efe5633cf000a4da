// LDS return-path probe: how many cycles does a wave-uniform (broadcast) ds_read_b128 / b64 / b32
// cost per CU, and does it overlap with full-rate VALU work?  256 threads/block, 5 blocks/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITERS = 2048;

// MODE 0: 12 broadcast ds_read_b128 per iteration, no VALU
// MODE 1: 120 bitop3 per iteration, no LDS
// MODE 2: both
// MODE 3: 24 broadcast ds_read_b64
// MODE 4: 12 per-lane (non-broadcast, conflict-free) ds_read_b128
// MODE 5: 6 broadcast ds_read_b128 + 120 bitop3
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t *in, uint32_t *out, unsigned long long *clk)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    uint32_t acc[8];
    for (int c = 0; c < 8; c++) acc[c] = in[threadIdx.x + c];
    uint32_t z = in[threadIdx.x ^ 3];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        const uint32_t base = (it & 15) * 192; // wave-uniform word offset
        uint4 m[12];
        if (MODE == 0 || MODE == 2 || MODE == 5) {
#pragma unroll
            for (int q = 0; q < (MODE == 5 ? 6 : 12); q++) m[q] = *reinterpret_cast<const uint4 *>(&lds[base + 4 * q]);
        }
        if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 12; q++) {
                uint2 a = *reinterpret_cast<const uint2 *>(&lds[base + 4 * q]);
                uint2 b = *reinterpret_cast<const uint2 *>(&lds[base + 4 * q + 2]);
                m[q] = make_uint4(a.x, a.y, b.x, b.y);
            }
        }
        if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 12; q++) m[q] = *reinterpret_cast<const uint4 *>(&lds[((threadIdx.x & 63) * 4 + q * 256 + base) & 4092]);
        }
        if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 12; q++) m[q] = make_uint4(z + q, z ^ q, z, z - q);
        }
        if (MODE != 0 && MODE != 3 && MODE != 4) {
#pragma unroll
            for (int u = 0; u < 15; u++)
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const uint4 v = m[(u + c) % (MODE == 5 ? 6 : 12)];
                    acc[c] = __builtin_amdgcn_bitop3_b32(acc[c], (u & 1) ? v.x : v.z, (u & 2) ? v.y : v.w, 0xf6);
                }
        } else {
#pragma unroll
            for (int q = 0; q < 12; q++) acc[q & 7] ^= m[q].x ^ m[q].y ^ m[q].z ^ m[q].w;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t r = 0;
    for (int c = 0; c < 8; c++) r += acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int MODE> void run(const char *name, uint32_t *in, uint32_t *out, unsigned long long *clk, int nblk)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); k<MODE><<<nblk, 256>>>(in, out, clk); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> hc(2 * nblk);
    CK(hipMemcpy(hc.data(), clk, nblk * 16, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0; for (int b = 0; b < nblk; b++) { cyc += hc[2 * b]; real += hc[2 * b + 1]; }
    double ghz = cyc / real * 0.1;
    // wave-iterations per SIMD = nblk*4 waves / (256 CU * 4 SIMD) * ITERS
    double iters_per_simd = (double)nblk / 256.0 * ITERS;
    printf("%-44s %7.3f ms  %.2f GHz -> %7.1f SIMD-cycles per wave-iteration\n", name, best, ghz,
           best * 1e-3 * ghz * 1e9 / iters_per_simd);
}
int main()
{
    int nblk = 256 * 5;
    uint32_t *in, *out; unsigned long long *clk;
    CK(hipMalloc(&in, 8192 * 4)); CK(hipMalloc(&out, (size_t)nblk * 256 * 4)); CK(hipMalloc(&clk, nblk * 16));
    std::vector<uint32_t> h(8192); for (auto &x : h) x = rand();
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<0>("12 x broadcast ds_read_b128", in, out, clk, nblk);
    run<3>("24 x broadcast ds_read_b64", in, out, clk, nblk);
    run<4>("12 x per-lane ds_read_b128", in, out, clk, nblk);
    run<1>("120 x v_bitop3", in, out, clk, nblk);
    run<2>("12 x broadcast ds_read_b128 + 120 x v_bitop3", in, out, clk, nblk);
    run<5>(" 6 x broadcast ds_read_b128 + 120 x v_bitop3", in, out, clk, nblk);
    return 0;
}
