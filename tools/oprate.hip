// Micro-benchmark: issue rate of the integer VALU ops the pair kernel is made of
// (v_xor_b32, v_bcnt_u32_b32, v_min3_u32) on gfx950, to calibrate the VALU roofline.
// Build: hipcc -O3 --offload-arch=gfx950 tools/oprate.hip -o build/oprate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int CH = 8; // independent chains per lane

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(const uint32_t *in, uint32_t *out, unsigned long long *clk)
{
    uint32_t a[CH], m[CH];
    for (int c = 0; c < CH; c++) { a[c] = in[threadIdx.x + c * 256]; m[c] = 255; }
    uint32_t col = in[blockIdx.x & 1023];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (MODE == 0) { // xor only: 2 ops
                    asm volatile("v_xor_b32 %0, %2, %0\n\tv_xor_b32 %1, %0, %1" : "+v"(a[c]), "+v"(m[c]) : "s"(col));
                } else if (MODE == 1) { // bcnt only: 2 ops (accumulating form)
                    asm volatile("v_bcnt_u32_b32 %1, %0, %1\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(a[c]), "+v"(m[c]));
                } else if (MODE == 2) { // min3 only: 1 op
                    asm volatile("v_min3_u32 %1, %1, %0, %2" : "+v"(a[c]), "+v"(m[c]) : "s"(col));
                } else if (MODE == 3) { // the pair-kernel mix: 2 xor + 2 bcnt + 1 min3 = 5 ops / 2 pairs
                    uint32_t p0 = __builtin_popcount(a[c] ^ col);
                    uint32_t p1 = __builtin_popcount(a[c] ^ (col + 1));
                    m[c] = min(min(m[c], p0), p1);
                } else if (MODE == 4) { // v_add only: 2 ops
                    asm volatile("v_add_u32 %0, %2, %0\n\tv_add_u32 %1, %0, %1" : "+v"(a[c]), "+v"(m[c]) : "s"(col));
                }
            }
            col = col * 3 + 1; // scalar, keeps iterations distinct
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int c = 0; c < CH; c++) acc += a[c] + m[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char **argv)
{
    int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 8;
    int nblk = 256 * blocks_per_cu;
    uint32_t *in, *out; unsigned long long *clk;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, (size_t)nblk * 256 * 4)); CK(hipMalloc(&clk, nblk * 16));
    std::vector<uint32_t> h(4096); for (auto &x : h) x = rand();
    CK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"v_xor x2", "v_bcnt x2", "v_min3 x1", "pair mix (5 ops / 2 pairs)", "v_add x2"};
    const double ops[] = {2, 2, 1, 5, 2};
    for (int mode = 0; mode < 5; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            switch (mode) {
            case 0: rate_kernel<0><<<nblk, 256>>>(in, out, clk); break;
            case 1: rate_kernel<1><<<nblk, 256>>>(in, out, clk); break;
            case 2: rate_kernel<2><<<nblk, 256>>>(in, out, clk); break;
            case 3: rate_kernel<3><<<nblk, 256>>>(in, out, clk); break;
            case 4: rate_kernel<4><<<nblk, 256>>>(in, out, clk); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        std::vector<unsigned long long> hc(2 * nblk);
        CK(hipMemcpy(hc.data(), clk, nblk * 16, hipMemcpyDeviceToHost));
        double cyc = 0, real = 0; for (int b = 0; b < nblk; b++) { cyc += hc[2 * b]; real += hc[2 * b + 1]; }
        double clock_ghz = cyc / real * 0.1; // s_memrealtime ticks at 100 MHz
        double laneops = (double)nblk * 256 * ITERS * 4 * CH * ops[mode];
        printf("%-28s %8.3f ms  %7.2f Tlaneop/s  in-kernel clock %.2f GHz  (%d blocks/CU)\n", names[mode], best,
               laneops / (best * 1e-3) / 1e12, clock_ghz, blocks_per_cu);
    }
    return 0;
}
