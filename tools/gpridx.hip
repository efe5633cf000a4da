// Probe: (1) v_bitop3_b32 issue rate, (2) does VGPR index mode (s_set_gpr_idx_on/idx/off)
// work on gfx950 hardware and what does an indexed VOP2/VOP3 cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITERS = 2048;

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t *in, uint32_t *out, unsigned long long *clk, uint32_t sel)
{
    uint32_t p0 = in[threadIdx.x], p1 = in[threadIdx.x + 256], p2 = in[threadIdx.x + 512], p3 = in[threadIdx.x + 768];
    uint32_t a[8];
    for (int c = 0; c < 8; c++) a[c] = in[threadIdx.x + 1024 + c * 256];
    uint32_t x = __builtin_amdgcn_readfirstlane(sel + blockIdx.x) & 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 8; c++)
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(a[c]) : "v"(p0), "v"(p1));
            } else if (MODE == 1) { // indexed or: a[c] |= plane[x]
                asm volatile("s_set_gpr_idx_on %8, gpr_idx(SRC0)\n\t"
                             "v_or_b32 %0, v40, %0\n\tv_or_b32 %1, v40, %1\n\tv_or_b32 %2, v40, %2\n\tv_or_b32 %3, v40, %3\n\t"
                             "v_or_b32 %4, v40, %4\n\tv_or_b32 %5, v40, %5\n\tv_or_b32 %6, v40, %6\n\tv_or_b32 %7, v40, %7\n\t"
                             "s_set_gpr_idx_off"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "s"(x), "{v40}"(p0), "{v41}"(p1), "{v42}"(p2), "{v43}"(p3));
            } else if (MODE == 2) { // set idx per op (1 SALU per VALU)
                asm volatile("s_set_gpr_idx_on %8, gpr_idx(SRC0)\n\t"
                             "v_or_b32 %0, v40, %0\n\ts_set_gpr_idx_idx %8\n\tv_or_b32 %1, v40, %1\n\ts_set_gpr_idx_idx %8\n\t"
                             "v_or_b32 %2, v40, %2\n\ts_set_gpr_idx_idx %8\n\tv_or_b32 %3, v40, %3\n\ts_set_gpr_idx_idx %8\n\t"
                             "v_or_b32 %4, v40, %4\n\ts_set_gpr_idx_idx %8\n\tv_or_b32 %5, v40, %5\n\ts_set_gpr_idx_idx %8\n\t"
                             "v_or_b32 %6, v40, %6\n\ts_set_gpr_idx_idx %8\n\tv_or_b32 %7, v40, %7\n\t"
                             "s_set_gpr_idx_off"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "s"(x), "{v40}"(p0), "{v41}"(p1), "{v42}"(p2), "{v43}"(p3));
            } else if (MODE == 3) { // indexed and_or
                asm volatile("s_set_gpr_idx_on %8, gpr_idx(SRC0)\n\t"
                             "v_and_or_b32 %0, v40, %1, %0\n\tv_and_or_b32 %1, v40, %2, %1\n\tv_and_or_b32 %2, v40, %3, %2\n\tv_and_or_b32 %3, v40, %4, %3\n\t"
                             "v_and_or_b32 %4, v40, %5, %4\n\tv_and_or_b32 %5, v40, %6, %5\n\tv_and_or_b32 %6, v40, %7, %6\n\tv_and_or_b32 %7, v40, %0, %7\n\t"
                             "s_set_gpr_idx_off"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "s"(x), "{v40}"(p0), "{v41}"(p1), "{v42}"(p2), "{v43}"(p3));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int c = 0; c < 8; c++) acc |= a[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// correctness of the indexed read: out = plane[x] for x = blockIdx & 3
__global__ void check(const uint32_t *in, uint32_t *out)
{
    uint32_t p0 = in[threadIdx.x], p1 = in[threadIdx.x + 256], p2 = in[threadIdx.x + 512], p3 = in[threadIdx.x + 768];
    uint32_t x = blockIdx.x & 3, r = 0;
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_or_b32 %0, v40, %0\n\ts_set_gpr_idx_off"
                 : "+v"(r) : "s"(x), "{v40}"(p0), "{v41}"(p1), "{v42}"(p2), "{v43}"(p3));
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, uint32_t *in, uint32_t *out, unsigned long long *clk, int nblk)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); k<MODE><<<nblk, 256>>>(in, out, clk, 1); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> hc(2 * nblk);
    CK(hipMemcpy(hc.data(), clk, nblk * 16, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0; for (int b = 0; b < nblk; b++) { cyc += hc[2 * b]; real += hc[2 * b + 1]; }
    double ghz = cyc / real * 0.1;
    double laneops = (double)nblk * 256 * ITERS * 8 * 8;
    double rate = laneops / (best * 1e-3) / 1e12;
    printf("%-34s %7.3f ms %7.2f Tlaneop/s clk %.2f GHz -> %.2f SIMD-cycles per VALU instr\n", name, best, rate, ghz,
           64.0 / (rate * 1e12 / (256.0 * 4 * ghz * 1e9)));
}

int main()
{
    int nblk = 256 * 8;
    uint32_t *in, *out; unsigned long long *clk;
    CK(hipMalloc(&in, 8192 * 4)); CK(hipMalloc(&out, (size_t)nblk * 256 * 4)); CK(hipMalloc(&clk, nblk * 16));
    std::vector<uint32_t> h(8192); for (auto &x : h) x = rand();
    CK(hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice));
    check<<<8, 256>>>(in, out); CK(hipDeviceSynchronize());
    std::vector<uint32_t> o(8 * 256); CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int b = 0; b < 8; b++) for (int t = 0; t < 256; t++) if (o[b * 256 + t] != h[t + 256 * (b & 3)]) bad++;
    printf("VGPR index mode check: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    run<0>("v_bitop3_b32 (vgpr x3)", in, out, clk, nblk);
    run<1>("indexed v_or_b32 (1 set per 8)", in, out, clk, nblk);
    run<2>("indexed v_or_b32 (set_idx per op)", in, out, clk, nblk);
    run<3>("indexed v_and_or_b32", in, out, clk, nblk);
    return 0;
}
