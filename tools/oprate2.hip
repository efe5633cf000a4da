// Issue-rate probe for candidate VALU instructions (inline asm, 8 independent chains/lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITERS = 2048;
constexpr int CH = 8;

#define OP1(str) asm volatile(str : "+v"(a[c]), "+v"(m[c]) : "s"(col), "v"(z))

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t *in, uint32_t *out, unsigned long long *clk)
{
    uint32_t a[CH], m[CH];
    for (int c = 0; c < CH; c++) { a[c] = in[threadIdx.x + c * 256]; m[c] = in[threadIdx.x + c * 256 + 7]; }
    uint32_t col = in[blockIdx.x & 1023];
    uint32_t z = in[threadIdx.x ^ 5];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (MODE == 0) OP1("v_xor_b32 %0, %2, %0");
                if (MODE == 1) OP1("v_bcnt_u32_b32 %0, %1, %0");
                if (MODE == 2) OP1("v_min3_u32 %0, %0, %1, %2");
                if (MODE == 3) OP1("v_min_u32 %0, %2, %0");
                if (MODE == 4) OP1("v_and_or_b32 %0, %1, %3, %0");
                if (MODE == 5) OP1("v_or3_b32 %0, %0, %1, %3");
                if (MODE == 6) OP1("v_bfi_b32 %0, %0, %1, %3");
                if (MODE == 7) OP1("v_lshl_or_b32 %0, %0, 1, %1");
                if (MODE == 8) OP1("v_fma_f32 %0, %0, %1, %3");
                if (MODE == 9) OP1("v_add3_u32 %0, %0, %1, %3");
                if (MODE == 10) OP1("v_or_b32 %0, %1, %0");
                if (MODE == 11) OP1("v_xor_b32_e64 %0, %2, %0");
                if (MODE == 12) OP1("v_xad_u32 %0, %0, %1, %3");
                if (MODE == 13) OP1("v_cmp_ne_u32 vcc, %0, %1");
                if (MODE == 14) OP1("v_xor_b32 %0, %2, %0\n\tv_xor_b32 %1, %2, %1\n\tv_or_b32 %0, %1, %0\n\tv_and_or_b32 %1, %0, %3, %1\n\tv_or_b32 %0, %3, %0");
                if (MODE == 15) OP1("v_pk_max_u16 %0, %0, %1");
                if (MODE == 16) OP1("v_sad_u8 %0, %0, %1, %3");
                if (MODE == 17) OP1("v_dot4_i32_i8 %0, %0, %1, %3");
                if (MODE == 18) OP1("v_mul_u32_u24 %0, %1, %0");
                if (MODE == 19) OP1("v_perm_b32 %0, %0, %1, %3");
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int c = 0; c < CH; c++) acc += a[c] + m[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE> void run(const char *name, double ops, uint32_t *in, uint32_t *out, unsigned long long *clk, int nblk)
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
    double laneops = (double)nblk * 256 * ITERS * 8 * CH * ops;
    double ghz = cyc / real * 0.1;
    double rate = laneops / (best * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD = (4 SIMD * 32 lanes... ) normalise to "slots": peak lanes/clk/CU = 128
    printf("%-22s %7.3f ms %7.2f Tlaneop/s  clk %.2f GHz  -> %.2f SIMD-cycles per wave-instr\n", name, best, rate, ghz,
           64.0 / (rate * 1e12 / (256.0 * 4 * ghz * 1e9)));
}

int main()
{
    int nblk = 256 * 8;
    uint32_t *in, *out; unsigned long long *clk;
    CK(hipMalloc(&in, 8192 * 4)); CK(hipMalloc(&out, (size_t)nblk * 256 * 4)); CK(hipMalloc(&clk, nblk * 16));
    std::vector<uint32_t> h(8192); for (auto &x : h) x = rand();
    CK(hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice));
    run<0>("v_xor_b32 (VOP2,sgpr)", 1, in, out, clk, nblk);
    run<11>("v_xor_b32_e64", 1, in, out, clk, nblk);
    run<10>("v_or_b32", 1, in, out, clk, nblk);
    run<3>("v_min_u32 (VOP2)", 1, in, out, clk, nblk);
    run<1>("v_bcnt_u32_b32", 1, in, out, clk, nblk);
    run<2>("v_min3_u32", 1, in, out, clk, nblk);
    run<4>("v_and_or_b32", 1, in, out, clk, nblk);
    run<5>("v_or3_b32", 1, in, out, clk, nblk);
    run<6>("v_bfi_b32", 1, in, out, clk, nblk);
    run<7>("v_lshl_or_b32", 1, in, out, clk, nblk);
    run<8>("v_fma_f32", 1, in, out, clk, nblk);
    run<9>("v_add3_u32", 1, in, out, clk, nblk);
    run<12>("v_xad_u32", 1, in, out, clk, nblk);
    run<13>("v_cmp_ne_u32", 1, in, out, clk, nblk);
    run<15>("v_pk_max_u16", 1, in, out, clk, nblk);
    run<16>("v_sad_u8", 1, in, out, clk, nblk);
    run<17>("v_dot4_i32_i8", 1, in, out, clk, nblk);
    run<18>("v_mul_u32_u24", 1, in, out, clk, nblk);
    run<19>("v_perm_b32", 1, in, out, clk, nblk);
    run<14>("bitslice base (5 ops)", 5, in, out, clk, nblk);
    return 0;
}
