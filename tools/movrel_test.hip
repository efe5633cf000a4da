// Does gfx950 execute v_movrels_b32 (LLVM emits it for dynamic register indexing only with
// -Xclang -target-feature -Xclang +movrel; the default on gfx9 is s_set_gpr_idx_on/off)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const uint32_t *in, uint32_t *out, int idx)
{
    u32x16 t;
    for (int i = 0; i < 16; i++) t[i] = in[threadIdx.x * 16 + i];
    out[threadIdx.x] = t[idx & 15] ^ (t[(idx >> 4) & 15] << 1);
}
int main()
{
    uint32_t *in, *out;
    hipMalloc(&in, 64 * 16 * 4);
    hipMalloc(&out, 64 * 4);
    std::vector<uint32_t> h(64 * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int idx = 0; idx < 256; idx++) {
        k<<<1, 64>>>(in, out, idx);
        std::vector<uint32_t> o(64);
        hipMemcpy(o.data(), out, 64 * 4, hipMemcpyDeviceToHost);
        for (int t = 0; t < 64; t++)
            bad += o[t] != (h[t * 16 + (idx & 15)] ^ (h[t * 16 + ((idx >> 4) & 15)] << 1));
    }
    printf("movrel test: %d mismatches\n", bad);
    return bad != 0;
}
