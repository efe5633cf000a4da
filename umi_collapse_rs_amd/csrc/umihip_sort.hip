// Key sort for the optional "prune" mode: entries of one large bucket ordered by their
// filter key, so that a tile of consecutive rows/columns shares its leading bases and
// whole (row tile, column chunk) tasks whose shared prefixes already differ in more than k
// bases can be skipped without evaluating a pair.  rocPRIM's radix sort is used as a plain
// library primitive (the pair kernels stay hand-written); its own TU keeps the main one quick
// to compile.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "umihip_internal.h"

namespace umihip {

namespace {
__global__ __launch_bounds__(256) void iota_kernel(uint32_t *out, uint32_t start, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = start + i;
}

template <typename KeyT>
__global__ __launch_bounds__(256) void gather_kernel(const KeyT *__restrict__ keys,
                                                     const uint32_t *__restrict__ pos, uint32_t n,
                                                     uint64_t *__restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = (uint64_t)keys[pos[i]];
}
} // namespace

// Onesweep radix passes above SORT_MERGE_LIMIT items (the library's default switches to a block
// sort + ten merge passes below 2^20 items: 0.16 ms for the 970,714 keys of BASELINE config 2),
// and only over the bits a key of this UMI length can have.  rocPRIM takes its merge path for
// size <= merge_sort_limit (device_radix_sort.hpp:644), i.e. a bucket of exactly
// SORT_MERGE_LIMIT entries still merges.
constexpr uint32_t SORT_MERGE_LIMIT = 65536;
using sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                               rocprim::default_config, SORT_MERGE_LIMIT>;

bool sort_is_onesweep(uint32_t n) { return n > SORT_MERGE_LIMIT; }

size_t sort_temp_bytes(bool key32, uint32_t n, int key_bits)
{
    size_t bytes = 0;
    if (key32)
        (void)rocprim::radix_sort_pairs<sort_config>(nullptr, bytes, (const uint32_t *)nullptr,
                                                     (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                                     (uint32_t *)nullptr, n, 0, key_bits);
    else
        (void)rocprim::radix_sort_pairs<sort_config>(nullptr, bytes, (const uint64_t *)nullptr,
                                                     (uint64_t *)nullptr, (const uint32_t *)nullptr,
                                                     (uint32_t *)nullptr, n, 0, key_bits);
    return bytes;
}

hipError_t sort_bucket(const void *fkey, bool key32, int begin_bit, int key_bits, uint32_t start, uint32_t n,
                       void *fkey_sorted, uint32_t *perm, uint32_t *iota_tmp, void *tmp,
                       size_t tmp_bytes, hipStream_t s)
{
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    iota_kernel<<<blocks, 256, 0, s>>>(iota_tmp + start, start, n);
    if (key32)
        return rocprim::radix_sort_pairs<sort_config>(tmp, tmp_bytes, (const uint32_t *)fkey + start,
                                                      (uint32_t *)fkey_sorted + start, iota_tmp + start,
                                                      perm + start, n, begin_bit, key_bits, s);
    return rocprim::radix_sort_pairs<sort_config>(tmp, tmp_bytes, (const uint64_t *)fkey + start,
                                                  (uint64_t *)fkey_sorted + start, iota_tmp + start,
                                                  perm + start, n, begin_bit, key_bits, s);
}

hipError_t gather_keys(const void *keys, bool key32, const uint32_t *pos, uint32_t n, uint64_t *out,
                       hipStream_t s)
{
    if (n == 0) return hipSuccess;
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (key32)
        gather_kernel<uint32_t><<<blocks, 256, 0, s>>>((const uint32_t *)keys, pos, n, out);
    else
        gather_kernel<uint64_t><<<blocks, 256, 0, s>>>((const uint64_t *)keys, pos, n, out);
    return hipGetLastError();
}

} // namespace umihip
