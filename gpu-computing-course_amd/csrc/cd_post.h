// cd_post.h -- post-processing of the pair list on the device: the step right after the hot path in the
// reference harness (main.cu:149-154 prints the pairs, main.cu:33-45 makeAndPrintSet builds the std::set of
// colliding triangle IDs on the host).  Deterministic order for diffing: pairs sorted ascending by (a, b);
// triangle set = sorted unique IDs.  Sorting reuses the onesweep kernels of cd_sort.h on 64-bit keys.
#pragma once
#include "cd_sort.h"

namespace cd {

// keys[i] = (a << 32) | b for pair i
__global__ __launch_bounds__(256) void k_pairs_to_keys(const uint32_t *__restrict__ pairs, uint32_t m, uint64_t *__restrict__ keys)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) keys[i] = ((uint64_t)pairs[2 * (size_t)i] << 32) | pairs[2 * (size_t)i + 1];
}
__global__ __launch_bounds__(256) void k_keys_to_pairs(const uint64_t *__restrict__ keys, uint32_t m, uint32_t *__restrict__ pairs)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) { pairs[2 * (size_t)i] = (uint32_t)(keys[i] >> 32); pairs[2 * (size_t)i + 1] = (uint32_t)keys[i]; }
}
// keys[i] = ID i of the flattened pair list (2m entries)
__global__ __launch_bounds__(256) void k_ids_to_keys(const uint32_t *__restrict__ pairs, uint32_t m2, uint64_t *__restrict__ keys)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < m2) keys[i] = pairs[i];
}
// flags[i] = 1 where a new value starts in the sorted sequence
__global__ __launch_bounds__(256) void k_unique_flags(const uint64_t *__restrict__ keys, uint32_t m, uint32_t *__restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// pos = exclusive scan of flags; out[pos[i]] = keys[i] at every run start; total written to *count by the last item
__global__ __launch_bounds__(256) void k_unique_scatter(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ pos, uint32_t m,
                                                        uint32_t *__restrict__ out, uint32_t cap, uint32_t *__restrict__ count)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const bool first = (i == 0 || keys[i] != keys[i - 1]);
    if (first && pos[i] < cap) out[pos[i]] = (uint32_t)keys[i];
    if (i == m - 1) *count = pos[i] + (first ? 1u : 0u);
}

}  // namespace cd
