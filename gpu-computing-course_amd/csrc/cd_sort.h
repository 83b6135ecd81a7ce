// cd_sort.h -- 64-bit-key / 32-bit-value LSD radix sort for gfx950 (wave64), replacing the host
// thrust::sort_by_key of load_obj.h:107.  8 passes x 8 bits; per pass: tile histogram -> exclusive
// scan of (digit, tile) counts -> stable scatter.  Stability inside a tile comes from wave-level
// match masks (__ballot over the 8 digit bits, rank = popcount of lower matching lanes), per-wave
// digit counters in LDS and a cross-wave prefix; no atomics on the data path, so the result is the
// unique stable ascending order (ties keep original index order), run to run identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cd {

constexpr int SORT_THREADS = 256;                      // 4 waves
constexpr int SORT_WAVES   = SORT_THREADS / 64;
constexpr int SORT_ITEMS   = 16;                       // keys per thread
constexpr int SORT_TILE    = SORT_THREADS * SORT_ITEMS; // 4096 keys per workgroup
constexpr int RADIX_BITS   = 8;
constexpr int RADIX        = 1 << RADIX_BITS;

// counts[digit * ntiles + tile]
__global__ __launch_bounds__(SORT_THREADS) void k_radix_hist(const uint64_t *__restrict__ keys, uint32_t n, int shift,
                                                             uint32_t *__restrict__ counts, uint32_t ntiles)
{
    __shared__ uint32_t h[RADIX];
    const uint32_t tile = blockIdx.x;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = tile * SORT_TILE;
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t i = base + it * SORT_THREADS + threadIdx.x;      // coalesced 8-byte loads
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    counts[threadIdx.x * ntiles + tile] = h[threadIdx.x];
}

// In-place exclusive scan of `total` uint32 by ONE workgroup of 1024 threads (total = 256*ntiles,
// 62 720 at 1 M keys): each thread owns a contiguous chunk, wave shuffles + LDS for the chunk sums.
__global__ __launch_bounds__(1024) void k_scan_exclusive(uint32_t *__restrict__ data, uint32_t total)
{
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (total + 1023u) / 1024u;
    const uint32_t lo = tid * chunk;
    const uint32_t hi = min(lo + chunk, total);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += data[i];
    // inclusive scan of s across the block
    uint32_t v = s;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o); if (lane >= o) v += t; }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    if (w == 0) {
        uint32_t x = (lane < 16) ? wsum[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { uint32_t t = __shfl_up(x, o); if (lane >= o) x += t; }
        if (lane < 16) wsum[lane] = x;                                   // inclusive over waves
    }
    __syncthreads();
    uint32_t run = v - s + (w ? wsum[w - 1] : 0);                         // exclusive prefix of this chunk
    for (uint32_t i = lo; i < hi; ++i) { uint32_t t = data[i]; data[i] = run; run += t; }
}

// Stable scatter of one tile. offsets[digit * ntiles + tile] = global position of the tile's first
// key with that digit (from k_scan_exclusive).
__global__ __launch_bounds__(SORT_THREADS) void k_radix_scatter(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                                uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                                uint32_t n, int shift, const uint32_t *__restrict__ offsets, uint32_t ntiles,
                                                                int first_pass)
{
    __shared__ uint32_t wcnt[SORT_WAVES][RADIX];       // per-wave running digit counts -> then wave bases
    __shared__ uint32_t gbase[RADIX];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < SORT_WAVES * RADIX; i += SORT_THREADS) (&wcnt[0][0])[i] = 0;
    gbase[tid] = offsets[tid * ntiles + tile];
    __syncthreads();

    // Wave w owns the contiguous keys [base_w, base_w + ITEMS*64): iteration `it` covers 64 consecutive
    // keys, so (it, lane) order == index order inside the wave, and waves are ordered by w.
    const uint32_t base_w = tile * SORT_TILE + w * (SORT_ITEMS * 64);
    uint64_t k[SORT_ITEMS];
    uint32_t v[SORT_ITEMS];
    uint32_t rk[SORT_ITEMS];                           // rank of the key among equal digits of this wave
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t i = base_w + it * 64 + lane;
        const bool ok = i < n;
        k[it] = ok ? keys_in[i] : ~0ull;
        v[it] = ok ? (first_pass ? i : vals_in[i]) : 0u;
        const uint32_t d = (uint32_t)(k[it] >> shift) & (RADIX - 1);
        // lanes with the same digit (invalid lanes form their own group through the `ok` ballot)
        uint64_t m = __ballot(ok) ;
        m = ok ? m : ~m;
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const uint64_t bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t below = __popcll(m & lt_mask);
        const uint32_t cnt = __popcll(m);
        const int leader = __ffsll((unsigned long long)m) - 1;
        uint32_t old = 0;
        if (ok && lane == leader) { old = wcnt[w][d]; wcnt[w][d] = old + cnt; }
        old = __shfl(old, leader);
        rk[it] = old + below;
    }
    __syncthreads();
    // exclusive prefix over waves for each digit
    {
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < SORT_WAVES; ++ww) { const uint32_t c = wcnt[ww][tid]; wcnt[ww][tid] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t i = base_w + it * 64 + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(k[it] >> shift) & (RADIX - 1);
            const uint32_t pos = gbase[d] + wcnt[w][d] + rk[it];
            keys_out[pos] = k[it];
            vals_out[pos] = v[it];
        }
    }
}

}  // namespace cd
