// cd_sort.h -- 64-bit-key / 32-bit-value LSD radix sort for gfx950 (wave64), replacing the host
// thrust::sort_by_key of load_obj.h:107.  8-bit digits, onesweep form (one pass over the data per digit, tile
// offsets by decoupled look-back), 4 passes + a fix-up hop in half-key mode or 8 passes.  Stability inside a tile
// comes from wave-level match masks (__ballot over the 8 digit bits, rank = popcount of lower matching lanes),
// per-wave digit counters in LDS and a cross-wave prefix; no atomics on the data path, so the result is the
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

// In-place exclusive scan of `total` uint32 by ONE workgroup of 1024 threads (the unique-flag scan of the pair
// post-processing, cd_post.h): each thread owns a contiguous chunk, wave shuffles + LDS for the chunk sums.
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

// ====================================================================================================
// Onesweep form (single pass over the data per digit, decoupled look-back) instead of a histogram / scan /
// scatter triple per digit (the first version of this file: 950 us at 1 M keys):
//   (k_morton)     : builds the global digit histograms while it writes the keys
//   k_os_pass x 4|8: per tile -- stable local ranking, then the tile's global
//                    offsets come from a chained look-back over the preceding tiles' published
//                    {status, count} granules instead of a separate scan kernel.
// Hand-off protocol (cdna_hip_programming.md Guideline 16, form R2): the datum IS the flag -- one
// naturally aligned 8-byte granule {status:2 | value:62} written by ONE relaxed agent-scope atomic store
// (sc1, write-through) and polled with relaxed agent-scope atomic loads (sc1, L1-bypassing); no fence is
// needed because nothing else is published.  Tile ids are handed out by an atomic ticket, so every
// predecessor of a spinning tile is already resident and publishes without waiting on anyone: no deadlock
// whatever the dispatch order.  Granules are zeroed by one hipMemsetAsync per sort.
// Measured alternatives at 1 M keys (profiles/r01_experiments/): wider look-back batches (32) and a flat sum
// over all predecessors' aggregates did not help -- the ~10 us a pass spends here is the rendezvous itself (a
// tile cannot scatter before every earlier tile has ranked), not the walk; forwarding each key into the NEXT
// pass's per-tile histogram with global atomics while scattering (no look-back at all) cost 3x (1 M scattered
// atomics per pass).
constexpr int OS_THREADS = 1024;                       // 16 waves x 4 keys per lane = the same 4096-key tile; the serial ranking chain per wave is what a pass waits for
constexpr int OS_WAVES   = OS_THREADS / 64;
constexpr int OS_ITEMS   = SORT_TILE / OS_THREADS;
constexpr unsigned long long OS_AGG = 1ull << 62;      // value = this tile's count for the digit
constexpr unsigned long long OS_PREFIX = 2ull << 62;   // value = inclusive prefix over tiles 0..t
constexpr unsigned long long OS_VALUE_MASK = (1ull << 62) - 1;

// Digit histograms: ghist[p][d] = number of keys whose digit p equals d, accumulated by k_morton (cd_bvh.h) while it
// writes the keys -- workgroup-local LDS histograms flushed with one global atomic per non-empty bin.  hist_add() is
// the per-key part.
__device__ __forceinline__ void hist_add(uint32_t (*h)[RADIX], uint64_t k, int first_digit)
{
    if (first_digit == 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) atomicAdd(&h[p][(k >> (8 * p)) & 255], 1u);
    }
#pragma unroll
    for (int p = 4; p < 8; ++p) atomicAdd(&h[p][(k >> (8 * p)) & 255], 1u);
}

// Stand-alone histogram kernel for keys that do not come from k_morton (the pair post-processing sort).
__global__ __launch_bounds__(SORT_THREADS) void k_os_hist(const uint64_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ ghist /* [8][256] */)
{
    __shared__ uint32_t h[8][RADIX];
    for (int i = threadIdx.x; i < 8 * RADIX; i += SORT_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * SORT_THREADS + threadIdx.x; i < n; i += gridDim.x * SORT_THREADS) hist_add(h, keys[i], 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * RADIX; i += SORT_THREADS) {
        const uint32_t v = (&h[0][0])[i];
        if (v) atomicAdd(&ghist[i], v);
    }
}

__global__ __launch_bounds__(OS_THREADS) void k_os_pass(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                          uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                          uint32_t n, int shift, const uint32_t *__restrict__ digit_hist /* [256] raw counts of this digit */,
                                                          unsigned long long *lookback /* [ntiles][256] */, uint32_t *ticket /* [pass]; ticket[8 - pass] = timeout flag */, int first_pass)
{
    __shared__ uint32_t wcnt[OS_WAVES][RADIX];
    __shared__ uint32_t gbase[RADIX];
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_wsum[RADIX / 64];
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < OS_WAVES * RADIX; i += OS_THREADS) (&wcnt[0][0])[i] = 0;
    // digit bases = exclusive scan of the 256 raw counts; every tile does it for itself (no scan kernel)
    uint32_t dbase = 0;
    if (tid < RADIX) {
        const uint32_t c = digit_hist[tid];
        uint32_t v = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(v, o); if (lane >= o) v += t; }
        if (lane == 63) s_wsum[w] = v;
        dbase = v - c;
    }
    __syncthreads();
    if (tid < RADIX) for (int ww = 0; ww < w; ++ww) dbase += s_wsum[ww];
    const uint32_t tile = s_tile;

    const uint32_t base_w = tile * SORT_TILE + w * (OS_ITEMS * 64);
    uint64_t k[OS_ITEMS];
    uint32_t v[OS_ITEMS];
    uint32_t rk[OS_ITEMS];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < OS_ITEMS; ++it) {
        const uint32_t i = base_w + it * 64 + lane;
        const bool ok = i < n;
        k[it] = ok ? keys_in[i] : ~0ull;
        v[it] = ok ? (first_pass ? i : vals_in[i]) : 0u;
    }
#pragma unroll
    for (int it = 0; it < OS_ITEMS; ++it) {
        const uint32_t i = base_w + it * 64 + lane;
        const bool ok = i < n;
        const uint32_t d = (uint32_t)(k[it] >> shift) & (RADIX - 1);
        uint64_t m = __ballot(ok);
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
    // thread = digit (the first RADIX threads): wave-exclusive bases, the tile's count, publish, look back
    if (tid < RADIX) {
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < OS_WAVES; ++ww) { const uint32_t c = wcnt[ww][tid]; wcnt[ww][tid] = run; run += c; }
        const unsigned long long count = run;
        unsigned long long *mine = lookback + (size_t)tile * RADIX + tid;
        unsigned long long excl = 0;
        if (tile == 0) {
            __hip_atomic_store(mine, OS_PREFIX | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(mine, OS_AGG | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Walk back over the predecessors LB at a time: the LB granule loads are independent and in flight
            // together, so a step costs one L2 round trip instead of LB of them.
            constexpr int LB = 8;
            int t = (int)tile - 1;
            uint32_t spins = 0;
            bool done = false;
            while (!done) {
                unsigned long long g[LB];
#pragma unroll
                for (int b = 0; b < LB; ++b)
                    g[b] = (t - b >= 0) ? __hip_atomic_load(lookback + (size_t)(t - b) * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : OS_PREFIX;
                int used = 0;
#pragma unroll
                for (int b = 0; b < LB; ++b) {
                    if (done || used != b) continue;                          // stop at the first unpublished granule
                    const unsigned long long st = g[b] & ~OS_VALUE_MASK;
                    if (st == 0) continue;
                    excl += g[b] & OS_VALUE_MASK;
                    used = b + 1;
                    if (st == OS_PREFIX) done = true;
                }
                t -= used;
                if (!done && used < LB) {                                     // hit an unpublished predecessor: wait a little
                    if (++spins > (1u << 22)) { atomicExch(ticket + 8, 1u); break; }   // bounded: give up, flag the sort as failed
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __hip_atomic_store(mine, OS_PREFIX | (excl + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gbase[tid] = dbase + (uint32_t)excl;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < OS_ITEMS; ++it) {
        const uint32_t i = base_w + it * 64 + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(k[it] >> shift) & (RADIX - 1);
            const uint32_t pos = gbase[d] + wcnt[w][d] + rk[it];
            keys_out[pos] = k[it];
            vals_out[pos] = v[it];
        }
    }
}


// ====================================================================================================
// Half-key sort + fix-up.  The onesweep passes are latency-bound at 1 M keys (~25 us each), so the cheapest pass
// is the one not run: sort (stably) by the HIGH 32 bits only -- 4 passes -- and repair the order inside every run
// of equal high halves with a stable insertion sort on the full key.  The result is exactly the stable sort by
// the full 64-bit key (ties of the full key keep input order in both), so keys / permutation stay bit-identical
// to the 8-pass sort.  Morton codes of distinct triangles rarely share their top 32 bits (~10.7 bits per axis),
// so runs are 1-3 long; a run longer than FIX_MAX sets `overflow` and the host redoes the sort with all 8 passes
// (and keeps doing so for that context).
// ====================================================================================================
constexpr int FIX_MAX = 16;

// Out of place (in -> out): thread i finds its run by looking at most FIX_MAX keys back and forward and counts the
// keys of the run that must precede its own -- earlier ones with key <= mine, later ones with key < mine (stable) --
// which is its position inside the run.  Typical cost: two neighbour reads and one copy.
// Final position of key i (value k0) inside its run of equal high halves; false if the run may exceed FIX_MAX.
// The two neighbours are fetched up front (independent loads): for most keys they already end the run.
__device__ __forceinline__ bool fixup_position(const uint64_t *__restrict__ keys_in, uint32_t n, uint32_t i, uint64_t k0, uint32_t &pos)
{
    const uint32_t h = (uint32_t)(k0 >> 32);
    const uint64_t kp = i > 0 ? keys_in[i - 1] : ~k0, kn = i + 1 < n ? keys_in[i + 1] : ~k0;   // ~k0: a different high half
    uint32_t back = 0, before = 0, fwd = 0;
    if ((uint32_t)(kp >> 32) == h) {
        before += kp <= k0; back = 1;
        while (back < (uint32_t)FIX_MAX && back < i) {
            const uint64_t kb = keys_in[i - 1 - back];
            if ((uint32_t)(kb >> 32) != h) break;
            before += kb <= k0;
            ++back;
        }
    }
    if ((uint32_t)(kn >> 32) == h) {
        before += kn < k0; fwd = 1;
        while (fwd < (uint32_t)FIX_MAX && i + 1 + fwd < n) {
            const uint64_t kf = keys_in[i + 1 + fwd];
            if ((uint32_t)(kf >> 32) != h) break;
            before += kf < k0;
            ++fwd;
        }
    }
    pos = i - back + before;
    return back < (uint32_t)FIX_MAX && fwd < (uint32_t)FIX_MAX;
}

}  // namespace cd
