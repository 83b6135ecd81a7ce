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
// The sort's flag word (one word, or-ed by the kernels of a sort, read by the host with the step's report: mi355cd.hip judge_sort_flags) -- WHY this form of the
// sort cannot finish, so that the host goes to the form that can instead of trying them one after the other:
constexpr uint32_t SORTF_RUN = 1u;        // a run of equal global digits is too long for k_local_sort's windows -> the larger windows, then the half-key form (no windows)
constexpr uint32_t SORTF_ABOVE = 2u;      // a key has bits above the shifted global digits (a centroid outside the Morton frame) -> the digits at key bits 48..63
constexpr uint32_t SORTF_FIXUP = 4u;      // more than FIX_MAX keys share their high half: no form with a fix-up hop can finish -> all eight passes

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
// Round 3: the FIRST global pass has no rendezvous at all.  k_morton takes whole tiles and leaves, per tile, the counts of the first
// global digit (tile_hist[tile][256]); a tile of the first pass sums the rows of the earlier tiles (16-byte loads, 16 waves x 4 rows
// in flight: ~1 us) instead of looking back, and is numbered by blockIdx.x (no arrival-order ticket, one round trip less): 17.4 -> 10.9 us
// for that pass at 1 M keys (k_morton 13.2 -> 13.4 us).  The later passes'
// input order only exists once the pass before them has run, so they keep the look-back.
// The histograms that come with the Morton keys are kept as HIST_COPIES partial tables (workgroup b of k_morton adds to table
// b mod HIST_COPIES; k_os_pass adds the tables up): its ~500 workgroups all flush at the end of the kernel, and atomics on one
// word retire one after the other.  1 / 2 / 4 / 8 / 16 tables: k_morton 17.0 / 12.9 / 12.4 / 12.6 / 12.5 us, k_os_pass 17.4 /
// 17.5 / 17.7 / 18.2 / 19.6 us per pass (1 M keys).
constexpr int HIST_COPIES = 2, HIST_STRIDE = 8 * RADIX;
constexpr int OS_THREADS = 1024;                       // 16 waves x 4 keys per lane = the same 4096-key tile; the serial ranking chain per wave is what a pass waits for
constexpr int OS_WAVES   = OS_THREADS / 64;
constexpr int OS_ITEMS   = SORT_TILE / OS_THREADS;
constexpr unsigned long long OS_AGG = 1ull << 62;      // value = this tile's count for the digit
constexpr unsigned long long OS_PREFIX = 2ull << 62;   // value = inclusive prefix over tiles 0..t
constexpr unsigned long long OS_VALUE_MASK = (1ull << 62) - 1;

// Digit histograms: ghist[p][d] = number of keys whose digit p equals d, accumulated by k_morton (cd_bvh.h) while it
// writes the keys -- workgroup-local LDS histograms flushed with one global atomic per non-empty bin.  hist_add() is
// the per-key part.
// `down`: the digits are taken `down` bits lower (the shifted hybrid sort: digits 6 and 7 are key bits 44..51 and 52..59).
__device__ __forceinline__ void hist_add(uint32_t (*h)[RADIX], uint64_t k, int first_digit, int down = 0)
{
#pragma unroll
    for (int p = 0; p < 8; ++p)
        if (p >= first_digit) atomicAdd(&h[p][(k >> (8 * p - down)) & 255], 1u);      // (first_digit, down are workgroup-uniform; down <= 8 * first_digit)
}

// The lanes of the wave whose digit equals this lane's (and only lanes with `ok`; a lane without matches the lanes without): the multi-split of the
// ranking, a ballot per digit bit.  Written with the compare as inline asm: from `__ballot((d >> b) & 1u)` the compiler makes an add-with-carry (the
// carry IS the ballot, the difference the select mask), then writes the carry out as 0 / 1 and compares it again for the ballot intrinsic -- ten
// instructions and two s_nop a bit where this is seven; a key is 8 bits, a lane of k_local_sort ranks 10 keys a pass (1 M keys: k_os_pass 11.7 + 18.4 -> 11.3 + 18.1 us,
// k_local_sort 27.6 -> 26.8; same keys, same permutation).
__device__ __forceinline__ uint64_t digit_match(uint32_t d, bool ok)
{
    uint64_t m = __ballot(ok);
    m = ok ? m : ~m;
    uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
        const uint32_t bit = (d >> b) & 1u;
        uint64_t bb;
        asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(bb) : "v"(bit));           // (= the ballot of `bit`: a VALU compare writes 0 for the lanes that are not executing)
        const uint32_t sm = bit - 1u;                                       // 0 where the bit is set (keep the lanes that have it), ~0 where it is not (keep the others)
        lo &= (uint32_t)bb ^ sm; hi &= (uint32_t)(bb >> 32) ^ sm;
    }
    return ((uint64_t)hi << 32) | lo;
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

// The per-tile digit counts of k_morton summed over chunks of OS_CHUNK tiles, for sorts of more than OS_CHUNK_MIN_TILES tiles (k_os_pass, first pass): a workgroup
// per chunk, a thread per digit, the chunk's rows in flight together.
constexpr int OS_CHUNK = 32, OS_CHUNK_MIN_TILES = 512;
__global__ __launch_bounds__(RADIX) void k_tile_chunks(const uint32_t *__restrict__ tile_hist, uint32_t ntiles, uint32_t *__restrict__ chunk_tot)
{
    const uint32_t c = blockIdx.x, d = threadIdx.x, t0 = c * OS_CHUNK;
    uint32_t v[OS_CHUNK];
#pragma unroll
    for (int u = 0; u < OS_CHUNK; ++u) v[u] = t0 + u < ntiles ? tile_hist[(size_t)(t0 + u) * RADIX + d] : 0u;
    uint32_t sum = 0;
#pragma unroll
    for (int u = 0; u < OS_CHUNK; ++u) sum += v[u];
    chunk_tot[(size_t)c * RADIX + d] = sum;
}

#ifndef OS_WAVES_PER_EU
#define OS_WAVES_PER_EU 8
#endif
// (68 VGPRs left to itself: seven waves a SIMD, i.e. ONE 1024-thread workgroup a CU.  Held at 64 -- two workgroups a CU -- a sort of many tiles has somebody to run while a
//  workgroup waits for its look-back: DESIGN.md 4a)
__global__ __launch_bounds__(OS_THREADS) __attribute__((amdgpu_waves_per_eu(OS_WAVES_PER_EU, OS_WAVES_PER_EU))) void k_os_pass(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                          uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                          uint32_t n, int shift, const uint32_t *__restrict__ digit_hist /* [256] raw counts of this digit, hist_copies partial tables HIST_STRIDE words apart */,
                                                          unsigned long long *lookback /* [ntiles][256] */, uint32_t *ticket /* [pass]; ticket[8 - pass] = timeout flag */, int first_pass, int hist_copies,
                                                          const uint32_t *__restrict__ tile_hist /* NULL, or [ntiles][256]: this digit's counts per INPUT tile, left by the kernel that wrote the keys
                                                                                                    (k_morton): the tile offsets are their sums, no look-back */,
                                                          const uint32_t *__restrict__ chunk_tot /* NULL, or [ntiles / OS_CHUNK][256]: the same counts summed over chunks of OS_CHUNK tiles (k_tile_chunks) */)
{
    __shared__ uint32_t wcnt[OS_WAVES][RADIX];
    __shared__ uint32_t gbase[RADIX];
    __shared__ uint32_t tpart[OS_THREADS / 64][RADIX];     // flat tile offsets: 16 partial sums per digit
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_wsum[RADIX / 64];
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_tile = tile_hist ? blockIdx.x : atomicAdd(ticket, 1u);      // (no look-back, no need for arrival order: the ticket's round trip is saved)
    for (int i = tid; i < OS_WAVES * RADIX; i += OS_THREADS) (&wcnt[0][0])[i] = 0;
    // digit bases = exclusive scan of the 256 raw counts; every tile does it for itself (no scan kernel)
    uint32_t dbase = 0;
    if (tid < RADIX) {
        uint32_t c = 0;
        for (int k = 0; k < hist_copies; ++k) c += digit_hist[(size_t)k * HIST_STRIDE + tid];
        uint32_t v = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(v, o); if (lane >= o) v += t; }
        if (lane == 63) s_wsum[w] = v;
        dbase = v - c;
    }
    __syncthreads();
    if (tid < RADIX) for (int ww = 0; ww < w; ++ww) dbase += s_wsum[ww];
    const uint32_t tile = s_tile;
    // The first global pass of a sort whose keys come with per-tile digit counts (k_morton): how many keys of digit d the EARLIER
    // tiles hold is a sum over rows that were complete before this kernel started -- plain loads, no dependence between the tiles of
    // this pass at all, where the look-back below is a rendezvous of all of them (~9 us of a 17 us pass at 245 tiles).
    // (round 5) The sum over ALL earlier tiles is quadratic in the number of tiles: 30 MB of L2 reads at 1 M keys (245 tiles), 1.9 GB at 8 M -- that pass took 113 us
    // there, longer than the pass WITH look-back.  Beyond OS_CHUNK_MIN_TILES tiles a small kernel in front (k_tile_chunks) sums the rows of every chunk of OS_CHUNK
    // tiles; a tile then adds the totals of the chunks before its own and the rows of the earlier tiles of its own chunk.
    if (tile_hist) {
        // 64 lanes x 4 digits cover a row (16-byte loads), the 16 waves take every 16th row, four rows in flight per lane
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, a2 = a0, a3 = a0;
        auto add_rows = [&](const uint32_t *table, uint32_t from, uint32_t to) {
            const uint4 *rows = reinterpret_cast<const uint4 *>(table);
            uint32_t t2 = from + (uint32_t)w;
            for (; t2 + 3u * OS_WAVES < to; t2 += 4u * OS_WAVES) {
                const uint4 r0 = rows[(size_t)t2 * (RADIX / 4) + lane], r1 = rows[(size_t)(t2 + OS_WAVES) * (RADIX / 4) + lane];
                const uint4 r2 = rows[(size_t)(t2 + 2u * OS_WAVES) * (RADIX / 4) + lane], r3 = rows[(size_t)(t2 + 3u * OS_WAVES) * (RADIX / 4) + lane];
                a0.x += r0.x; a0.y += r0.y; a0.z += r0.z; a0.w += r0.w; a1.x += r1.x; a1.y += r1.y; a1.z += r1.z; a1.w += r1.w;
                a2.x += r2.x; a2.y += r2.y; a2.z += r2.z; a2.w += r2.w; a3.x += r3.x; a3.y += r3.y; a3.z += r3.z; a3.w += r3.w;
            }
            for (; t2 < to; t2 += OS_WAVES) { const uint4 r0 = rows[(size_t)t2 * (RADIX / 4) + lane]; a0.x += r0.x; a0.y += r0.y; a0.z += r0.z; a0.w += r0.w; }
        };
        const uint32_t own = chunk_tot ? tile / (uint32_t)OS_CHUNK : 0u;     // (workgroup-uniform)
        if (chunk_tot) add_rows(chunk_tot, 0u, own);
        add_rows(tile_hist, own * (uint32_t)OS_CHUNK, tile);
        reinterpret_cast<uint4 *>(&tpart[w][0])[lane] = make_uint4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z, a0.w + a1.w + a2.w + a3.w);
    }

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
        const uint64_t m = digit_match(d, ok);
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
        if (tile_hist) {
#pragma unroll
            for (int q = 0; q < OS_THREADS / 64; ++q) excl += tpart[q][tid];
        } else if (tile == 0) {
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
// Hybrid sort (default).  The onesweep passes cost ~20 us each at 1 M keys, nearly all of it the cross-XCD hand-off
// between tiles; a pass that stays inside a workgroup has no hand-off.  So only 16 key bits are sorted globally (2
// onesweep passes) -- bits 44..59: a Morton key of a centroid inside the frame is below 2^60 (morton.h:70-89), so these
// are its top 16 bits, 2^16 cells over the frame; k_morton notices a key at or above 2^60 and the sort is redone on bits
// 48..63 -- which leaves runs of equal top bits (tens to hundreds of keys), contiguous and in input order, and
// k_local_sort finishes bits 32..63 inside LDS:
//   * workgroup b takes the window [s_b, s_{b+1}), where s_b is the run start nearest to b * LOCAL_W -- no run
//     straddles two windows, so sorting the windows independently sorts the array;
//   * inside the window: stable LSD passes over digits 4..7 with the same __ballot ranking as k_os_pass, {high key
//     half, position} held in registers and permuted through LDS; a pass whose digit is the same for the whole
//     window is skipped; keys and values are gathered once at the end.
// The result is the stable order by the high 32 bits, exactly what 4 global passes give; the kernel's epilogue then places
// every key by its low half as k_sort_fixup_fill does after the half-key sort (a run of equal high halves lies inside one
// window), so keys and permutation stay bit-identical to the full 8-pass sort.  A run
// longer than LOCAL_LIMIT cannot be windowed: the kernel flags it (same word as the fix-up's overflow) and the host
// falls back to the 4-pass half-key sort, then to 8 passes.
// ====================================================================================================
// The window configuration is a template parameter (round 5): 512 threads x 10 keys a lane (5120 keys of LDS capacity: nominal windows of 2048 keys, runs up to
// 3072 -- 48 KB and 8 waves: TWO workgroups a CU, one computing while the other waits at one of its ~20 barriers; the default at every size: 100 k keys 18.3 -> 14.4 us,
// 1 M 30.9 -> 27.8, 8 M 269 -> 199), or rounds 2-4's 1024 threads x 10 keys (10240: windows of 4096, runs up to 6144 -- 96 KB, one workgroup a CU).  A run too
// long for the small form raises the same flag as always; the host then redoes the sort with the large form before it escalates to more global passes.
#ifndef LOCAL_SMALL_BLOCKS
#define LOCAL_SMALL_BLOCKS 2
#endif
template <int THREADS_, int ITEMS_, int LIMIT_, int BLOCKS_> struct LocalCfg {
    static constexpr int THREADS = THREADS_, WAVES = THREADS_ / 64, ITEMS = ITEMS_, LIMIT = LIMIT_, CAP = ITEMS_ * THREADS_, W = CAP - LIMIT_, BLOCKS = BLOCKS_ /* workgroups a CU the registers are held to */;
    static_assert(THREADS_ >= RADIX && THREADS_ % 64 == 0 && LIMIT_ % THREADS_ == 0 && W > 0 && ITEMS_ % 2 == 0, "window configuration");
};
using LocalLarge = LocalCfg<1024, 10, 6144, 1>;            // (a planar 1 M cloth in the reference's frame: runs of ~500 on key bits 44..59; 441 runs, the longest 4590, on bits 48..63)
using LocalSmall = LocalCfg<512, 10, 3072, LOCAL_SMALL_BLOCKS>;             // (8 M triangles of cloth objects: runs up to 1878 on bits 44..59; a 4 M cloth pair: 2312)
constexpr int LOCAL_W = LocalLarge::W;                  // nominal keys per workgroup of the large form (the host's window arithmetic)

__device__ __forceinline__ bool fixup_position_lds(const uint2 *item, uint32_t cnt, uint32_t j, uint32_t low, uint32_t high, uint32_t &pos);
template <class Emit, class CFG>
__global__ __launch_bounds__(CFG::THREADS) __attribute__((amdgpu_waves_per_eu(CFG::BLOCKS * CFG::WAVES / 4, CFG::BLOCKS * CFG::WAVES / 4))) void k_local_sort(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                              uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n,
                                                              int run_shift /* a run = equal key bits [run_shift, 64): what the global passes sorted by */,
                                                              uint32_t *__restrict__ overflow, Emit emit /* load(value) -> payload, store(final position, value, payload): what the fix-up hop does per key */,
                                                              uint32_t *__restrict__ zero_words /* fused step: counters of the kernels that follow, zeroed here instead of by a memset */, uint32_t n_zero,
                                                              uint32_t win /* nominal keys per workgroup, <= CFG::W: n spread over the chip's CUs when that is less */)
{
    constexpr int LOCAL_THREADS = CFG::THREADS, LOCAL_WAVES = CFG::WAVES, LOCAL_ITEMS = CFG::ITEMS, LOCAL_LIMIT = CFG::LIMIT, LOCAL_CAP = CFG::CAP;
    if (blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < n_zero; i += LOCAL_THREADS) zero_words[i] = 0u;
    __shared__ uint2 sitem[LOCAL_CAP];                   // 80 KB: {high 32 key bits, position inside the window}
    __shared__ uint32_t wcnt[LOCAL_WAVES][RADIX];        // 16 KB
    __shared__ uint32_t s_wsum[RADIX / 64];
    __shared__ uint32_t s_enc[2];
    __shared__ uint32_t s_and, s_or;
    __shared__ uint32_t s_wruns[LOCAL_WAVES];
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const uint32_t p0 = blockIdx.x * win, p1 = min(p0 + win, n);
    const uint32_t *khw = reinterpret_cast<const uint32_t *>(keys_in) + 1;      // khw[2 i]: the high word of key i
    const int hshift = run_shift - 32;                                          // (run_shift >= 44: the run bits are in the high word)
    if (tid == 0) { s_enc[0] = s_enc[1] = 0xffffffffu; }
    __syncthreads();
    // Window ends: the run start NEAREST to p0 and to p1 (a run starts where the top 16 bits change; equal distance:
    // the lower one), searched LOCAL_LIMIT / 2 positions to either side -- the windows then differ from their nominal
    // size by at most half a run to either side, instead of a whole run to one.  The function p -> nearest start is
    // monotone, so windows never overlap.  All the probes of a thread (two keys each) are issued before any is used.
    // First the LOCAL_THREADS positions around p0 and p1 (one per thread: runs of a mesh are a few hundred keys, the nearest
    // start is almost always there, and it is THE nearest if it is); the whole range only for an end that found nothing.
    {
        // (UNCONDITIONAL loads at clamped positions, of the keys' high words only -- run_shift >= 32 --, selected afterwards: a load in
        //  one arm of a ?: is a branch around the load with its wait inside, i.e. a round trip per probe instead of one for all four)
        constexpr int NEAR = LOCAL_THREADS / 2;
        uint32_t top[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t p = e ? p1 : p0;
            const long long q = (long long)p - NEAR + (int)tid;
            const bool live = p != 0 && p < n && q >= 0 && q < (long long)n;
            top[e][0] = khw[2 * (size_t)((live && q > 0) ? q - 1 : 0)];
            top[e][1] = khw[2 * (size_t)(live ? q : 0)];
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t p = e ? p1 : p0;
            const long long q = (long long)p - NEAR + (int)tid;
            const bool live = p != 0 && p < n && q >= 0 && q < (long long)n;
            top[e][0] = (live && q > 0) ? top[e][0] >> hshift : 0xffffffffu;
            top[e][1] = live ? top[e][1] >> hshift : 0xffffffffu;
            if (live && top[e][0] != top[e][1]) {
                const uint32_t dist = (uint32_t)(q > (long long)p ? q - p : p - q);
                atomicMin(&s_enc[e], (dist << 1) | (q > (long long)p ? 1u : 0u));
            }
        }
    }
    __syncthreads();
    const bool far0 = p0 != 0 && s_enc[0] == 0xffffffffu, far1 = p1 < n && s_enc[1] == 0xffffffffu;    // (workgroup-uniform)
    if (far0 || far1) {
        constexpr int PROBES = LOCAL_LIMIT / LOCAL_THREADS, HALF = LOCAL_LIMIT / 2;
        uint32_t top[2][PROBES][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t p = e ? p1 : p0;
#pragma unroll
            for (int u = 0; u < PROBES; ++u) {
                const long long q = (long long)p - HALF + (u * LOCAL_THREADS + (int)tid);
                const bool live = (e ? far1 : far0) && q >= 0 && q < (long long)n;
                top[e][u][0] = khw[2 * (size_t)((live && q > 0) ? q - 1 : 0)];
                top[e][u][1] = khw[2 * (size_t)(live ? q : 0)];
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t p = e ? p1 : p0;
#pragma unroll
            for (int u = 0; u < PROBES; ++u) {
                const long long q = (long long)p - HALF + (u * LOCAL_THREADS + (int)tid);
                const bool live = (e ? far1 : far0) && q >= 0 && q < (long long)n;
                top[e][u][0] = (live && q > 0) ? top[e][u][0] >> hshift : 0xffffffffu;    // 0xffffffff: "differs" (q == 0 is a start); run_shift >= 44
                top[e][u][1] = live ? top[e][u][1] >> hshift : 0xffffffffu;
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t p = e ? p1 : p0;
#pragma unroll
            for (int u = 0; u < PROBES; ++u) {
                const long long q = (long long)p - HALF + (u * LOCAL_THREADS + (int)tid);
                const bool live = (e ? far1 : far0) && q >= 0 && q < (long long)n;
                if (live && top[e][u][0] != top[e][u][1]) {
                    const uint32_t dist = (uint32_t)(q > (long long)p ? q - p : p - q);
                    atomicMin(&s_enc[e], (dist << 1) | (q > (long long)p ? 1u : 0u));
                }
            }
        }
    }
    __syncthreads();
    const uint32_t e0 = s_enc[0], e1 = s_enc[1];
    const bool found0 = p0 == 0 || e0 != 0xffffffffu, found1 = p1 >= n || e1 != 0xffffffffu;
    if (!found0 || !found1) {                            // a run longer than LOCAL_LIMIT: flag it, pass the nominal range through
        if (tid == 0) atomicOr(overflow, SORTF_RUN);
        for (uint32_t i = p0 + tid; i < p1; i += LOCAL_THREADS) { const uint32_t t = vals_in[i]; keys_out[i] = keys_in[i]; vals_out[i] = t; emit.store(i, t, emit.load(t)); }
        return;
    }
    const int lo = p0 == 0 ? 0 : (int)((e0 & 1u) ? p0 + (e0 >> 1) : p0 - (e0 >> 1));
    const int hi = p1 >= n ? (int)n : (int)((e1 & 1u) ? p1 + (e1 >> 1) : p1 - (e1 >> 1));
    const uint32_t cnt = (uint32_t)(hi - lo);            // <= LOCAL_W + LOCAL_LIMIT - 1
    // Only {high key half, position} travels through the passes (3 registers per key with its rank; whole keys and
    // values spilled); keys and values are gathered from the window once, at the end.
    // Wave w owns contiguous items (item order == position order, as stability needs), whole rounds of 64: every wave
    // cnt / 1024 rounds, the first waves one more until the window is covered -- a window of 4200 keys costs fourteen
    // waves 4 item rounds per pass and two waves 5 (with equal shares rounded up to 64 it cost thirteen waves 5 rounds and
    // left three idle; the windows end at run starts, so half of them are a little over their nominal 4096 keys).  Every
    // sweep below stops at `nit` (wave-uniform; no barrier sits inside a sweep).
    const uint32_t full = cnt / (uint32_t)LOCAL_THREADS, extra = ((cnt % (uint32_t)LOCAL_THREADS) + 63u) >> 6;      // (a power of two: a shift and a mask)
    const int nit = (int)(full + ((uint32_t)w < extra ? 1u : 0u));            // <= LOCAL_ITEMS
    const uint32_t base_w = ((uint32_t)w * full + min((uint32_t)w, extra)) * 64u;
    uint32_t kh[LOCAL_ITEMS], ix[LOCAL_ITEMS], rk[LOCAL_ITEMS];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // What travels is not the high key half itself but {index of the key's run inside the window, key bits 32 .. run_shift - 1}:
    // the runs are in order already (the global passes), so this composite sorts like the high half -- and it is SHORT: a
    // window of a mesh holds a handful of long runs, 4 bits of run index + the 12 bits below the run bits are two
    // 8-bit passes instead of three (a window of many short runs needs the third, as before).  The run index of an item
    // is the number of run starts at or before it: a ballot per item round, a running count, the waves' totals through LDS.
    {
        const int low_bits = run_shift - 32;             // 12 or 16
        const unsigned long long le_mask = lt_mask | (1ull << lane);
        // Every item's high key word in ONE round trip: unconditional loads at clamped positions, all issued before any is used (the
        // key BEFORE an item is the neighbour lane's, the item round before for lane 0, one more load for the wave's first item).
        uint32_t before = 0, ridx[LOCAL_ITEMS], hw[LOCAL_ITEMS];
        const uint32_t prev0 = khw[2 * (size_t)(lo + (base_w > 0 ? (int)min(base_w, cnt) - 1 : 0))];
#pragma unroll
        for (int it = 0; it < LOCAL_ITEMS; ++it) if (it < nit) {
            const uint32_t j = base_w + it * 64 + lane;
            hw[it] = khw[2 * (size_t)(lo + (int)(j < cnt ? j : cnt - 1))];
        }
#pragma unroll
        for (int it = 0; it < LOCAL_ITEMS; ++it) if (it < nit) {
            const uint32_t j = base_w + it * 64 + lane;
            const uint32_t edge = it ? (uint32_t)__builtin_amdgcn_readlane((int)hw[it ? it - 1 : 0], 63) : prev0;      // what lane 0 compares with
            const uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)hw[it], 0x138, 0xf, 0xf, false);   // wave_shr:1
            const unsigned long long m = __ballot(j < cnt && j > 0 && (hw[it] >> hshift) != (hp >> hshift));
            ridx[it] = before + (uint32_t)__popcll(m & le_mask);
            before += (uint32_t)__popcll(m);
            kh[it] = hw[it] & ((1u << low_bits) - 1u);
            ix[it] = j;
        }
        if (lane == 0) s_wruns[w] = before;
        __syncthreads();
        uint32_t woff = 0;
        for (int ww = 0; ww < w; ++ww) woff += s_wruns[ww];
#pragma unroll
        for (int it = 0; it < LOCAL_ITEMS; ++it) if (it < nit)
            kh[it] = base_w + it * 64 + lane < cnt ? (((woff + ridx[it]) << low_bits) | kh[it]) : ~0u;
    }
    // which digits vary at all inside the window (typically not the top one): one AND / OR reduction for all four
    if (tid == 0) { s_and = 0xffffffffu; s_or = 0u; }
    __syncthreads();
    {
        uint32_t a = 0xffffffffu, o = 0u;
#pragma unroll
        for (int it = 0; it < LOCAL_ITEMS; ++it)
            if (it < nit && base_w + it * 64 + lane < cnt) { a &= kh[it]; o |= kh[it]; }
        for (int off = 32; off; off >>= 1) { a &= __shfl_xor(a, off); o |= __shfl_xor(o, off); }
        if (lane == 0) { atomicAnd(&s_and, a); atomicOr(&s_or, o); }
    }
    __syncthreads();
    const uint32_t varying = s_and ^ s_or;
    for (int digit = 0; digit < 4; ++digit) {            // digits of the composite (see above)
        const int shift = digit * RADIX_BITS;
        if (((varying >> shift) & (RADIX - 1)) == 0) continue;            // (workgroup-uniform) one value in this digit: already in order
        for (int i = tid; i < LOCAL_WAVES * RADIX; i += LOCAL_THREADS) (&wcnt[0][0])[i] = 0;
        __syncthreads();
        // Stable rank of every item among the items of its wave with the same digit (k_os_pass's scheme), arranged so
        // that no item waits for the previous one's LDS round trip: one returning LDS add per match group (the LDS
        // executes a wave's atomics in order, so a later group of the same digit sees the earlier add), and the old
        // counts are broadcast in a second sweep.
        uint32_t old[LOCAL_ITEMS]; int lead[LOCAL_ITEMS];
#pragma unroll
        for (int it = 0; it < LOCAL_ITEMS; ++it) if (it < nit) {
            const bool ok = base_w + it * 64 + lane < cnt;
            const uint32_t d = (kh[it] >> shift) & (RADIX - 1);
            const uint64_t m = digit_match(d, ok);
            rk[it] = __popcll(m & lt_mask);
            lead[it] = __ffsll((unsigned long long)m) - 1;
            old[it] = 0;
            if (ok && lane == lead[it]) old[it] = atomicAdd(&wcnt[w][d], (uint32_t)__popcll(m));   // result not needed before the sweep below
        }
#pragma unroll
        for (int it = 0; it < LOCAL_ITEMS; ++it) if (it < nit) rk[it] += __shfl(old[it], lead[it]);
        __syncthreads();
        // thread = digit: exclusive offsets of (digit, wave) in digit-major order
        uint32_t dtot = 0, dincl = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int ww = 0; ww < LOCAL_WAVES; ++ww) { const uint32_t c = wcnt[ww][tid]; wcnt[ww][tid] = dtot; dtot += c; }
            dincl = dtot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(dincl, o); if (lane >= o) dincl += t; }
            if (lane == 63) s_wsum[w] = dincl;
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t dbase = dincl - dtot;
            for (int ww = 0; ww < w; ++ww) dbase += s_wsum[ww];
#pragma unroll
            for (int ww = 0; ww < LOCAL_WAVES; ++ww) wcnt[ww][tid] += dbase;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < LOCAL_ITEMS; ++it) if (it < nit) {
            if (base_w + it * 64 + lane < cnt) {
                const uint32_t d = (kh[it] >> shift) & (RADIX - 1);
                sitem[wcnt[w][d] + rk[it]] = make_uint2(kh[it], ix[it]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < LOCAL_ITEMS; ++it) if (it < nit) {
            const uint32_t j = base_w + it * 64 + lane;
            if (j < cnt) { const uint2 t = sitem[j]; kh[it] = t.x; ix[it] = t.y; }
        }
        __syncthreads();                                                   // sitem / wcnt are rewritten by the next pass
    }
    // The window is now in stable order by the high key half.  The fix-up hop (below: every key placed inside its run of
    // equal high halves by its low half) needs only the run's other keys, and a run lies inside ONE window (equal high
    // halves have equal top bits): whole keys go to LDS, every item finds its final position there, and the kernel
    // writes keys, permutation and (emit) the leaf of that position -- no k_sort_fixup_fill launch after it.
    // (all gathers issued -- unconditionally, a lane without an item fetches the window's first key -- before the first is used)
    // (into FRESH registers: a value that was defined before the `if` needs a copy inside it -- which waits for the load)
    uint32_t tv[LOCAL_ITEMS], lowk[LOCAL_ITEMS], high[LOCAL_ITEMS];
#pragma unroll
    for (int it = 0; it < LOCAL_ITEMS; ++it) tv[it] = 0u;    // (a valid value everywhere: what emit fetches below is fetched unconditionally)
#pragma unroll
    for (int it = 0; it < LOCAL_ITEMS; ++it) if (it < nit) {
        const uint32_t j = base_w + it * 64 + lane;
        const size_t src = (size_t)lo + (j < cnt ? ix[it] : 0u);
        const uint2 k = reinterpret_cast<const uint2 *>(keys_in)[src];
        tv[it] = vals_in[src];
        high[it] = k.y;                                  // the true high half (kh[] held the composite)
        lowk[it] = k.x;
    }
#pragma unroll
    for (int it = 0; it < LOCAL_ITEMS; ++it) if (it < nit) {
        const uint32_t j = base_w + it * 64 + lane;
        if (j < cnt) sitem[j] = make_uint2(lowk[it], high[it]);
    }
    __syncthreads();
    // (in chunks of 5 items -- a window of the nominal size has 4 per lane: what emit gathers for an item is requested
    //  before the LDS search of the chunk, and lands while it runs)
    constexpr int CH = LOCAL_ITEMS / 2;
#pragma unroll
    for (int c0 = 0; c0 < LOCAL_ITEMS; c0 += CH) if (c0 < nit) {
        typename Emit::Payload pl[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) pl[u] = emit.load(tv[c0 + u]);      // (unconditional, one basic block: all of a chunk's gathers go out together; a lane or round without an item fetches a valid dummy)
#pragma unroll
        for (int u = 0; u < CH; ++u) if (c0 + u < nit) {
            const int it = c0 + u;
            const uint32_t j = base_w + it * 64 + lane;
            if (j < cnt) {
                const uint32_t low = lowk[it];
                uint32_t pos;
                // run too long: flag it (the host redoes the sort with 8 passes) but still emit a valid permutation and valid leaves
                if (!fixup_position_lds(sitem, cnt, j, low, high[it], pos)) { atomicOr(overflow, SORTF_FIXUP); pos = j; }
                keys_out[lo + pos] = ((uint64_t)high[it] << 32) | low; vals_out[lo + pos] = tv[it];
                emit.store(lo + pos, tv[it], pl[u]);
            }
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

// The same count over a window held in LDS as {low half, high half} (k_local_sort's epilogue).
__device__ __forceinline__ bool fixup_position_lds(const uint2 *item, uint32_t cnt, uint32_t j, uint32_t low, uint32_t high, uint32_t &pos)
{
    uint32_t back = 0, before = 0, fwd = 0;
    while (back < (uint32_t)FIX_MAX && back < j) {
        const uint2 b = item[j - 1 - back];
        if (b.y != high) break;
        before += b.x <= low;
        ++back;
    }
    while (fwd < (uint32_t)FIX_MAX && j + 1 + fwd < cnt) {
        const uint2 f = item[j + 1 + fwd];
        if (f.y != high) break;
        before += f.x < low;
        ++fwd;
    }
    pos = j - back + before;
    return back < (uint32_t)FIX_MAX && fwd < (uint32_t)FIX_MAX;
}

}  // namespace cd
