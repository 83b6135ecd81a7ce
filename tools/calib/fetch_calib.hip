// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ_* on gfx950 for the access patterns of this project's kernels (MI355X_MICROARCH.md: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel reads a table that no cache holds
// (1 GiB; the last kernel's is read once too) EXACTLY once, so the useful bytes are known:
//   stream16   : 16 B per lane, coalesced (what k_morton / k_os_pass / k_build_block do with keys, leaves, vertices in order)
//   gather32   : each lane one 32-byte record at a pseudo-random index (a bijection), as two 16-byte loads: the descent's record halves
//   gather64   : each lane one 64-byte record (four 16-byte loads): a descent visit (left + right half)
//   gather24   : each lane three 8-byte loads of one 24-byte vertex at a random index: the vertex gathers of k_morton / k_build_block
//   sload32    : each WAVE one 32-byte record through the scalar cache (wave-uniform address): the shared chain of k_descend_half
// Build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip ; run under rocprofv3 --pmc ... (tools/calib/run_calib.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

__global__ void stream16(const float4 *__restrict__ t, size_t n, float *out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = t[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) *out = acc;
}
__device__ __forceinline__ size_t scramble(size_t i, size_t mask) { return (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask; }   // odd multiplier: a bijection mod 2^k
__global__ void gather32(const float4 *__restrict__ t, size_t nrec, float *out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = scramble(i, nrec - 1);
        const float4 a = t[2 * r], b = t[2 * r + 1]; acc += (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
    }
    if (acc == 12345.678f) *out = acc;
}
__global__ void gather64(const float4 *__restrict__ t, size_t nrec, float *out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = scramble(i, nrec - 1);
        const float4 a = t[4 * r], b = t[4 * r + 1], c = t[4 * r + 2], d = t[4 * r + 3]; acc += (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w) + (c.x + c.y) + (c.z + c.w) + (d.x + d.y) + (d.z + d.w);
    }
    if (acc == 12345.678f) *out = acc;
}
__global__ void gather24(const double *__restrict__ t, size_t nrec, float *out)
{
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = scramble(i, nrec - 1);                      // nrec is a power of two <= table size / 24
        acc += t[3 * r] + t[3 * r + 1] + t[3 * r + 2];
    }
    if (acc == 12345.678) *out = (float)acc;
}
__global__ void sload32(const int4 *__restrict__ t, size_t nrec, int *out)
{
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    int acc = 0;
    for (size_t i = wave; i < nrec; i += nwaves) {
        const size_t r = (size_t)__builtin_amdgcn_readfirstlane((int)scramble(i, nrec - 1));      // wave-uniform -> s_load_dwordx4 x 2
        const int4 a = t[2 * r], b = t[2 * r + 1]; acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    }
    if (acc == 0x12345678) *out = acc;
}

int main()
{
    const size_t bytes = 1ull << 30;
    void *t = nullptr; float *out = nullptr;
    CHK(hipMalloc(&t, bytes)); CHK(hipMalloc(&out, 64));
    CHK(hipMemset(t, 1, bytes));
    const int blocks = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        stream16<<<blocks, 256>>>((const float4 *)t, bytes / 16, out);
        gather32<<<blocks, 256>>>((const float4 *)t, bytes / 32, out);
        gather64<<<blocks, 256>>>((const float4 *)t, bytes / 64, out);
        gather24<<<blocks, 256>>>((const double *)t, (size_t)1 << 25, out);            // 2^25 vertices x 24 B = 768 MiB of the table
        sload32<<<blocks, 256>>>((const int4 *)t, (size_t)1 << 22, (int *)out);        // 2^22 records x 32 B = 128 MiB, one per wave-iteration (scattered over 128 MiB)
    }
    CHK(hipDeviceSynchronize());
    printf("useful bytes: stream16 %zu gather32 %zu gather64 %zu gather24 %zu sload32 %zu\n", bytes, bytes, bytes, (size_t)24 << 25, (size_t)32 << 22);
    return 0;
}
