// mi355rt.hip -- C-ABI implementation of include/mi355rt.h (libmi355rt.so), gfx950 only.
// The per-pixel ray-sphere loop of RayTracing/anime_ray.cu:41-88 + Sphere::hit (sphere.cuh:34-44).
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 (float math must not be contracted: the hit
// predicate dx*dx + dy*dy < r*r and the two sqrtf / one divide decide pixel bytes).
#include "../../include/mi355rt.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstring>
#include <new>

namespace {

#define HIPCHK(expr)                                  \
    do {                                              \
        hipError_t e_ = (expr);                       \
        if (e_ != hipSuccess) return -(int)e_;        \
    } while (0)

constexpr float RT_INF = 2e10f;        // sphere.cuh:10
constexpr int TILE = 64;               // pixels per tile edge; one 256-thread workgroup per tile
constexpr int THREADS = 256;

// Per-sphere values that Sphere::hit recomputes for every pixel but that depend on the sphere only.
// Each is produced by exactly the float operation of the reference, so reusing it is bit-identical:
//   cx = x + (float)x_shift   (sphere.cuh:36: `x + x_shift`, int promoted to float)
//   rr = radius * radius      (sphere.cuh:38,39,40)
//   sr = sqrtf(radius*radius) (sphere.cuh:40)
struct alignas(16) SphGeom { float cx, cy, rr, z; };
struct alignas(16) SphShade { float r, g, b, sr; };

__device__ __forceinline__ void prepare_one(const RtSphere *__restrict__ s, const int32_t *__restrict__ shifts, int i, SphGeom &g, SphShade &h)
{
    const RtSphere sp = s[i];
    // sphere.cuh:35: the shift row is the sphere's idx field; spheres normally carry idx == their position, so that row is
    // fetched beside the sphere instead of after it (one memory round trip less) and only fetched again when idx differs
    int2 sh = reinterpret_cast<const int2 *>(shifts)[2 * i];
    if (sp.idx != i) sh = reinterpret_cast<const int2 *>(shifts)[2 * sp.idx];
    const int xs = sh.x, ys = sh.y;
    g.cx = sp.x + (float)xs; g.cy = sp.y + (float)ys; g.rr = sp.radius * sp.radius; g.z = sp.z;
    h.r = sp.r; h.g = sp.g; h.b = sp.b; h.sr = sqrtf(sp.radius * sp.radius);
}

// Per pixel only the winner is tracked -- its index, its dz and the running maximum of t: the normalisation
// n = dz / sqrtf(r*r) and the three colour products (sphere.cuh:41, anime_ray.cu:77-80) depend on nothing but the
// winning sphere, so they are evaluated once per pixel after the loop instead of once per hit (same operands, same
// operations, same bits).  That takes the IEEE division and the shade fetch out of the divergent hit branch.
// (Round 5 tried a LAZY depth -- a pixel's first hit leaves the square root's argument behind, sqrtf runs when a second sphere covers the pixel or in the
//  epilogue: pixel-exact, and 1.6 us per frame SLOWER, the second-hit path runs for a whole 8x8 block as soon as two discs meet in it;
//  profiles/r05_experiments/rt_render.log.)
struct Px { float maxz, dz; int win; };

// One sphere against one pixel: sphere.cuh:36-43 + anime_ray.cu:75-81.
__device__ __forceinline__ void shade_one(Px &p, float ox, float oy, const SphGeom g, int i)
{
    const float dx = ox - g.cx;
    const float dy = oy - g.cy;
    const float dx2 = dx * dx, dy2 = dy * dy;
    if (dx2 + dy2 < g.rr) {
        const float dz = sqrtf(g.rr - dx2 - dy2);
        const float t = dz + g.z;
        // anime_ray.cu:75 is a strict > in ascending sphere order, i.e. the lowest index wins a tie; stated
        // explicitly so that the spheres may arrive in any order (binned lists are unordered)
        // (a branch-free form of this -- bitwise | and &, three selects -- measured the same: 30.6 vs 30.8 us per frame)
        if (t > p.maxz || (t == p.maxz && i < p.win)) { p.maxz = t; p.dz = dz; p.win = i; }
    }
}

__device__ __forceinline__ uint32_t pack_px(const Px &p, const SphShade *__restrict__ shade)
{
    // a pixel no sphere covers: r = g = b = 0 (anime_ray.cu:68) -> bytes 0, 0, 0, 255 -- nothing to compute, and most
    // pixels of a frame are such (the compiler skips the block below for a whole wave when none of its lanes has a hit)
    uint32_t out = 255u << 24;
    if (p.win >= 0) {
        const SphShade h = shade[p.win];
        const float n = p.dz / h.sr;                                                   // sphere.cuh:41
        const float r = h.r * n, g = h.g * n, b = h.b * n;                             // anime_ray.cu:77-79
        // anime_ray.cu:84-87: (int)(c * 255) stored to unsigned char; alpha 255
        const uint32_t ri = (uint32_t)(unsigned char)(int)(r * 255);
        const uint32_t gi = (uint32_t)(unsigned char)(int)(g * 255);
        const uint32_t bi = (uint32_t)(unsigned char)(int)(b * 255);
        out = ri | (gi << 8) | (bi << 16) | (255u << 24);
    }
    return out;
}


// Exact conservative cull of one sphere against the pixel rectangle [X0,X1] x [Y0,Y1] (inclusive).  For a column x
// of the rectangle dx(x) = fl(ox(x) - cx) is monotone in x, so over the rectangle |dx| >= m where m = dx(X0) if that
// is > 0, -dx(X1) if dx(X1) < 0, else 0; likewise y.  fl is monotone, hence fl(dx^2 + dy^2) >= fl(mx^2 + my^2) for
// every pixel: if that lower bound is not < rr, the reference's own predicate (sphere.cuh:38) is false on the whole
// rectangle.  A larger rectangle has smaller or equal m: whatever survives for a tile survives for its super-tile.
// NaN anywhere -> keep (the exact per-pixel test then rejects).
__device__ __forceinline__ bool may_touch(const SphGeom g, float ox0, float ox1, float oy0, float oy1)
{
    const float dx0 = ox0 - g.cx, dx1 = ox1 - g.cx;
    const float dy0 = oy0 - g.cy, dy1 = oy1 - g.cy;
    const float mx = dx0 > 0.f ? dx0 : (dx1 < 0.f ? dx1 : 0.f);
    const float my = dy0 > 0.f ? dy0 : (dy1 < 0.f ? dy1 : 0.f);
    return !(mx * mx + my * my >= g.rr);
}

constexpr int SUPER = 256;             // super-tile edge in pixels (4 x 4 tiles)
#ifndef RT_SPLIT
#define RT_SPLIT 2                     // binned mode: workgroups per 64x64 tile (2: 64 x 32 pixels each)
#endif
constexpr int TILE_CAP = 48;           // entries a tile's own list holds; a tile that is touched by more spheres falls back to its super-tile's list

// What a pixel loop needs from one sphere, 32 bytes: the hit geometry, the colour and the index (for the tie rule).
// sqrtf(rr) is recomputed where needed: it is the very operation of prepare_one on the very same operand.
struct alignas(16) TileEnt { float cx, cy, rr, z; float r, g, b; int idx; };

// The binning (bin_sphere, called by k_prepare for the sphere it has just prepared) is sphere-centric: every SPHERE
// appends itself to the lists of the 64x64 tiles it may touch (a handful: a conservative index range, then the exact
// may_touch) as a complete TileEnt, so that a tile's pixel loop is ONE dependent fetch away from its survivors; and to
// the index list of every 256x256 super-tile it may touch, which a tile with more than TILE_CAP survivors walks
// instead.  One atomic per (sphere, list).  List order is arbitrary: the tie rule of anime_ray.cu:75 ("strict >, so the
// lowest index wins a tie in t") is applied explicitly in shade_one instead of through the processing order.
__device__ __forceinline__ void bin_sphere(const SphGeom g, const SphShade h, int i, int n, int dim,
                                           int c_shift_x, int c_shift_y, int nsx, int ty0, int ty1,
                                           int *__restrict__ super_list, int *__restrict__ super_count,
                                           TileEnt *__restrict__ tile_list, int *__restrict__ tile_count)
{
    const double sr = (double)h.sr, cx = (double)g.cx, cy = (double)g.cy;
    // pixel x sees ox = x - dim/2 + c_shift_x; the sphere can only touch |ox - cx| < sr (+ float rounding: margin)
    const double mx = 2.0 + 1e-6 * (fabs(cx) + sr), my = 2.0 + 1e-6 * (fabs(cy) + sr);
    double xlo = cx - sr - mx + dim / 2 - c_shift_x, xhi = cx + sr + mx + dim / 2 - c_shift_x;
    double ylo = cy - sr - my + dim / 2 - c_shift_y, yhi = cy + sr + my + dim / 2 - c_shift_y;
    const int ntx = dim / TILE;
    const double Y0 = (double)ty0 * TILE, Y1 = (double)ty1 * TILE - 1.0;     // the rows being rendered
    if (!(xlo == xlo && xhi == xhi && ylo == ylo && yhi == yhi)) { xlo = 0.0; xhi = (double)(dim - 1); ylo = Y0; yhi = Y1; }   // NaN anywhere: keep the full range (may_touch keeps NaN too)
    if (xhi < 0.0 || yhi < Y0 || xlo > (double)(dim - 1) || ylo > Y1) return;
    const int px0 = (int)fmax(xlo, 0.0), px1 = (int)fmin(xhi, (double)(dim - 1)), py0 = (int)fmax(ylo, Y0), py1 = (int)fmin(yhi, Y1);
    const TileEnt ent{g.cx, g.cy, g.rr, g.z, h.r, h.g, h.b, i};
    // The list positions come from returning atomics, a microsecond each: the tiles' and the super-tiles' are issued
    // together, eight at a time (a sphere of the reference's sizes touches at most 2 x 2 tiles and 2 x 2 super-tiles), and
    // waited for once.  Entries 0-3 of a batch are tiles, 4-7 super-tiles.
    int pend[8], npt = 0, nps = 0;
    auto flush = [&]() {
        int pos[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) pos[k] = k < npt ? atomicAdd(&tile_count[pend[k]], 1) : TILE_CAP;
#pragma unroll
        for (int k = 0; k < 4; ++k) pos[4 + k] = k < nps ? atomicAdd(&super_count[pend[4 + k]], 1) : -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (pos[k] < TILE_CAP) tile_list[(size_t)pend[k] * TILE_CAP + pos[k]] = ent;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (pos[4 + k] >= 0) super_list[(size_t)pend[4 + k] * n + pos[4 + k]] = i;
        npt = nps = 0;
    };
    const int tx0 = px0 / TILE, tx1 = px1 / TILE, tyA = py0 / TILE, tyB = py1 / TILE;
    const int sx0 = px0 / SUPER, sx1 = px1 / SUPER, syA = py0 / SUPER, syB = py1 / SUPER;
    int tx = tx0, ty = tyA, sx = sx0, sy = syA;
    bool more_t = true, more_s = true;
    while (more_t || more_s) {
        while (more_t && npt < 4) {
            const int X0 = tx * TILE, Yt = ty * TILE;
            const float ox0 = (float)(X0 - dim / 2 + c_shift_x), ox1 = (float)(X0 + TILE - 1 - dim / 2 + c_shift_x);
            const float oy0 = (float)(Yt - dim / 2 + c_shift_y), oy1 = (float)(Yt + TILE - 1 - dim / 2 + c_shift_y);
            if (may_touch(g, ox0, ox1, oy0, oy1)) pend[npt++] = ty * ntx + tx;
            if (++tx > tx1) { tx = tx0; if (++ty > tyB) more_t = false; }
        }
        while (more_s && nps < 4) {
            const int X0 = sx * SUPER, Ys = sy * SUPER;
            const int X1 = min(X0 + SUPER, dim) - 1, Ye = min(Ys + SUPER, dim) - 1;
            const float ox0 = (float)(X0 - dim / 2 + c_shift_x), ox1 = (float)(X1 - dim / 2 + c_shift_x);
            const float oy0 = (float)(Ys - dim / 2 + c_shift_y), oy1 = (float)(Ye - dim / 2 + c_shift_y);
            if (may_touch(g, ox0, ox1, oy0, oy1)) pend[4 + nps++] = sy * nsx + sx;
            if (++sx > sx1) { sx = sx0; if (++sy > syB) more_s = false; }
        }
        flush();
    }
}

// The per-sphere prepass; in binned mode (tile_list != nullptr) the same thread also bins its sphere, and the launch
// zeroes the list counters of the NEXT frame (two sets, used alternately: no memset between frames).
constexpr int PREP_THREADS = 64;       // one wave per workgroup: 4096 spheres spread over 64 CUs instead of 16
struct PrepArgs {
    const RtSphere *s; const int32_t *shifts; int n;
    SphGeom *geom; SphShade *shade;
    int dim, c_shift_x, c_shift_y, nsx, ty0, ty1;
    int *super_list, *super_count; TileEnt *tile_list; int *tile_count;
    int *next_counts; int n_counts;
};
__device__ __forceinline__ void prepare_and_bin(const PrepArgs &a, int i)
{
    SphGeom g; SphShade h;
    prepare_one(a.s, a.shifts, i, g, h);
    a.geom[i] = g; a.shade[i] = h;
    if (a.tile_list) bin_sphere(g, h, i, a.n, a.dim, a.c_shift_x, a.c_shift_y, a.nsx, a.ty0, a.ty1, a.super_list, a.super_count, a.tile_list, a.tile_count);
}
__global__ __launch_bounds__(PREP_THREADS) void k_prepare(PrepArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (a.next_counts) for (int j = i; j < a.n_counts; j += gridDim.x * blockDim.x) a.next_counts[j] = 0;
    if (i >= a.n) return;
    prepare_and_bin(a, i);
}

// Thread layout inside a 64x64 tile: tx = tid & 15 owns 4 consecutive columns, ty = tid >> 4 owns rows
// ty, ty+16, ty+32, ty+48 -> each row of the tile is written by 16 lanes x 16 B = 256 contiguous bytes.
//
// BINNED == false: anime_ray.cu:70-82 verbatim -- every pixel loops over all spheres.  The sphere index
// is wave-uniform, so the geometry comes through the scalar cache (s_load), not LDS.
// BINNED == true: pixels loop over the tile's own list (k_prepare, above) -- identical pixels, ~S / (survivors) times
// fewer hit() evaluations.  The list is the same for the whole workgroup, so its entries come through the scalar
// cache too: count -> entries -> pixels is the whole dependency chain, no LDS, no barrier.  A tile with more than
// TILE_CAP survivors walks its super-tile's index list instead, culling with may_touch (workgroup-uniform) as it goes:
// any number of spheres works, only slower.
struct RenderArgs {
    const SphGeom *geom; const SphShade *shade; int n;
    int dim, c_shift_x, c_shift_y, tile_y0;
    uint32_t *rgba; uint32_t *tile_tests;
    const int *super_list, *super_count; int nsx;
    const TileEnt *tile_list; const int *tile_count;
};
template <bool BINNED>
__device__ __forceinline__ void render_tile(const RenderArgs &ra, const int bx, const int by)
{
    const SphGeom *__restrict__ geom = ra.geom; const SphShade *__restrict__ shade = ra.shade; const int n = ra.n;
    const int dim = ra.dim, c_shift_x = ra.c_shift_x, c_shift_y = ra.c_shift_y, tile_y0 = ra.tile_y0;
    uint32_t *__restrict__ rgba = ra.rgba; uint32_t *__restrict__ tile_tests = ra.tile_tests;
    const int *__restrict__ super_list = ra.super_list, *__restrict__ super_count = ra.super_count; const int nsx = ra.nsx;
    const TileEnt *__restrict__ tile_list = ra.tile_list; const int *__restrict__ tile_count = ra.tile_count;
    const int tid = threadIdx.x;
    // BINNED: a workgroup renders HALF a tile (64 x 32 pixels, blockIdx.y counts halves): 8 pixels per thread instead of 16
    // keep the kernel under 64 VGPRs (8 waves per SIMD instead of 5) and make the work items small against the frame's tail
    constexpr int SPLIT = BINNED ? RT_SPLIT : 1;                         // workgroups per tile
    constexpr int NA = 4 / SPLIT;                                        // rows of pixel slots per thread
    const int X0 = bx * TILE, Y0 = ((int)(by / SPLIT) + tile_y0) * TILE;
    const int HY = (int)(by % SPLIT) * (TILE / SPLIT);
    // BRUTE: tx = tid & 15 owns 4 consecutive columns, ty = tid >> 4 owns rows ty, ty+16, ty+32, ty+48 (every sphere
    // is tested against every pixel: the mapping only has to store well).
    // BINNED: what a wave pays for a sphere is decided by its pixels that are processed TOGETHER -- the sqrt / depth
    // branch of shade_one runs for all 64 lanes as soon as one of them is inside the disc.  So a wave owns one 32 x 16
    // REGION of the half tile, and its 8 pixel slots are the region's eight 8x8 BLOCKS (lane = one pixel of the
    // block): a sphere costs the blocks its disc touches (~area + perimeter) instead of every 64-pixel row segment it
    // crosses, a sphere that cannot touch the region is skipped by the whole wave (wave-uniform may_touch), and a block
    // no sphere covers skips the shading epilogue.  The region goes through LDS once at the end so that it is stored
    // as rows of 128 contiguous bytes.
    const int wv = tid >> 6, ln = tid & 63;
    const int qx = (wv & 1) * 32, qy = HY + (wv >> 1) * (8 * NA);
    const int tx = tid & 15, ty = tid >> 4;
    const int x = X0 + 4 * tx;
    float ox[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ox[k] = (float)((BINNED ? X0 + qx + 8 * k + (ln & 7) : x + k) - dim / 2 + c_shift_x);          // anime_ray.cu:65
    float oy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) oy[k] = (float)((BINNED ? Y0 + qy + 8 * k + (ln >> 3) : Y0 + ty + 16 * k) - dim / 2 + c_shift_y); // anime_ray.cu:66
    Px px[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) px[a][b] = Px{-RT_INF, 0.f, -1};                // anime_ray.cu:68-69

    if (!BINNED) {
        for (int i = 0; i < n; ++i) {
            const SphGeom g = geom[i];
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) shade_one(px[a][b], ox[b], oy[a], g, i);
        }
    } else {
        const int t = (Y0 / TILE) * (dim / TILE) + (X0 / TILE);
        const int cnt = tile_count[t];
        uint32_t mytests = 0;
        // this wave's quadrant, in ray coordinates
        const int QX0 = X0 + qx, QY0 = Y0 + qy;
        const float qx0 = (float)(QX0 - dim / 2 + c_shift_x), qx1 = (float)(QX0 + 31 - dim / 2 + c_shift_x);
        const float qy0 = (float)(QY0 - dim / 2 + c_shift_y), qy1 = (float)(QY0 + 8 * NA - 1 - dim / 2 + c_shift_y);
        if (cnt <= TILE_CAP) {
            const TileEnt *ents = tile_list + (size_t)t * TILE_CAP;                  // workgroup-uniform: scalar loads
            // (fetching the entries four at a time, the first four before the count is known: no change, 35.0 us)
            for (int k = 0; k < cnt; ++k) {
                const TileEnt e = ents[k];
                const SphGeom g{e.cx, e.cy, e.rr, e.z};
                if (!may_touch(g, qx0, qx1, qy0, qy1)) continue;             // (wave-uniform)
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) shade_one(px[a][b], ox[b], oy[a], g, e.idx);
            }
            mytests = (uint32_t)cnt;
        } else {
            const float ox0 = (float)(X0 - dim / 2 + c_shift_x), ox1 = (float)(X0 + TILE - 1 - dim / 2 + c_shift_x);
            const float oy0 = (float)(Y0 - dim / 2 + c_shift_y), oy1 = (float)(Y0 + TILE - 1 - dim / 2 + c_shift_y);
            const int st = (Y0 / SUPER) * nsx + (X0 / SUPER);
            const int *slist = super_list + (size_t)st * n;
            const int scount = super_count[st];
            for (int k = 0; k < scount; ++k) {
                const int i = slist[k];
                const SphGeom g = geom[i];
                if (!may_touch(g, ox0, ox1, oy0, oy1)) continue;                     // (workgroup-uniform)
                ++mytests;
                if (!may_touch(g, qx0, qx1, qy0, qy1)) continue;             // (wave-uniform)
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) shade_one(px[a][b], ox[b], oy[a], g, i);
            }
        }
        if (tid == 0 && HY == 0 && tile_tests) tile_tests[t] = mytests;                         // sphere tests per pixel of this tile (summed by the host)
    }
    if (BINNED) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        // pixel (8b + lx, 8a + ly) of the region -> LDS (row stride 36 words: the eight rows of a block land in different
        // banks), then every lane stores 4 consecutive pixels of rows ly and ly + 8.  Wave-private: no barrier.
        __shared__ uint32_t quad[THREADS / 64][8 * NA][36];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) quad[wv][8 * a + (ln >> 3)][8 * b + (ln & 7)] = pack_px(px[a][b], shade);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int r = 8 * k + (ln >> 3), cx = 4 * (ln & 7);
            const uint4 o = *reinterpret_cast<const uint4 *>(&quad[wv][r][cx]);
            // (a frame is written once and not read by this library: non-temporal stores -- 30.1 -> 29.3 us per frame)
            { const v4u ov = {o.x, o.y, o.z, o.w};
              __builtin_nontemporal_store(ov, reinterpret_cast<v4u *>(rgba + (size_t)(Y0 + qy + r) * dim + (X0 + qx + cx))); }   // offset = x + y*dim, anime_ray.cu:64
        }
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int y = Y0 + ty + 16 * a;
            uint4 o;
            o.x = pack_px(px[a][0], shade); o.y = pack_px(px[a][1], shade); o.z = pack_px(px[a][2], shade); o.w = pack_px(px[a][3], shade);
            *reinterpret_cast<uint4 *>(rgba + (size_t)y * dim + x) = o;                   // offset = x + y*dim, anime_ray.cu:64
        }
    }
}

template <bool BINNED>
__global__ __launch_bounds__(THREADS) void k_render(RenderArgs ra) { render_tile<BINNED>(ra, (int)blockIdx.x, (int)blockIdx.y); }

// ---------------------------------------------------------------- animation state (sphere.cuh:50-118)
// XORWOW as <curand_kernel.h> implements it for curand_init(seed, 0, 0): no skip-ahead, seed scrambled by two odd
// multipliers, five xorshift words + a Weyl counter.  Restated from the published algorithm (the header is not in the
// reference tree); see oracle/rt_oracle.c for the parity status of the random stream.
struct Xorwow { uint32_t v[5]; uint32_t d; };

__device__ __forceinline__ void xorwow_init(Xorwow &s, unsigned long long seed)
{
    const uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    s.d = 6615241u + t1 + t0;
    s.v[0] = 123456789u + t0; s.v[1] = 362436069u ^ t0; s.v[2] = 521288629u + t1; s.v[3] = 88675123u ^ t1; s.v[4] = 5783321u + t0;
}
__device__ __forceinline__ uint32_t xorwow_next(Xorwow &s)
{
    const uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3]; s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v[4] + s.d;
}
// dev_rnd(x, s), sphere.cuh:26
__device__ __forceinline__ double dev_rnd(int x, Xorwow &s) { return xorwow_next(s) % 1000000u * 1.0 / 1000000 * x; }

constexpr double ANIM_PI = 3.1415926535898;                           // sphere.cuh:19

// The reference launches these with <<<128, 1>>> and a block-stride loop; one thread per sphere gives the same state.
__global__ __launch_bounds__(256) void k_anim_init(int n, Xorwow *__restrict__ st, int32_t *__restrict__ shifts, double *__restrict__ angles)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xorwow s; xorwow_init(s, (unsigned long long)i);                   // curand_init(i, 0, 0, &states[i]), sphere.cuh:53
    st[i] = s;
    shifts[4 * i] = shifts[4 * i + 1] = 0;
    shifts[4 * i + 2] = (i % 5 + 1) * 5;
    shifts[4 * i + 3] = (i % 2) * 2 - 1;
    angles[i] = 0.0;
}

__device__ __forceinline__ void anim_axis_one(int i, Xorwow *__restrict__ st, int32_t *__restrict__ shifts, int shake_width)   // sphere.cuh:66-77
{
    Xorwow s = st[i];
    const int x_shift = (int)dev_rnd(shake_width, s);
    const int y_shift = (int)dev_rnd(shake_width, s);
    shifts[4 * i] = x_shift; shifts[4 * i + 1] = y_shift;
    st[i] = s;
}
__device__ __forceinline__ void anim_curve_one(int i, int32_t *__restrict__ shifts, double *__restrict__ angles)              // sphere.cuh:82-97
{
    const int speed = shifts[4 * i + 2];
    const float a = (float)angles[i];
    // correctly rounded float cosine / sine (through the double functions); CUDA's cosf / sinf are within 2 ulp of these
    const int x_shift = (int)((float)speed * (float)cos((double)a));
    const int y_shift = (int)((float)speed * (float)sin((double)a));
    shifts[4 * i] += x_shift; shifts[4 * i + 1] += y_shift;
    angles[i] = fmod(angles[i] + ANIM_PI / 12 * shifts[4 * i + 3], 2 * ANIM_PI);
}
__device__ __forceinline__ void anim_speed_angle_one(int i, Xorwow *__restrict__ st, int32_t *__restrict__ shifts, double *__restrict__ angles,
                                                     int update_prob, int max_speed)                                            // sphere.cuh:102-118
{
    Xorwow s = st[i];
    const int p = (int)dev_rnd(10, s);
    if (p < update_prob) {
        shifts[4 * i + 2] = (int)dev_rnd(max_speed, s);
        shifts[4 * i + 3] = ((int)dev_rnd(2, s)) * 2 - 1;
        angles[i] = fmod(angles[i] + ((int)dev_rnd(2, s)) * ANIM_PI, 2 * ANIM_PI);
    }
    st[i] = s;
}

__global__ __launch_bounds__(256) void k_anim_axis(int n, Xorwow *__restrict__ st, int32_t *__restrict__ shifts, int shake_width)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) anim_axis_one(i, st, shifts, shake_width);
}
__global__ __launch_bounds__(256) void k_anim_curve(int n, int32_t *__restrict__ shifts, double *__restrict__ angles)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) anim_curve_one(i, shifts, angles);
}
__global__ __launch_bounds__(256) void k_anim_speed_angle(int n, Xorwow *__restrict__ st, int32_t *__restrict__ shifts, double *__restrict__ angles,
                                                          int update_prob, int max_speed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) anim_speed_angle_one(i, st, shifts, angles, update_prob, max_speed);
}

// ---------------------------------------------------------------- the animation loop, one launch per frame (round 5)
// generate_frame (anime_ray.cu:98-140) runs, per frame: [the spheres' shifts move] -> kernel -> D2H.  Frame f + 1's shifts and lists depend on nothing
// frame f's pixels do, so ONE launch renders frame f (its own list set) while the workgroups of its first grid row move the spheres on to frame f + 1
// and prepare + bin them into the OTHER list set: k_prepare's 7 us -- three dependent round trips around a kernel boundary, a quarter of a frame --
// run beside 8192 rendering workgroups instead of in front of them.  No data crosses workgroups inside the launch: the render role reads set f & 1 and
// the counters f % 3, the prepare role writes set (f + 1) & 1 and the counters (f + 1) % 3 and zeroes the counters (f + 2) % 3 (last read by frame f - 1's
// render, a launch ago).  A sphere's state is its thread's alone (idx == position is required: rt_anim_loop checks), so moving it and preparing it in one
// thread is the reference's kernel sequence, sphere by sphere.
struct AnimStep { int shake /* 0: none, 1: updateSphereShiftsWithAxisMove, 2: ...WithCurveMove + updateSphereCurveSpeedAngle */, p0, p1, p2;
                  Xorwow *st; int32_t *shifts; double *angles; };
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_frame(RenderArgs ra, PrepArgs pa, AnimStep an, int prep /* 0: the last frame of the loop, nothing to prepare */)
{
    if (blockIdx.y == 0) {                                              // (workgroup-uniform) the prepare row
        if (!prep) return;
        const int gtid = blockIdx.x * THREADS + threadIdx.x, gsz = gridDim.x * THREADS;
        for (int j = gtid; j < pa.n_counts; j += gsz) pa.next_counts[j] = 0;
        for (int i = gtid; i < pa.n; i += gsz) {
            if (an.shake == 1) anim_axis_one(i, an.st, an.shifts, an.p0);
            else if (an.shake == 2) { anim_curve_one(i, an.shifts, an.angles); anim_speed_angle_one(i, an.st, an.shifts, an.angles, an.p1, an.p2); }
            prepare_and_bin(pa, i);
        }
        return;
    }
    render_tile<true>(ra, (int)blockIdx.x, (int)blockIdx.y - 1);
}
// The loop's first frame has nobody to prepare it: the spheres move and are prepared by a launch of their own.
__global__ __launch_bounds__(THREADS) void k_anim_prepare(PrepArgs pa, AnimStep an)
{
    const int gtid = blockIdx.x * THREADS + threadIdx.x, gsz = gridDim.x * THREADS;
    for (int j = gtid; j < pa.n_counts; j += gsz) pa.next_counts[j] = 0;
    for (int i = gtid; i < pa.n; i += gsz) {
        if (an.shake == 1) anim_axis_one(i, an.st, an.shifts, an.p0);
        else if (an.shake == 2) { anim_curve_one(i, an.shifts, an.angles); anim_speed_angle_one(i, an.st, an.shifts, an.angles, an.p1, an.p2); }
        prepare_and_bin(pa, i);
    }
}

}  // namespace

struct rt_ctx {
    int32_t n = 0, dim = 0;
    int mode = RT_MODE_BINNED;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    RtSphere *d_spheres = nullptr; int32_t *d_shifts = nullptr;
    Xorwow *d_rng = nullptr; double *d_angles = nullptr; bool anim_ready = false;   // animation state, sphere.cuh:50-61
    // What a frame's render reads is a SET, two of them used alternately (frame & 1): the animation loop prepares frame f + 1 while frame f renders (k_frame)
    SphGeom *d_geom[2] = {nullptr, nullptr}; SphShade *d_shade[2] = {nullptr, nullptr};
    uint32_t *d_rgba = nullptr;
    int *d_super_list[2] = {nullptr, nullptr};                  // [nsuper][n] indices of the spheres that may touch a super-tile (unordered)
    TileEnt *d_tile_list[2] = {nullptr, nullptr};               // [ntiles][TILE_CAP]
    int *d_counts[3] = {nullptr, nullptr, nullptr};             // three sets of [ntiles tile counts | nsuper super-tile counts] (frame % 3): a frame's prepare zeroes
    int n_counts = 0, ntiles = 0; uint32_t frame = 0;           //   the set of the next one -- which, in the loop, the frame BEFORE is no longer reading
    bool idx_identity = true;                                   // every sphere's idx is its position (what the loop's fused launch needs)
    uint32_t *d_tile_tests = nullptr, *h_tile_tests = nullptr;  // sphere tests per pixel of every tile; pinned copy
    rt_stats stats = {};
};

namespace {
void rt_free(rt_ctx *c)
{
    hipFree(c->d_rng); hipFree(c->d_angles);
    hipFree(c->d_spheres); hipFree(c->d_shifts); hipFree(c->d_geom[0]); hipFree(c->d_shade[0]); hipFree(c->d_rgba); hipFree(c->d_super_list[0]);
    hipFree(c->d_tile_list[0]); hipFree(c->d_counts[0]); hipFree(c->d_tile_tests);
    if (c->h_tile_tests) hipHostFree(c->h_tile_tests);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->stream) hipStreamDestroy(c->stream);
}
}  // namespace

extern "C" {

const char *rt_version(void) { return "mi355rt 0.1 gfx950"; }

int rt_create(rt_ctx **out, const RtSphere *spheres, int32_t n_spheres, int32_t dim)
{
    if (!out || !spheres || n_spheres <= 0 || dim <= 0 || dim % TILE) return RT_ERR_ARG;
    for (int i = 0; i < n_spheres; ++i) if (spheres[i].idx < 0 || spheres[i].idx >= n_spheres) return RT_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return RT_ERR_NO_DEVICE;
    rt_ctx *c = new (std::nothrow) rt_ctx();
    if (!c) return RT_ERR_ARG;
    c->n = n_spheres; c->dim = dim;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess && r != hipSuccess) e = r; };
    ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ok(hipEventCreate(&c->ev0)); ok(hipEventCreate(&c->ev1));
    ok(hipMalloc(&c->d_spheres, sizeof(RtSphere) * (size_t)n_spheres));
    ok(hipMalloc(&c->d_shifts, sizeof(int32_t) * 4 * (size_t)n_spheres));
    ok(hipMalloc(&c->d_rng, sizeof(Xorwow) * (size_t)n_spheres));
    ok(hipMalloc(&c->d_angles, sizeof(double) * (size_t)n_spheres));
    ok(hipMalloc(&c->d_geom[0], sizeof(SphGeom) * 2 * (size_t)n_spheres));
    ok(hipMalloc(&c->d_shade[0], sizeof(SphShade) * 2 * (size_t)n_spheres));
    if (e == hipSuccess) { c->d_geom[1] = c->d_geom[0] + n_spheres; c->d_shade[1] = c->d_shade[0] + n_spheres; }
    ok(hipMalloc(&c->d_rgba, sizeof(uint32_t) * (size_t)dim * dim));
    for (int i = 0; i < n_spheres; ++i) c->idx_identity = c->idx_identity && spheres[i].idx == i;
    { const size_t ns = (size_t)((dim + SUPER - 1) / SUPER) * ((dim + SUPER - 1) / SUPER);
      c->ntiles = (dim / TILE) * (dim / TILE);
      c->n_counts = c->ntiles + (int)ns;
      ok(hipMalloc(&c->d_counts[0], sizeof(int) * 3 * (size_t)c->n_counts));
      if (e == hipSuccess) { c->d_counts[1] = c->d_counts[0] + c->n_counts; c->d_counts[2] = c->d_counts[1] + c->n_counts; ok(hipMemset(c->d_counts[0], 0, sizeof(int) * 3 * (size_t)c->n_counts)); }
      ok(hipMalloc(&c->d_tile_list[0], sizeof(TileEnt) * 2 * (size_t)c->ntiles * TILE_CAP));
      if (e == hipSuccess) c->d_tile_list[1] = c->d_tile_list[0] + (size_t)c->ntiles * TILE_CAP;
      ok(hipMalloc(&c->d_tile_tests, sizeof(uint32_t) * (size_t)c->ntiles));
      ok(hipHostMalloc(&c->h_tile_tests, sizeof(uint32_t) * (size_t)c->ntiles, hipHostMallocDefault));
      ok(hipMalloc(&c->d_super_list[0], sizeof(int) * 2 * ns * (size_t)n_spheres));
      if (e == hipSuccess) c->d_super_list[1] = c->d_super_list[0] + ns * (size_t)n_spheres; }
    if (e == hipSuccess) ok(hipMemcpy(c->d_spheres, spheres, sizeof(RtSphere) * (size_t)n_spheres, hipMemcpyHostToDevice));
    if (e != hipSuccess) { rt_free(c); delete c; return -(int)e; }
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_ctx *c)
{
    if (!c) return;
    hipStreamSynchronize(c->stream);
    rt_free(c);
    delete c;
}

int rt_set_spheres(rt_ctx *c, const RtSphere *spheres)
{
    if (!c || !spheres) return RT_ERR_ARG;
    for (int i = 0; i < c->n; ++i) if (spheres[i].idx < 0 || spheres[i].idx >= c->n) return RT_ERR_ARG;
    HIPCHK(hipMemcpy(c->d_spheres, spheres, sizeof(RtSphere) * (size_t)c->n, hipMemcpyHostToDevice));
    c->idx_identity = true;
    for (int i = 0; i < c->n; ++i) c->idx_identity = c->idx_identity && spheres[i].idx == i;
    return RT_OK;
}

int rt_set_mode(rt_ctx *c, int mode)
{
    if (!c || (mode != RT_MODE_BRUTE && mode != RT_MODE_BINNED)) return RT_ERR_ARG;
    c->mode = mode;
    return RT_OK;
}

// the arguments of frame number `frame`: list set frame & 1, counters frame % 3 (its prepare zeroes the counters of frame + 1)
static PrepArgs prep_args(rt_ctx *c, uint32_t frame, int32_t csx, int32_t csy, int ty0, int ty1)
{
    const int nsx = (c->dim + SUPER - 1) / SUPER, k = (int)(frame & 1u);
    int *cur = c->d_counts[frame % 3u], *nxt = c->d_counts[(frame + 1u) % 3u];
    return PrepArgs{c->d_spheres, c->d_shifts, c->n, c->d_geom[k], c->d_shade[k], c->dim, (int)csx, (int)csy, nsx, ty0, ty1,
                    c->d_super_list[k], cur + c->ntiles, c->d_tile_list[k], cur, nxt, c->n_counts};
}
static RenderArgs render_args(rt_ctx *c, uint32_t frame, int32_t csx, int32_t csy, int ty0)
{
    const int nsx = (c->dim + SUPER - 1) / SUPER, k = (int)(frame & 1u);
    const int *cur = c->d_counts[frame % 3u];
    return RenderArgs{c->d_geom[k], c->d_shade[k], c->n, c->dim, (int)csx, (int)csy, ty0, c->d_rgba, c->d_tile_tests,
                      c->d_super_list[k], cur + c->ntiles, nsx, c->d_tile_list[k], cur};
}

// `frames` > 1: the same frame that many times back to back (rt_render_repeat); time stamps on the first and the last kernel only
static int render_frames(rt_ctx *c, const int32_t *shifts4, int32_t csx, int32_t csy, int32_t y0, int32_t y1, uint8_t *rgba_out, int frames)
{
    if (!c || (!shifts4 && !c->anim_ready) || y0 < 0 || y1 > c->dim || y0 >= y1 || y0 % TILE || y1 % TILE || frames < 1) return RT_ERR_ARG;
    hipStream_t s = c->stream;
    if (shifts4) HIPCHK(hipMemcpyAsync(c->d_shifts, shifts4, sizeof(int32_t) * 4 * (size_t)c->n, hipMemcpyHostToDevice, s));   // NULL: the device-resident animation state
    // (the frame's time stamps ride on the dispatch packets of its first and last kernel: an event record of its own is a
    //  barrier packet, a few idle microseconds each)
    const dim3 grid(c->dim / TILE, (y1 - y0) / TILE), grid_binned(c->dim / TILE, RT_SPLIT * ((y1 - y0) / TILE));   // binned: a workgroup per half tile
    const int ty0 = y0 / TILE, ty1 = y1 / TILE, ntx = c->dim / TILE;
    for (int f = 0; f < frames; ++f) {
    hipEvent_t e0 = f == 0 ? c->ev0 : nullptr, e1 = f == frames - 1 ? c->ev1 : nullptr;
    if (c->mode == RT_MODE_BINNED) {
        const PrepArgs pa = prep_args(c, c->frame, csx, csy, ty0, ty1);
        const RenderArgs ra = render_args(c, c->frame, csx, csy, ty0);
        ++c->frame;
        hipExtLaunchKernelGGL(k_prepare, dim3((c->n + PREP_THREADS - 1) / PREP_THREADS), dim3(PREP_THREADS), 0u, s, e0, nullptr, 0u, pa);
        hipExtLaunchKernelGGL(k_render<true>, grid_binned, dim3(THREADS), 0u, s, nullptr, e1, 0u, ra);
    } else {
        PrepArgs pa = prep_args(c, c->frame, csx, csy, 0, 0);
        pa.super_list = nullptr; pa.super_count = nullptr; pa.tile_list = nullptr; pa.tile_count = nullptr; pa.next_counts = nullptr; pa.n_counts = 0;   // prepare only: no lists
        RenderArgs ra = render_args(c, c->frame, csx, csy, ty0);
        ra.tile_tests = nullptr; ra.super_list = nullptr; ra.super_count = nullptr; ra.tile_list = nullptr; ra.tile_count = nullptr; ra.nsx = 0;
        hipExtLaunchKernelGGL(k_prepare, dim3((c->n + PREP_THREADS - 1) / PREP_THREADS), dim3(PREP_THREADS), 0u, s, e0, nullptr, 0u, pa);
        hipExtLaunchKernelGGL(k_render<false>, grid, dim3(THREADS), 0u, s, nullptr, e1, 0u, ra);
    }
    }
    // from here on an early return must not leave a copy into caller / context memory in flight: synchronise first
    hipError_t e = hipGetLastError();
    const size_t t0 = (size_t)ty0 * ntx, nt = (size_t)(ty1 - ty0) * ntx;
    if (e == hipSuccess && c->mode == RT_MODE_BINNED)
        e = hipMemcpyAsync(c->h_tile_tests + t0, c->d_tile_tests + t0, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rgba_out)                                // anime_ray.cu:128-131 D2H of the frame
        e = hipMemcpyAsync(rgba_out, c->d_rgba + (size_t)y0 * c->dim, sizeof(uint32_t) * (size_t)(y1 - y0) * c->dim, hipMemcpyDeviceToHost, s);
    const hipError_t es = hipStreamSynchronize(s);
    if (e != hipSuccess) return -(int)e;
    if (es != hipSuccess) return -(int)es;
    unsigned long long tests = 0;
    if (c->mode == RT_MODE_BINNED) for (size_t i = 0; i < nt; ++i) tests += c->h_tile_tests[t0 + i];
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_render = ms / (float)frames;
    c->stats.mode = (uint32_t)c->mode;
    c->stats.sphere_tests = c->mode == RT_MODE_BINNED ? tests * (unsigned long long)(TILE * TILE) : (uint64_t)c->n * (uint64_t)c->dim * (uint64_t)(y1 - y0);
    return RT_OK;
}

int rt_render_rows(rt_ctx *c, const int32_t *shifts4, int32_t csx, int32_t csy, int32_t y0, int32_t y1, uint8_t *rgba_out)
{
    return render_frames(c, shifts4, csx, csy, y0, y1, rgba_out, 1);
}

int rt_render_repeat(rt_ctx *c, const int32_t *shifts4, int32_t csx, int32_t csy, int32_t frames, uint8_t *rgba_out)
{
    if (!c) return RT_ERR_ARG;
    return render_frames(c, shifts4, csx, csy, 0, c->dim, rgba_out, frames);
}

int rt_render(rt_ctx *c, const int32_t *shifts4, int32_t csx, int32_t csy, uint8_t *rgba_out)
{
    if (!c) return RT_ERR_ARG;
    return rt_render_rows(c, shifts4, csx, csy, 0, c->dim, rgba_out);
}

int rt_init_shifts(int32_t n, int32_t *shifts4, double *angles)
{
    if (n < 0 || !shifts4) return RT_ERR_ARG;
    for (int i = 0; i < n; ++i) {                                     // sphere.cuh:54-57
        shifts4[4 * i] = shifts4[4 * i + 1] = 0;
        shifts4[4 * i + 2] = (i % 5 + 1) * 5;
        shifts4[4 * i + 3] = (i % 2) * 2 - 1;
        if (angles) angles[i] = 0.0;
    }
    return RT_OK;
}

int rt_anim_init(rt_ctx *c)
{
    if (!c) return RT_ERR_ARG;
    k_anim_init<<<(c->n + 255) / 256, 256, 0, c->stream>>>(c->n, c->d_rng, c->d_shifts, c->d_angles);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    c->anim_ready = true;
    return RT_OK;
}

int rt_anim_axis_move(rt_ctx *c, int32_t shake_width)
{
    if (!c || !c->anim_ready) return RT_ERR_ARG;
    k_anim_axis<<<(c->n + 255) / 256, 256, 0, c->stream>>>(c->n, c->d_rng, c->d_shifts, shake_width);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

int rt_anim_curve_move(rt_ctx *c)
{
    if (!c || !c->anim_ready) return RT_ERR_ARG;
    k_anim_curve<<<(c->n + 255) / 256, 256, 0, c->stream>>>(c->n, c->d_shifts, c->d_angles);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

int rt_anim_update_speed_angle(rt_ctx *c, int32_t update_prob, int32_t max_speed)
{
    if (!c || !c->anim_ready) return RT_ERR_ARG;
    k_anim_speed_angle<<<(c->n + 255) / 256, 256, 0, c->stream>>>(c->n, c->d_rng, c->d_shifts, c->d_angles, update_prob, max_speed);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

int rt_anim_loop(rt_ctx *c, int32_t frames, int32_t shake, int32_t p0, int32_t p1, int32_t p2, int32_t csx, int32_t csy, uint8_t *rgba_last)
{
    if (!c || !c->anim_ready || frames < 1 || shake < 0 || shake > 2) return RT_ERR_ARG;
    if (c->mode != RT_MODE_BINNED || !c->idx_identity) {
        // brute mode, or a sphere whose shift row is another sphere's (sphere.cuh:35 with idx != position): the reference's launch sequence, frame by frame
        for (int f = 0; f < frames; ++f) {
            int rc = RT_OK;
            if (shake == 1) rc = rt_anim_axis_move(c, p0);
            else if (shake == 2) { rc = rt_anim_curve_move(c); if (rc == RT_OK) rc = rt_anim_update_speed_angle(c, p1, p2); }
            if (rc == RT_OK) rc = rt_render(c, nullptr, csx, csy, f == frames - 1 ? rgba_last : nullptr);
            if (rc != RT_OK) return rc;
        }
        return RT_OK;
    }
    hipStream_t s = c->stream;
    const AnimStep an{(int)shake, (int)p0, (int)p1, (int)p2, c->d_rng, c->d_shifts, c->d_angles};
    const int ty1 = c->dim / TILE;
    const dim3 grid(c->dim / TILE, RT_SPLIT * ty1 + 1);                 // row 0: the prepare role, the rest: a workgroup per half tile
    hipExtLaunchKernelGGL(k_anim_prepare, dim3(16), dim3(THREADS), 0u, s, c->ev0, nullptr, 0u, prep_args(c, c->frame, csx, csy, 0, ty1), an);
    for (int f = 0; f < frames; ++f) {
        const int more = f + 1 < frames;
        const RenderArgs ra = render_args(c, c->frame, csx, csy, 0);
        const PrepArgs pa = prep_args(c, c->frame + 1u, csx, csy, 0, ty1);
        ++c->frame;
        hipExtLaunchKernelGGL(k_frame, grid, dim3(THREADS), 0u, s, nullptr, f == frames - 1 ? c->ev1 : nullptr, 0u, ra, pa, an, more);
    }
    hipError_t e = hipGetLastError();
    const size_t nt = (size_t)c->ntiles;
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_tile_tests, c->d_tile_tests, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rgba_last) e = hipMemcpyAsync(rgba_last, c->d_rgba, sizeof(uint32_t) * (size_t)c->dim * c->dim, hipMemcpyDeviceToHost, s);
    const hipError_t es = hipStreamSynchronize(s);
    if (e != hipSuccess) return -(int)e;
    if (es != hipSuccess) return -(int)es;
    unsigned long long tests = 0;
    for (size_t i = 0; i < nt; ++i) tests += c->h_tile_tests[i];
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_render = ms / (float)frames;
    c->stats.mode = (uint32_t)c->mode;
    c->stats.sphere_tests = tests * (unsigned long long)(TILE * TILE);   // (the last frame's)
    return RT_OK;
}

int rt_anim_get_state(rt_ctx *c, int32_t *shifts4, double *angles, uint32_t *rng6)
{
    if (!c || !c->anim_ready) return RT_ERR_ARG;
    if (shifts4) HIPCHK(hipMemcpyAsync(shifts4, c->d_shifts, sizeof(int32_t) * 4 * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    if (angles) HIPCHK(hipMemcpyAsync(angles, c->d_angles, sizeof(double) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    if (rng6) HIPCHK(hipMemcpyAsync(rng6, c->d_rng, sizeof(Xorwow) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_get_stats(rt_ctx *c, rt_stats *out) { if (!c || !out) return RT_ERR_ARG; *out = c->stats; return RT_OK; }

}  // extern "C"
