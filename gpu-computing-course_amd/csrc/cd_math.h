// cd_math.h -- device arithmetic of the collision path: every FP64 operation is written in the
// reference's operand order and the TU is compiled with -ffp-contract=off, so each compare sees
// bit-identical operands to the reference's host twin (cpu.cuh) and to oracle/cd_oracle.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cd {

struct d3 { double x, y, z; };

// mathop.cuh:17-44 -- compare-select, NOT fmin/fmax (NaN-asymmetric on purpose)
__device__ __forceinline__ double fmax2(double a, double b) { return (a > b) ? a : b; }
__device__ __forceinline__ double fmin2(double a, double b) { return (a < b) ? a : b; }
__device__ __forceinline__ double fmax3(double a, double b, double c) { double t = a; if (b > t) t = b; if (c > t) t = c; return t; }
__device__ __forceinline__ double fmin3(double a, double b, double c) { double t = a; if (b < t) t = b; if (c < t) t = c; return t; }

// vec3f.cuh:100-103, 118-125
__device__ __forceinline__ d3 sub(const d3 a, const d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 neg(const d3 a) { return d3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ d3 cross(const d3 a, const d3 b)
{ return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dot(const d3 a, const d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Box = {x1,x2,y1,y2,z1,z2}, box.cuh:9
struct Box { double x1, x2, y1, y2, z1, z2; };

// box.cuh:13-22
__device__ __forceinline__ Box box_set(const d3 a, const d3 b, const d3 c)
{
    Box r;
    r.x1 = fmin3(a.x, b.x, c.x); r.x2 = fmax3(a.x, b.x, c.x);
    r.y1 = fmin3(a.y, b.y, c.y); r.y2 = fmax3(a.y, b.y, c.y);
    r.z1 = fmin3(a.z, b.z, c.z); r.z2 = fmax3(a.z, b.z, c.z);
    return r;
}
// box.cuh:24-32
__device__ __forceinline__ Box box_merge(const Box &a, const Box &b)
{
    Box r;
    r.x1 = fmin2(a.x1, b.x1); r.x2 = fmax2(a.x2, b.x2);
    r.y1 = fmin2(a.y1, b.y1); r.y2 = fmax2(a.y2, b.y2);
    r.z1 = fmin2(a.z1, b.z1); r.z2 = fmax2(a.z2, b.z2);
    return r;
}
// box.cuh:40-43 -- strict overlap, product form
__device__ __forceinline__ bool box_overlap(const Box &a, const Box &b)
{
    return (a.x1 - b.x2) * (b.x1 - a.x2) > 0 && (a.y1 - b.y2) * (b.y1 - a.y2) > 0 &&
           (a.z1 - b.z2) * (b.z1 - a.z2) > 0;
}

// triangle.cuh:18-30
__device__ __forceinline__ int neighbor_count(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t b0, uint32_t b1, uint32_t b2)
{
    return (a0 == b0) + (a0 == b1) + (a0 == b2) + (a1 == b0) + (a1 == b1) + (a1 == b2) +
           (a2 == b0) + (a2 == b1) + (a2 == b2);
}

// vec3f.cuh:257-270
__device__ __forceinline__ bool project3(const d3 ax, const d3 p1, const d3 p2, const d3 p3)
{
    const double P1 = dot(ax, p1), P2 = dot(ax, p2), P3 = dot(ax, p3);
    const double mx1 = fmax3(P1, P2, P3), mn1 = fmin3(P1, P2, P3);
    if (mn1 > 0) return false;
    if (0 > mx1) return false;
    return true;
}
// vec3f.cuh:272-291
__device__ __forceinline__ bool project6(const d3 ax, const d3 p1, const d3 p2, const d3 p3,
                                         const d3 q1, const d3 q2, const d3 q3)
{
    const double P1 = dot(ax, p1), P2 = dot(ax, p2), P3 = dot(ax, p3);
    const double Q1 = dot(ax, q1), Q2 = dot(ax, q2), Q3 = dot(ax, q3);
    const double mx1 = fmax3(P1, P2, P3), mn1 = fmin3(P1, P2, P3);
    const double mx2 = fmax3(Q1, Q2, Q3), mn2 = fmin3(Q1, Q2, Q3);
    if (mn1 > mx2) return false;
    if (mn2 > mx1) return false;
    return true;
}

// tri_contact.cuh:19-78: 17-axis SAT.  The verdict is a pure conjunction of the 17 interval tests,
// so evaluating axes lazily (cross product only when its test is reached) returns the same value
// as the reference's eager evaluation; each axis itself is computed with the reference's operations.
__device__ __forceinline__ bool tri_contact(const d3 P1, const d3 P2, const d3 P3, const d3 Q1, const d3 Q2, const d3 Q3)
{
    const d3 p1 = d3{0.0, 0.0, 0.0};
    const d3 p2 = sub(P2, P1), p3 = sub(P3, P1);
    const d3 q1 = sub(Q1, P1), q2 = sub(Q2, P1), q3 = sub(Q3, P1);
    const d3 e1 = sub(p2, p1), e2 = sub(p3, p2), e3 = sub(p1, p3);
    const d3 f1 = sub(q2, q1), f2 = sub(q3, q2), f3 = sub(q1, q3);
    const d3 n1 = cross(e1, e2);
    if (!project3(n1, q1, q2, q3)) return false;
    const d3 m1 = cross(f1, f2);
    if (!project3(m1, neg(q1), sub(p2, q1), sub(p3, q1))) return false;
    if (!project6(cross(e1, f1), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e1, f2), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e1, f3), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e2, f1), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e2, f2), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e2, f3), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e3, f1), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e3, f2), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e3, f3), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e1, n1), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e2, n1), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(e3, n1), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(f1, m1), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(f2, m1), p1, p2, p3, q1, q2, q3)) return false;
    if (!project6(cross(f3, m1), p1, p2, p3, q1, q2, q3)) return false;
    return true;
}

// The same verdict with the hardware's FP64 max / min (k_exact: the SAT of ~100 k survivors is 6.8 of its 14.6 us, and a third of an axis' ~65 instructions
// are the compare + two selects + wait states of each of its eight compare-selects).  v_max_f64 / v_min_f64 return what mathop.cuh's compare-selects return whenever
// neither operand is a NaN -- up to the SIGN of a zero (max(+0, -0)) and which of two EQUAL operands comes back, and the projections' results only ever meet a `>`,
// which sees neither.  With a NaN among an axis' dot products (non-finite vertices, or products that overflow to inf - inf) the two differ, on purpose in the
// reference (mathop.cuh:17-44 is NaN-asymmetric): `nan` says whether any evaluated axis had one, and tri_contact_fast then takes the verdict from tri_contact itself.
// (inline asm: fmax() would be canonicalised -- an extra v_max_f64 x, x per loaded operand under IEEE mode)
__device__ __forceinline__ double hw_max64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double hw_min64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ bool project3_hw(const d3 ax, const d3 p1, const d3 p2, const d3 p3, bool &nan)
{
    const double P1 = dot(ax, p1), P2 = dot(ax, p2), P3 = dot(ax, p3);
    nan |= __builtin_isunordered(P1, P2) | __builtin_isunordered(P3, P3);
    const double mx1 = hw_max64(hw_max64(P1, P2), P3), mn1 = hw_min64(hw_min64(P1, P2), P3);
    if (mn1 > 0) return false;
    if (0 > mx1) return false;
    return true;
}
__device__ __forceinline__ bool project6_hw(const d3 ax, const d3 p1, const d3 p2, const d3 p3, const d3 q1, const d3 q2, const d3 q3, bool &nan)
{
    const double P1 = dot(ax, p1), P2 = dot(ax, p2), P3 = dot(ax, p3);
    const double Q1 = dot(ax, q1), Q2 = dot(ax, q2), Q3 = dot(ax, q3);
    nan |= __builtin_isunordered(P1, P2) | __builtin_isunordered(P3, Q1) | __builtin_isunordered(Q2, Q3);
    const double mx1 = hw_max64(hw_max64(P1, P2), P3), mn1 = hw_min64(hw_min64(P1, P2), P3);
    const double mx2 = hw_max64(hw_max64(Q1, Q2), Q3), mn2 = hw_min64(hw_min64(Q1, Q2), Q3);
    if (mn1 > mx2) return false;
    if (mn2 > mx1) return false;
    return true;
}
__device__ __forceinline__ bool tri_contact_hw(const d3 P1, const d3 P2, const d3 P3, const d3 Q1, const d3 Q2, const d3 Q3, bool &nan)
{
    const d3 p1 = d3{0.0, 0.0, 0.0};
    const d3 p2 = sub(P2, P1), p3 = sub(P3, P1);
    const d3 q1 = sub(Q1, P1), q2 = sub(Q2, P1), q3 = sub(Q3, P1);
    const d3 e1 = sub(p2, p1), e2 = sub(p3, p2), e3 = sub(p1, p3);
    const d3 f1 = sub(q2, q1), f2 = sub(q3, q2), f3 = sub(q1, q3);
    const d3 n1 = cross(e1, e2);
    if (!project3_hw(n1, q1, q2, q3, nan)) return false;
    const d3 m1 = cross(f1, f2);
    if (!project3_hw(m1, neg(q1), sub(p2, q1), sub(p3, q1), nan)) return false;
    if (!project6_hw(cross(e1, f1), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e1, f2), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e1, f3), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e2, f1), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e2, f2), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e2, f3), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e3, f1), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e3, f2), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e3, f3), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e1, n1), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e2, n1), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(e3, n1), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(f1, m1), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(f2, m1), p1, p2, p3, q1, q2, q3, nan)) return false;
    if (!project6_hw(cross(f3, m1), p1, p2, p3, q1, q2, q3, nan)) return false;
    return true;
}
__device__ __forceinline__ bool tri_contact_fast(const d3 P1, const d3 P2, const d3 P3, const d3 Q1, const d3 Q2, const d3 Q3)
{
    bool nan = false;
    bool c = tri_contact_hw(P1, P2, P3, Q1, Q2, Q3, nan);
    if (nan) c = tri_contact(P1, P2, P3, Q1, Q2, Q3);                    // (a lane at a time, and rare: a NaN among the projections)
    return c;
}

__device__ __forceinline__ d3 load_vertex(const double *__restrict__ verts, uint32_t i)
{
    const double *p = verts + 3 * (size_t)i;
    return d3{p[0], p[1], p[2]};
}

// morton.h:7-29
__device__ __forceinline__ uint64_t expand64(uint64_t v)
{
    v &= 0x1fffffULL;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8)  & 0x100f00f00f00f00fULL;
    v = (v | v << 4)  & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2)  & 0x1249249249249249ULL;
    return v;
}
// double -> u64 of morton.h:80-82; negative / NaN -> 0, >= 2^63 -> 2^63-1 (undefined in the
// reference, defined here and identically in the oracle).
__device__ __forceinline__ uint64_t d2u64(double e)
{
    if (!(e > 0.0)) return 0;
    if (e >= 9223372036854775808.0) return 0x7fffffffffffffffULL;
    return (uint64_t)e;
}
// morton.h:70-89 with the frame as parameters
__device__ __forceinline__ uint64_t morton3d(double x, double y, double z, const double *off, const double *span)
{
    const double scale = 1048576.0;
    const double ex = ((x - off[0]) / span[0]) * scale;
    const double ey = ((y - off[1]) / span[1]) * scale;
    const double ez = ((z - off[2]) / span[2]) * scale;
    return (expand64(d2u64(ex)) << 2) | (expand64(d2u64(ey)) << 1) | expand64(d2u64(ez));
}

// ---------------------------------------------------------------- the ADAPTIVE frame (CD_FRAME_AUTO since round 6)
// Not reference behaviour: the reference has one frame, the constants of morton.h:43-58, and interleaves 20 bits an axis x, y, z
// (morton3d above: CD_FRAME_REFERENCE / CD_FRAME_CUSTOM, bit-identical to morton3D).  The pair set does not depend on the keys
// ("key freedom": a leaf is reached iff its own box overlaps the query's), the TREE does: per-axis normalisation of a
// 21 x 0.05 x 2.2 mesh (round 5's AUTO) gave cells of 400 : 1 and 44 node visits a query where this walks 29 (tools/sim/frame_study.py);
// an isotropic frame walks ~33 and leaves the thin axes' leading key bits constant -- the sort's 16 global bits (cd_sort.h) would hold 10
// that vary.  Here the 60 key bits are DEALT to the axes, all of them vary, and the cells of every level are as near to cubes as powers
// of two allow IN UNITS OF THE TRIANGLES' OWN EXTENT along each axis: a box query of size s meets a cell of length L with probability
// ~ (L + s); halving the cell along an axis costs (L + 2 s) / (L + s) -- least along the axis with the largest L / s, not the largest L.
// (A cloth is thin along one axis and so are its triangles: its sheets lie on top of each other there, and what separates them is worth a
// split early.  1 M cloth pair: 25.1 visits a query, the reference's hand-made frame 26.6, cubes by extent alone 30.9.)
//   statistic  per axis the mean of log2(box extent) over the triangles whose box is not flat on that axis, 8 fraction bits, piecewise
//              linear (flog2_fixed), summed as INTEGERS -- any order of summation gives the same sums, so this code and the oracle's
//              restatement (the CPU checker used by the tests) agree bit for bit;
//   E[a]       = flog2(extent of the centroids) + min(Lref - Lmean[a], LAYOUT_CAP): an axis whose triangles are thinner than those of the
//              axis where they are largest counts as longer by that ratio, at most 2^LAYOUT_CAP (every box flat on the axis: the cap);
//   layout     axes ordered by E, A >= B >= C (ties: the lower axis first); nA = round(E[A] - E[B]) leading bits split A alone,
//              nAB = round(E[B] - E[C]) pairs (A, B) follow, nABC triples (A, B, C) take the rest; one or two bits left over go to nA / nAB.
// layout word: bit 63 set | A | B << 2 | C << 4 | nA << 8 | nAB << 16 | nABC << 24;   0 = the reference's interleave.
constexpr unsigned long long LAYOUT_VALID = 1ull << 63;
constexpr int LAYOUT_CAP = 4;
constexpr double FLOG_MIN = 1e-300;                     // below this an extent counts as 0 (no subnormals in the statistic)
// 256 log2(x), piecewise linear between powers of two; x > 0 and normal
__device__ __forceinline__ long long flog2_fixed(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (((long long)((u >> 52) & 0x7ffull) - 1023) << 8) + (long long)((u >> 44) & 0xffull);
}
// (written on scalars, no arrays: with E[ord[j]] indexed at run time this function alone took > 100 VGPRs and put k_morton at 127 with spills)
__device__ __forceinline__ long long layout_mean(long long sum, long long cnt, long long none)
{
    // floor of the mean, by ONE IEEE division of two exactly represented integers (|sum| < 2^53): the same on every machine, and a tenth of the instructions of a 64-bit integer division
    return cnt > 0 ? (long long)floor((double)sum / (double)cnt) : none;
}
__device__ inline unsigned long long frame_layout(const double lo[3], const double hi[3], const long long sum[3], const long long cnt[3])
{
    const long long NONE = -(1ll << 40), CAPF = (long long)LAYOUT_CAP << 8;
    const long long L0 = layout_mean(sum[0], cnt[0], NONE), L1 = layout_mean(sum[1], cnt[1], NONE), L2 = layout_mean(sum[2], cnt[2], NONE);
    long long Lref = L0 > L1 ? L0 : L1; Lref = L2 > Lref ? L2 : Lref;
    auto weight = [&](double e, long long Lm) -> long long {
        if (!(e > FLOG_MIN)) return NONE;
        long long d = CAPF;
        if (Lm != NONE) { d = Lref - Lm; if (d > CAPF) d = CAPF; }
        if (Lref == NONE) d = 0;                                              // every box flat on every axis: points
        return flog2_fixed(e) + d;
    };
    long long EA = weight(hi[0] - lo[0], L0), EB = weight(hi[1] - lo[1], L1), EC = weight(hi[2] - lo[2], L2);
    int A = 0, B = 1, C = 2;
    // stable, descending (an insertion sort of three: swap only on a strict '>', so ties keep the lower axis first)
    if (EB > EA) { const long long t = EA; EA = EB; EB = t; const int u = A; A = B; B = u; }
    if (EC > EB) { const long long t = EB; EB = EC; EC = t; const int u = B; B = C; C = u; }
    if (EB > EA) { const long long t = EA; EA = EB; EB = t; const int u = A; A = B; B = u; }
    long long nA = EA == NONE ? 0 : (EB == NONE ? 60 : (EA - EB + 128) >> 8);
    if (nA > 60) nA = 60;
    long long rem = 60 - nA;
    long long nAB = EB == NONE ? 0 : (EC == NONE ? 30 : (EB - EC + 128) >> 8);
    if (2 * nAB > rem) nAB = rem / 2;
    rem -= 2 * nAB;
    const long long nABC = rem / 3, left = rem % 3;
    if (left == 1) ++nA;
    if (left == 2) ++nAB;
    return LAYOUT_VALID | (unsigned long long)A | ((unsigned long long)B << 2) | ((unsigned long long)C << 4) |
           ((unsigned long long)nA << 8) | ((unsigned long long)nAB << 16) | ((unsigned long long)nABC << 24);
}
// is `w` a layout word morton3d_layout can take?  (0: the reference's interleave)
__host__ __device__ inline bool layout_ok(unsigned long long w)
{
    if (w == 0ull) return true;
    if (!(w >> 63) || ((w >> 32) & 0x7fffffffull) || ((w >> 6) & 3ull)) return false;
    const int A = (int)(w & 3), B = (int)((w >> 2) & 3), C = (int)((w >> 4) & 3), nA = (int)((w >> 8) & 255), p = (int)((w >> 16) & 255), t = (int)((w >> 24) & 255);
    return A < 3 && B < 3 && C < 3 && A != B && A != C && B != C && t <= 20 && p <= 30 && nA + 2 * p + 3 * t <= 60;
}
// spread the low 32 bits to the even positions
__device__ __forceinline__ uint64_t expand2(uint64_t v)
{
    v &= 0xffffffffULL;
    v = (v | v << 16) & 0x0000ffff0000ffffULL;
    v = (v | v << 8)  & 0x00ff00ff00ff00ffULL;
    v = (v | v << 4)  & 0x0f0f0f0f0f0f0f0fULL;
    v = (v | v << 2)  & 0x3333333333333333ULL;
    v = (v | v << 1)  & 0x5555555555555555ULL;
    return v;
}
// A layout word decoded once (wave-uniform: scalar registers) for a loop over keys.
// The cell of a triangle along axis a, in a frame WITH a layout, is   floor(((p1 + p2 + p3) - 3 off) * (2^bits / (3 span)))   clamped to [0, 2^bits - 1]:
// the vertex SUM against thrice the offset, times one factor per axis formed once per frame -- no division per key.  (morton.h's ((c - off) / span) * 2^20 on the
// centroid c = (p1 + p2 + p3) / 3 is six FP64 divisions a key, ~90 of the ~150 vector instructions k_morton<false> spends on one: they stay where the reference's
// bits are the contract, CD_FRAME_REFERENCE / CD_FRAME_CUSTOM.  Here the contract is this library's own -- the oracle restates it -- and the clamp makes it safe
// against the last cell's end whatever the rounding.)
struct KeyLayout {
    int A, B, C, t, p, nA;                              // t triples, p pairs, nA leading bits
    double off3A, off3B, off3C, kA, kB, kC;             // 3 off, 2^bits / (3 span)
    uint64_t topA, topB, topC;
};
// (a wave-uniform double into scalar registers: the compiler cannot see that a value read from LDS or through a pointer is uniform)
__device__ __forceinline__ double uniform_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ KeyLayout key_layout(unsigned long long w, const double *off, const double *span)
{
    KeyLayout k;
    k.A = (int)(w & 3); k.B = (int)((w >> 2) & 3); k.C = (int)((w >> 4) & 3);
    k.nA = (int)((w >> 8) & 255); k.p = (int)((w >> 16) & 255); k.t = (int)((w >> 24) & 255);
    const int bA = k.nA + k.p + k.t, bB = k.p + k.t, bC = k.t;
    auto two_to = [](int b) { return __longlong_as_double((long long)(1023 + b) << 52); };                // 2^b, exact
    k.off3A = uniform_f64(3.0 * off[k.A]); k.off3B = uniform_f64(3.0 * off[k.B]); k.off3C = uniform_f64(3.0 * off[k.C]);
    k.kA = uniform_f64(two_to(bA) / (3.0 * span[k.A])); k.kB = uniform_f64(two_to(bB) / (3.0 * span[k.B])); k.kC = uniform_f64(two_to(bC) / (3.0 * span[k.C]));
    k.topA = (1ull << bA) - 1; k.topB = (1ull << bB) - 1; k.topC = (1ull << bC) - 1;
    return k;
}
// sx, sy, sz: the SUM of the triangle's three vertices per axis, p1 + p2 + p3 in that order (load_obj.h:90's numerator)
__device__ __forceinline__ uint64_t morton3d_layout(double sx, double sy, double sz, const KeyLayout &k)
{
    const double cA = k.A == 0 ? sx : (k.A == 1 ? sy : sz), cB = k.B == 0 ? sx : (k.B == 1 ? sy : sz), cC = k.C == 0 ? sx : (k.C == 1 ? sy : sz);
    uint64_t ia = d2u64((cA - k.off3A) * k.kA), ib = d2u64((cB - k.off3B) * k.kB), ic = d2u64((cC - k.off3C) * k.kC);
    ia = ia > k.topA ? k.topA : ia; ib = ib > k.topB ? k.topB : ib; ic = ic > k.topC ? k.topC : ic;   // a centroid beyond the frame takes the last cell: the key stays below 2^60
    const uint64_t mt = (1ull << k.t) - 1, mp = (1ull << k.p) - 1;
    const uint64_t triples = (expand64(ia & mt) << 2) | (expand64(ib & mt) << 1) | expand64(ic & mt);
    const uint64_t pairs = (expand2((ia >> k.t) & mp) << 1) | expand2((ib >> k.t) & mp);
    return ((ia >> (k.p + k.t)) << (2 * k.p + 3 * k.t)) | (pairs << (3 * k.t)) | triples;
}

}  // namespace cd
