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

}  // namespace cd
