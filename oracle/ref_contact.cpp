// Test infrastructure, build container only: exposes the REFERENCE's own exact-test predicates -- the unmodified
// /root/reference/CollisionDetection/{tri_contact,box,triangle,vec3f,mathop}.cuh, found through -I$(REFDIR) -- behind a C ABI
// so that tests/golden/make_contact_ref.py can write reference-compiled vectors (tests/golden/contact_ref.npz).
//
// No stand-in header is written for this: <cuda_runtime.h> is the genuine NVIDIA header (with crt/host_defines.h, which gives
// __device__ / __host__ their host-compiler meaning) that this image ships inside the triton wheel; oracle/Makefile DISCOVERS
// that directory at build time and skips this target where it is absent.  Nothing of it is copied into the repo.
// bvh.cuh / collision.cuh / check.cuh need __clzll, atomicAdd, threadIdx (device intrinsics -> would need stand-ins) and are NOT
// built; what ref_pair_set() below adds around the reference's predicates is only their SEQUENCE (collision.cuh:31-44), and the
// candidate proposal, which cannot change the result (every proposed pair is decided by the reference's checkBoxOverlap).
//
// Neither this library nor the reference travels to the GPU box; only the .npz does.  Built into oracle/_ref/ (git-ignored).
#include <cuda_runtime.h>
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <vector>
#include <algorithm>
#include <numeric>
#include "tri_contact.cuh"   // checkTriangleContact :19-78, checkTriangleContactHelper :80-87  (pulls triangle.cuh, vec3f.cuh, mathop.cuh)
#include "box.cuh"           // Box::set :13-22, Box::merge :24-32, checkBoxOverlap :40-43

static_assert(sizeof(vec3f) == 24, "vec3f.cuh:14-23 layout");
static_assert(sizeof(Triangle) == 56, "triangle.cuh:5-9 layout");
static_assert(sizeof(Box) == 56, "box.cuh:8-11 layout");

namespace {
inline vec3f* as_vec(const double* p) { return reinterpret_cast<vec3f*>(const_cast<double*>(p)); }   // vec3f IS double[3] (asserted)
inline Triangle make_tri(uint32_t id, const uint32_t* v)
{
    Triangle t; std::memset(&t, 0, sizeof t);
    t.ID = id; t.vIdx[0] = v[0]; t.vIdx[1] = v[1]; t.vIdx[2] = v[2];
    return t;
}
inline void box_out(const Box& b, double* o) { o[0] = b.x1; o[1] = b.x2; o[2] = b.y1; o[3] = b.y2; o[4] = b.z1; o[5] = b.z2; }
inline Box box_in(const double* o)
{
    Box b; std::memset(&b, 0, sizeof b);
    b.x1 = o[0]; b.x2 = o[1]; b.y1 = o[2]; b.y2 = o[3]; b.z1 = o[4]; b.z2 = o[5]; b.init = 1;
    return b;
}
}

extern "C" {

void ref_sizes(uint32_t out[3]) { out[0] = sizeof(vec3f); out[1] = sizeof(Triangle); out[2] = sizeof(Box); }

// tri_contact.cuh:19-78 on explicit vertex positions: tri[k] = 6 points x 3 doubles (P1 P2 P3 Q1 Q2 Q3)
void ref_tri_contact(const double* tri, size_t n, int32_t* out)
{
    for (size_t k = 0; k < n; ++k) {
        const double* t = tri + 18 * k;
        vec3f P1(*as_vec(t)), P2(*as_vec(t + 3)), P3(*as_vec(t + 6)), Q1(*as_vec(t + 9)), Q2(*as_vec(t + 12)), Q3(*as_vec(t + 15));
        out[k] = checkTriangleContact(P1, P2, P3, Q1, Q2, Q3);
    }
}

// tri_contact.cuh:80-87 (ID rule + vertex fetch) and triangle.cuh:18-30 on indexed triangles of one vertex array
void ref_contact_helper(const double* verts, const uint32_t* va, const uint32_t* ida, const uint32_t* vb, const uint32_t* idb,
                        size_t n, int32_t* out)
{
    for (size_t k = 0; k < n; ++k) {
        Triangle a = make_tri(ida[k], va + 3 * k), b = make_tri(idb[k], vb + 3 * k);
        out[k] = checkTriangleContactHelper(&a, &b, as_vec(verts));
    }
}
void ref_neighbor_count(const uint32_t* va, const uint32_t* vb, size_t n, int32_t* out)
{
    for (size_t k = 0; k < n; ++k) {
        Triangle a = make_tri(0, va + 3 * k), b = make_tri(1, vb + 3 * k);
        out[k] = a.neighborCount(&b);
    }
}

// box.cuh:13-22, :24-32, :40-43; boxes as {x1,x2,y1,y2,z1,z2}
void ref_box_set(const double* verts, const uint32_t* vidx, size_t n, double* out)
{
    for (size_t k = 0; k < n; ++k) {
        Triangle t = make_tri((uint32_t)k, vidx + 3 * k);
        Box b; std::memset(&b, 0, sizeof b);
        b.set(&t, as_vec(verts));
        box_out(b, out + 6 * k);
    }
}
void ref_box_merge(const double* a, const double* b, size_t n, double* out)
{
    for (size_t k = 0; k < n; ++k) {
        Box x = box_in(a + 6 * k), y = box_in(b + 6 * k), m; std::memset(&m, 0, sizeof m);
        m.merge(&x, &y);
        box_out(m, out + 6 * k);
    }
}
void ref_box_overlap(const double* a, const double* b, size_t n, int32_t* out)
{
    for (size_t k = 0; k < n; ++k) { Box x = box_in(a + 6 * k), y = box_in(b + 6 * k); out[k] = checkBoxOverlap(&x, &y); }
}

// vec3f.cuh:257-271 / :273-291 ; ax + 3 (resp. 6) points per item
void ref_project3(const double* v, size_t n, int32_t* out)
{
    for (size_t k = 0; k < n; ++k) { const double* t = v + 12 * k; out[k] = project3(*as_vec(t), *as_vec(t + 3), *as_vec(t + 6), *as_vec(t + 9)); }
}
void ref_project6(const double* v, size_t n, int32_t* out)
{
    for (size_t k = 0; k < n; ++k) {
        const double* t = v + 21 * k;
        out[k] = project6(*as_vec(t), *as_vec(t + 3), *as_vec(t + 6), *as_vec(t + 9), *as_vec(t + 12), *as_vec(t + 15), *as_vec(t + 18));
    }
}
// vec3f.cuh:118-121 cross, :123-125 dot ; two vectors per item
void ref_cross_dot(const double* v, size_t n, double* cross_out, double* dot_out)
{
    for (size_t k = 0; k < n; ++k) {
        const vec3f& a = *as_vec(v + 6 * k); const vec3f& b = *as_vec(v + 6 * k + 3);
        vec3f c = a.cross(b);
        cross_out[3 * k] = c.x; cross_out[3 * k + 1] = c.y; cross_out[3 * k + 2] = c.z;
        dot_out[k] = a.dot(b);
    }
}

// The END RESULT of findCollisions (collision.cuh:19-88) without its tree: (q, l) is reported iff the leaf boxes strictly overlap
// (checkBoxOverlap, :31-32), neighborCount < 1 (:36 / :52) and checkTriangleContactHelper > 0 (:37 / :53, which holds the ID rule);
// every (q, l) with overlapping boxes -- q == l included, both orders -- is one "pair tested" (SURVEY.md 8(d)).  A leaf is reached by
// the traversal iff its box overlaps the query's: an ancestor's box contains the leaf's (Box::merge) and strict interval overlap
// with a sub-interval implies it with the super-interval, so the tree cannot change this set.
//   mode 0: plain O(N^2), every ordered (q, l) goes to the reference's checkBoxOverlap;
//   mode 1: candidates proposed by this file's sweep along x (a superset: closed-interval x overlap on the reference's own
//           Box::set values), each unordered candidate decided by the reference's checkBoxOverlap in both orders, self pairs too.
// Output: pairs as (q ID, l ID) rows in discovery order (caller sorts), up to cap; returns the full count; *tested = pairs tested.
uint64_t ref_pair_set(const double* verts, const uint32_t* vidx, const uint32_t* ids, uint32_t n, int mode,
                      uint32_t* pairs, uint64_t cap, uint64_t* tested)
{
    std::vector<Triangle> tri(n);
    std::vector<Box> box(n);
    vec3f* vs = as_vec(verts);
    for (uint32_t i = 0; i < n; ++i) {
        tri[i] = make_tri(ids ? ids[i] : i, vidx + 3 * (size_t)i);
        std::memset(&box[i], 0, sizeof(Box));
        box[i].set(&tri[i], vs);
    }
    uint64_t np = 0, nt = 0;
    auto leaf = [&](uint32_t q, uint32_t l) {          // collision.cuh:34-44 once the boxes overlap
        ++nt;
        if (tri[q].neighborCount(&tri[l]) < 1 && checkTriangleContactHelper(&tri[q], &tri[l], vs) > 0) {
            if (np < cap) { pairs[2 * np] = tri[q].ID; pairs[2 * np + 1] = tri[l].ID; }
            ++np;
        }
    };
    if (mode == 0) {
        for (uint32_t q = 0; q < n; ++q)
            for (uint32_t l = 0; l < n; ++l)
                if (checkBoxOverlap(&box[q], &box[l]) > 0) leaf(q, l);
    } else {
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return box[a].x1 < box[b].x1 || (box[a].x1 == box[b].x1 && a < b); });
        for (uint32_t s = 0; s < n; ++s) {
            uint32_t i = order[s];
            if (checkBoxOverlap(&box[i], &box[i]) > 0) leaf(i, i);
            for (uint32_t u = s + 1; u < n && box[order[u]].x1 <= box[i].x2; ++u) {
                uint32_t j = order[u];
                if (checkBoxOverlap(&box[i], &box[j]) > 0) leaf(i, j);
                if (checkBoxOverlap(&box[j], &box[i]) > 0) leaf(j, i);
            }
        }
    }
    *tested = nt;
    return np;
}

}
