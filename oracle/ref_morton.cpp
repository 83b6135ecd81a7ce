// Test infrastructure, build container only: exposes the REFERENCE's own Morton code -- the unmodified
// /root/reference/CollisionDetection/morton.h, found through -I, no stand-in headers (it needs only <assert.h>) --
// behind a C ABI so that tests/golden/make_morton_ref.py can write reference-compiled vectors
// (tests/golden/morton_ref.npz).  Neither this library nor the reference travels to the GPU box; only the .npz does.
// Built by oracle/Makefile into oracle/_ref/ (git-ignored).  assert() stays enabled: a point outside morton3D's
// domain (morton.h:78) aborts the generator instead of producing an undefined key.
#include <cstdint>
#include <cstddef>
#include "morton.h"   // expand64Bits :7-29, normX/Y/Z :43-58, morton3D :70-89

extern "C" {
void ref_expand64Bits(const uint64_t* v, size_t n, uint64_t* out) { for (size_t i = 0; i < n; ++i) out[i] = expand64Bits(v[i]); }
void ref_morton3D(const double* xyz, size_t n, uint64_t* out) { for (size_t i = 0; i < n; ++i) out[i] = morton3D(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]); }
void ref_norm(const double* xyz, size_t n, double* out)
{
    for (size_t i = 0; i < n; ++i) { out[3 * i] = normX(xyz[3 * i]); out[3 * i + 1] = normY(xyz[3 * i + 1]); out[3 * i + 2] = normZ(xyz[3 * i + 2]); }
}
}
