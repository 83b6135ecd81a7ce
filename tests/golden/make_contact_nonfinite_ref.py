"""Writes tests/golden/contact_nonfinite_ref.npz: checkTriangleContact (tri_contact.cuh:19-87 with vec3f.cuh / mathop.cuh, compiled unmodified into
oracle/_ref/libref_contact.so: oracle/Makefile, oracle/ref_contact.cpp) on tests/contact_inputs.nonfinite_pairs() -- 65 536 triangle pairs with NaN, +-inf, +-0, huge and tiny
coordinates.  Runs in the build container only, from the repo root:   make -C oracle && python tests/golden/make_contact_nonfinite_ref.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import contact_inputs as ci  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_contact.so"))
L.ref_tri_contact.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
tri = ci.nonfinite_pairs()
r = np.zeros(tri.shape[0], dtype=np.int32)
L.ref_tri_contact(tri.ctypes.data_as(C.c_void_p), tri.shape[0], r.ctypes.data_as(C.c_void_p))
assert r.min() >= 0 and r.max() <= 1
nan_pairs = int(np.isnan(tri).any(axis=(1, 2)).sum())
np.savez_compressed(os.path.join(HERE, "contact_nonfinite_ref.npz"), tri_in_sha=ci.sha(tri), tri_contact_bits=np.packbits(r.astype(np.uint8)),
                    contacts=np.int64(r.sum()), pairs_with_a_nan=np.int64(nan_pairs))
print("pairs", tri.shape[0], "contacts", int(r.sum()), "pairs with a NaN coordinate", nan_pairs, "non-finite", int((~np.isfinite(tri)).any(axis=(1, 2)).sum()))
