"""Writes tests/golden/morton_ref.npz: vectors computed by the REFERENCE's own morton.h
(/root/reference/CollisionDetection/morton.h:7-29,43-58,70-89), compiled unmodified into oracle/_ref/libref_morton.so
(recipe: oracle/Makefile, wrapper TU oracle/ref_morton.cpp).  Runs in the build container only -- the reference does
not exist on the GPU box -- from the repo root:  make -C oracle && python tests/golden/make_morton_ref.py

What the fixture holds (inputs come from tests/morton_inputs.py, reproducible on any box; their SHA-256 is stored):
  expand_out                 expand64Bits of all N_EXPAND inputs, in full
  points_keys_head           morton3D of the first N_FULL frame points, in full
  points_keys_sha / _sample  SHA-256 of all N_POINTS keys (little-endian u64) and every 64th key
  cloth_keys_sha / _sample   the same for the 1 000 000 centroids of BASELINE config 3, in triangle order
  cloth_sorted_sha / _sample the same keys sorted ascending (what cd_export_keys returns for config 3)
  norm_head                  normX/Y/Z of the first 4096 points (the FP64 division alone)
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import morton_inputs as mi  # noqa: E402


def ref_lib():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_morton.so"))
    vp = C.c_void_p
    for f in (L.ref_expand64Bits, L.ref_morton3D, L.ref_norm):
        f.argtypes = [vp, C.c_size_t, vp]; f.restype = None
    return L


def main():
    L = ref_lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    out = {}
    v = mi.expand_inputs()
    e = np.zeros_like(v)
    L.ref_expand64Bits(p(v), v.size, p(e))
    out["expand_in_sha"] = mi.sha(v); out["expand_out"] = e

    pts = mi.frame_points()
    k = np.zeros(pts.shape[0], dtype=np.uint64)
    L.ref_morton3D(p(pts), pts.shape[0], p(k))
    nrm = np.zeros((4096, 3), dtype=np.float64)
    L.ref_norm(p(pts), 4096, p(nrm))
    out["points_in_sha"] = mi.sha(pts); out["points_keys_head"] = k[:mi.N_FULL].copy()
    out["points_keys_sha"] = mi.sha(k); out["points_keys_sample"] = k[::mi.SAMPLE_STRIDE].copy()
    out["norm_head"] = nrm

    cen, verts, vidx = mi.cloth_centroids(500)
    ck = np.zeros(cen.shape[0], dtype=np.uint64)
    L.ref_morton3D(p(cen), cen.shape[0], p(ck))
    cs = np.sort(ck, kind="stable")
    out["cloth_verts_sha"] = mi.sha(verts); out["cloth_vidx_sha"] = mi.sha(vidx); out["cloth_centroids_sha"] = mi.sha(cen)
    out["cloth_keys_sha"] = mi.sha(ck); out["cloth_keys_sample"] = ck[::mi.SAMPLE_STRIDE].copy()
    out["cloth_sorted_sha"] = mi.sha(cs); out["cloth_sorted_sample"] = cs[::mi.SAMPLE_STRIDE].copy()
    out["cloth_first_last"] = np.array([cs[0], cs[-1]], dtype=np.uint64)
    out["cloth_distinct"] = np.uint64(np.unique(ck).size)

    # the two anchors SURVEY.md recorded, now from the reference itself
    a = np.array([[1.0, 0.0, 0.5], [0.004501 + 3.08 / 2, -0.476622 + 0.76 / 2, -0.381965 + 2.36 / 2]])
    ak = np.zeros(2, dtype=np.uint64)
    L.ref_morton3D(p(a), 2, p(ak))
    out["anchor_points"] = a; out["anchor_keys"] = ak
    path = os.path.join(HERE, "morton_ref.npz")
    np.savez_compressed(path, **out)
    print("anchors", ak.tolist(), "| expand", v.size, "| points", k.size, "| cloth keys", ck.size, "distinct", int(out["cloth_distinct"]),
          "|", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
