"""Writes tests/golden/contact_ref.npz: vectors computed by the REFERENCE's own exact-test code --
/root/reference/CollisionDetection/tri_contact.cuh:19-87, box.cuh:13-43, triangle.cuh:18-30, vec3f.cuh:118-125,257-291,
mathop.cuh:17-44 -- compiled unmodified into oracle/_ref/libref_contact.so (recipe: oracle/Makefile, wrapper TU
oracle/ref_contact.cpp).  Runs in the build container only (the reference does not exist on the GPU box), from the
repo root:   make -C oracle && python tests/golden/make_contact_ref.py         (about two minutes, one core)

What the fixture holds (inputs come from tests/contact_inputs.py, reproducible on any box; their SHA-256 is stored):
  tri_contact_bits / _sha      checkTriangleContact of all 1 179 648 triangle pairs (seven families), bit-packed, in full
  helper_bits, neighbor        checkTriangleContactHelper and Triangle::neighborCount of 2^18 indexed pairs
  box_overlap_bits             checkBoxOverlap of 2^20 box pairs (touching faces, zero-thickness, identical, random)
  box_set_sha / _head          Box::set of the 2^18 + 2^18 indexed triangles (bit patterns); box_merge_* likewise
  project3_bits, project6_bits, cross, dot     the vec3f helpers on 2^16 operand sets each
  <cfg>_pairs_sha / _count / _sample / _tested [/ _pairs]   for cfg in soup100k (BASELINE config 2, plain O(N^2)),
       cloth1M (config 3), soup1M, cloth1M_double: the END RESULT as ref_pair_set() computes it from the reference's
       predicates -- SHA-256 of the sorted (a << 32 | b) keys, their count, every 64th key, the pairs-tested count, and
       the keys in full when there are at most 65 536 of them; plus the SHA-256 of the mesh itself.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import contact_inputs as ci  # noqa: E402


def ref_lib():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_contact.so"))
    vp, sz = C.c_void_p, C.c_size_t
    L.ref_sizes.argtypes = [vp]
    L.ref_tri_contact.argtypes = [vp, sz, vp]
    L.ref_contact_helper.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    L.ref_neighbor_count.argtypes = [vp, vp, sz, vp]
    L.ref_box_set.argtypes = [vp, vp, sz, vp]
    L.ref_box_merge.argtypes = [vp, vp, sz, vp]
    L.ref_box_overlap.argtypes = [vp, vp, sz, vp]
    L.ref_project3.argtypes = [vp, sz, vp]
    L.ref_project6.argtypes = [vp, sz, vp]
    L.ref_cross_dot.argtypes = [vp, sz, vp, vp]
    L.ref_pair_set.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, C.c_uint64, vp]; L.ref_pair_set.restype = C.c_uint64
    return L


def p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def bits(a):
    assert a.min() >= 0 and a.max() <= 1
    return np.packbits(a.astype(np.uint8))


def main():
    L = ref_lib()
    out = {}
    s = np.zeros(3, dtype=np.uint32); L.ref_sizes(p(s))
    assert s.tolist() == [24, 56, 56], s
    out["sizes"] = s

    tri, fam = ci.tri_pairs()
    r = np.zeros(tri.shape[0], dtype=np.int32)
    L.ref_tri_contact(p(tri), tri.shape[0], p(r))
    out["tri_in_sha"] = ci.sha(tri); out["tri_contact_bits"] = bits(r); out["tri_contact_sha"] = ci.sha(r.astype(np.uint8))
    out["tri_contact_per_family"] = np.array([int(r[fam == f].sum()) for f in range(len(ci.FAMILIES))], dtype=np.int64)
    print("tri_contact", tri.shape[0], "pairs; contacts per family", dict(zip(ci.FAMILIES, out["tri_contact_per_family"].tolist())))

    verts, va, ida, vb, idb = ci.indexed_pairs()
    n = va.shape[0]
    h = np.zeros(n, dtype=np.int32); nc = np.zeros(n, dtype=np.int32)
    L.ref_contact_helper(p(verts), p(va), p(ida), p(vb), p(idb), n, p(h))
    L.ref_neighbor_count(p(va), p(vb), n, p(nc))
    out["indexed_in_sha"] = ci.sha(np.concatenate([verts.view(np.uint8).ravel(), va.view(np.uint8).ravel(), ida.view(np.uint8), vb.view(np.uint8).ravel(), idb.view(np.uint8)]))
    out["helper_bits"] = bits(h); out["neighbor"] = nc.astype(np.uint8)
    print("helper", int(h.sum()), "of", n, "| neighborCount histogram", np.bincount(nc, minlength=10).tolist())
    both = np.ascontiguousarray(np.concatenate([va, vb]))
    bs = np.zeros((both.shape[0], 6), dtype=np.float64)
    L.ref_box_set(p(verts), p(both), both.shape[0], p(bs))
    out["box_set_sha"] = ci.sha(bs); out["box_set_head"] = bs[:4096].copy()
    bm = np.zeros((n, 6), dtype=np.float64)
    L.ref_box_merge(p(np.ascontiguousarray(bs[:n])), p(np.ascontiguousarray(bs[n:])), n, p(bm))
    out["box_merge_sha"] = ci.sha(bm); out["box_merge_head"] = bm[:4096].copy()

    a, b = ci.box_pairs()
    o = np.zeros(a.shape[0], dtype=np.int32)
    L.ref_box_overlap(p(a), p(b), a.shape[0], p(o))
    out["box_in_sha"] = ci.sha(np.concatenate([a, b])); out["box_overlap_bits"] = bits(o)
    print("box overlap", int(o.sum()), "of", a.shape[0])

    v4, v7, v2 = ci.small_vectors()
    r3 = np.zeros(v4.shape[0], dtype=np.int32); r6 = np.zeros(v7.shape[0], dtype=np.int32)
    L.ref_project3(p(v4), v4.shape[0], p(r3)); L.ref_project6(p(v7), v7.shape[0], p(r6))
    cr = np.zeros((v2.shape[0], 3), dtype=np.float64); dt = np.zeros(v2.shape[0], dtype=np.float64)
    L.ref_cross_dot(p(v2), v2.shape[0], p(cr), p(dt))
    out["small_in_sha"] = ci.sha(np.concatenate([v4.ravel(), v7.ravel(), v2.ravel()]))
    out["project3_bits"] = bits(r3); out["project6_bits"] = bits(r6); out["cross"] = cr; out["dot"] = dt

    for name, (mv, mi_), mode in ci.end_configs():
        t0 = time.time()
        cap = 1 << 22
        pairs = np.zeros((cap, 2), dtype=np.uint32); tested = np.zeros(1, dtype=np.uint64)
        cnt = L.ref_pair_set(p(mv), p(mi_), None, mi_.shape[0], mode, p(pairs), cap, p(tested))
        assert cnt <= cap
        keys = ci.pair_keys(pairs[:cnt])
        assert np.unique(keys).size == keys.size
        out[name + "_verts_sha"] = ci.sha(mv); out[name + "_vidx_sha"] = ci.sha(mi_)
        out[name + "_pairs_sha"] = ci.sha(keys); out[name + "_count"] = np.uint64(cnt); out[name + "_tested"] = tested[0]
        out[name + "_sample"] = keys[::ci.SAMPLE_STRIDE].copy()
        if cnt <= 65536:
            out[name + "_pairs"] = keys
        print("%s: %d triangles, mode %d -> %d pairs, %d pairs tested (%.1f s)" % (name, mi_.shape[0], mode, cnt, int(tested[0]), time.time() - t0), flush=True)

    path = os.path.join(HERE, "contact_ref.npz")
    np.savez_compressed(path, **out)
    print(os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
