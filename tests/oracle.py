"""ctypes binding of oracle/liboracle.so -- the CPU checker.  Test infrastructure only:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class OrcStats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("pairs_tested", C.c_uint64), ("node_visits", C.c_uint64),
                ("max_stack", C.c_uint32), ("overflow", C.c_uint32)]


class OrcTimes(C.Structure):
    _fields_ = [("ms_morton", C.c_double), ("ms_sort", C.c_double), ("ms_hierarchy", C.c_double),
                ("ms_refit", C.c_double), ("ms_traverse", C.c_double)]


SPHERE_DTYPE = np.dtype([("r", "<f4"), ("b", "<f4"), ("g", "<f4"), ("radius", "<f4"),
                         ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("idx", "<i4")])

_libs = {}


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib(omp: bool = False) -> C.CDLL:
    name = "liboracle_omp.so" if omp else "liboracle.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp = C.c_void_p
    L.orc_expand64.argtypes = [C.c_uint64]; L.orc_expand64.restype = C.c_uint64
    L.orc_morton3d.argtypes = [C.c_double, C.c_double, C.c_double, vp, vp]; L.orc_morton3d.restype = C.c_uint64
    L.orc_morton3d_batch.argtypes = [vp, C.c_uint64, vp, vp, vp]; L.orc_morton3d_batch.restype = None
    L.orc_expand64_batch.argtypes = [vp, C.c_uint64, vp]; L.orc_expand64_batch.restype = None
    L.orc_centroid_morton.argtypes = [vp, vp, C.c_uint32, vp, vp, vp, vp]; L.orc_centroid_morton.restype = None
    L.orc_sort_by_key.argtypes = [vp, vp, C.c_uint32]; L.orc_sort_by_key.restype = None
    L.orc_frame_layout.argtypes = [vp, vp, vp, vp]; L.orc_frame_layout.restype = C.c_uint64
    L.orc_frame_layout_cap.argtypes = [vp, vp, vp, vp, C.c_int]; L.orc_frame_layout_cap.restype = C.c_uint64
    L.orc_layout_stat.argtypes = [vp, vp, C.c_uint32, vp, vp, vp, vp]; L.orc_layout_stat.restype = None
    L.orc_morton3d_layout.argtypes = [C.c_double, C.c_double, C.c_double, vp, vp, C.c_uint64]; L.orc_morton3d_layout.restype = C.c_uint64
    L.orc_auto_frame.argtypes = [vp, vp, C.c_uint32, vp]; L.orc_auto_frame.restype = C.c_uint64
    L.orc_centroid_morton_layout.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_uint64, vp]; L.orc_centroid_morton_layout.restype = None
    L.orc_morton3d_layout_batch.argtypes = [vp, C.c_uint64, vp, vp, C.c_uint64, vp]; L.orc_morton3d_layout_batch.restype = None
    L.orc_clz64.argtypes = [C.c_uint64]; L.orc_clz64.restype = C.c_int
    L.orc_delta.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]; L.orc_delta.restype = C.c_int
    L.orc_find_split.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]; L.orc_find_split.restype = C.c_int
    L.orc_determine_range.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_determine_range.restype = None
    L.orc_build_hierarchy.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_uint32)]
    L.orc_build_hierarchy.restype = None
    L.orc_box_overlap.argtypes = [vp, vp]; L.orc_box_overlap.restype = C.c_int
    L.orc_refit.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]; L.orc_refit.restype = None
    L.orc_neighbor_count.argtypes = [vp, vp]; L.orc_neighbor_count.restype = C.c_int
    L.orc_tri_contact.argtypes = [vp] * 6; L.orc_tri_contact.restype = C.c_int
    L.orc_tri_contact_batch.argtypes = [vp, vp, vp, vp, C.c_uint64, vp]; L.orc_tri_contact_batch.restype = None
    for f, k in ((L.orc_tri_contact_points_batch, 3), (L.orc_helper_batch, 7), (L.orc_neighbor_count_batch, 4), (L.orc_box_set_batch, 4),
                 (L.orc_box_merge_batch, 4), (L.orc_box_overlap_batch, 4), (L.orc_project3_batch, 3), (L.orc_project6_batch, 3),
                 (L.orc_cross_dot_batch, 4)):
        f.argtypes = [vp] * k; f.restype = None
    for f, pos in ((L.orc_tri_contact_points_batch, 1), (L.orc_helper_batch, 5), (L.orc_neighbor_count_batch, 2), (L.orc_box_set_batch, 2),
                   (L.orc_box_merge_batch, 2), (L.orc_box_overlap_batch, 2), (L.orc_project3_batch, 1), (L.orc_project6_batch, 1),
                   (L.orc_cross_dot_batch, 1)):
        a = list(f.argtypes); a[pos] = C.c_uint64; f.argtypes = a
    L.orc_find_collisions.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_uint64, C.POINTER(OrcStats)]
    L.orc_find_collisions.restype = None
    L.orc_find_collisions_queries.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_uint32, vp, C.c_uint64, C.POINTER(OrcStats)]
    L.orc_find_collisions_queries.restype = None
    L.orc_brute_force.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.orc_brute_force.restype = C.c_uint64
    L.orc_check_internal.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp]; L.orc_check_internal.restype = None
    L.orc_check_leaves.argtypes = [C.c_int, vp, vp, vp, C.c_uint32, vp, vp]; L.orc_check_leaves.restype = None
    L.orc_check_triangle_idx.argtypes = [C.c_int, vp, vp, C.c_uint32]; L.orc_check_triangle_idx.restype = C.c_uint32
    L.orc_self_collide.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_uint64, C.POINTER(OrcStats), C.POINTER(OrcTimes)]
    L.orc_self_collide.restype = C.c_uint64
    L.orc_rt_hit.argtypes = [vp, C.c_float, C.c_float, C.POINTER(C.c_float), vp]; L.orc_rt_hit.restype = C.c_float
    L.orc_rt_render.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]; L.orc_rt_render.restype = None
    L.orc_rt_render_rows.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.orc_rt_render_rows.restype = None
    L.orc_rt_init_shifts.argtypes = [C.c_int, vp, vp]; L.orc_rt_init_shifts.restype = None
    L.orc_xorwow_init.argtypes = [vp, C.c_uint64]; L.orc_xorwow_init.restype = None
    L.orc_xorwow_next.argtypes = [vp]; L.orc_xorwow_next.restype = C.c_uint32
    L.orc_rt_anim_init.argtypes = [C.c_int, vp, vp, vp]; L.orc_rt_anim_init.restype = None
    L.orc_rt_anim_axis_move.argtypes = [C.c_int, vp, vp, C.c_int]; L.orc_rt_anim_axis_move.restype = None
    L.orc_rt_anim_curve_move.argtypes = [C.c_int, vp, vp]; L.orc_rt_anim_curve_move.restype = None
    L.orc_rt_anim_speed_angle.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.c_int]; L.orc_rt_anim_speed_angle.restype = None
    _libs[name] = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


REF_OFF = np.array([0.004501, -0.476622, -0.381965], dtype=np.float64)
REF_SPAN = np.array([3.08, 0.76, 2.36], dtype=np.float64)


def expand64(v: int) -> int:
    return lib().orc_expand64(v)


def morton3d(x, y, z, off=REF_OFF, span=REF_SPAN) -> int:
    off = np.ascontiguousarray(off, dtype=np.float64); span = np.ascontiguousarray(span, dtype=np.float64)
    return lib().orc_morton3d(x, y, z, _p(off), _p(span))


def morton3d_batch(xyz, off=REF_OFF, span=REF_SPAN):
    p = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    off = np.ascontiguousarray(off, dtype=np.float64); span = np.ascontiguousarray(span, dtype=np.float64)
    keys = np.zeros(p.shape[0], dtype=np.uint64)
    lib().orc_morton3d_batch(_p(p), p.shape[0], _p(off), _p(span), _p(keys))
    return keys


def expand64_batch(v):
    a = np.ascontiguousarray(v, dtype=np.uint64).ravel()
    out = np.zeros(a.shape[0], dtype=np.uint64)
    lib().orc_expand64_batch(_p(a), a.shape[0], _p(out))
    return out


def centroid_morton(verts, vidx, off=REF_OFF, span=REF_SPAN, want_centroids=False):
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    off = np.ascontiguousarray(off, dtype=np.float64); span = np.ascontiguousarray(span, dtype=np.float64)
    n = vidx.shape[0]
    keys = np.zeros(n, dtype=np.uint64)
    cen = np.zeros((n, 3), dtype=np.float64) if want_centroids else None
    lib().orc_centroid_morton(_p(verts), _p(vidx), n, _p(off), _p(span), _p(keys), _p(cen))
    return (keys, cen) if want_centroids else keys


# ---- the adaptive frame (CD_FRAME_AUTO since round 6; not reference behaviour: oracle/cd_oracle.c, "the ADAPTIVE frame")
def frame_layout(lo, hi, stat_sum, stat_cnt) -> int:
    """layout word from the centroids' bounds and the box statistic (per axis: sum of the fixed-point log2 of the triangles' box extents, their number)."""
    lo = np.ascontiguousarray(lo, dtype=np.float64); hi = np.ascontiguousarray(hi, dtype=np.float64)
    ss = np.ascontiguousarray(stat_sum, dtype=np.int64); sc = np.ascontiguousarray(stat_cnt, dtype=np.int64)
    return int(lib().orc_frame_layout(_p(lo), _p(hi), _p(ss), _p(sc)))


def layout_stat(verts, vidx):
    """centroid bounds lo[3], hi[3] and the box statistic sum[3], cnt[3] (int64) of a mesh."""
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    lo = np.zeros(3); hi = np.zeros(3); ss = np.zeros(3, dtype=np.int64); sc = np.zeros(3, dtype=np.int64)
    lib().orc_layout_stat(_p(verts), _p(vidx), vidx.shape[0], _p(lo), _p(hi), _p(ss), _p(sc))
    return lo, hi, ss, sc


def frame_layout_cap(lo, hi, stat_sum, stat_cnt, cap_bits) -> int:
    lo = np.ascontiguousarray(lo, dtype=np.float64); hi = np.ascontiguousarray(hi, dtype=np.float64)
    ss = np.ascontiguousarray(stat_sum, dtype=np.int64); sc = np.ascontiguousarray(stat_cnt, dtype=np.int64)
    return int(lib().orc_frame_layout_cap(_p(lo), _p(hi), _p(ss), _p(sc), int(cap_bits)))


def layout_word(order, nA, nAB, nABC) -> int:
    return (1 << 63) | order[0] | (order[1] << 2) | (order[2] << 4) | (nA << 8) | (nAB << 16) | (nABC << 24)


def layout_fields(layout: int):
    """(axis order A, B, C), nA, nAB, nABC of a layout word; None for 0 (the reference's interleave)."""
    if not layout >> 63:
        return None
    return ((layout & 3, (layout >> 2) & 3, (layout >> 4) & 3), (layout >> 8) & 255, (layout >> 16) & 255, (layout >> 24) & 255)


def auto_frame(verts, vidx):
    """off[3], span[3], layout word of the frame CD_FRAME_AUTO gives this mesh."""
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    fr = np.zeros(6, dtype=np.float64)
    lay = int(lib().orc_auto_frame(_p(verts), _p(vidx), vidx.shape[0], _p(fr)))
    return fr[:3].copy(), fr[3:].copy(), lay


def centroid_morton_layout(verts, vidx, off, span, layout):
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    off = np.ascontiguousarray(off, dtype=np.float64); span = np.ascontiguousarray(span, dtype=np.float64)
    keys = np.zeros(vidx.shape[0], dtype=np.uint64)
    lib().orc_centroid_morton_layout(_p(verts), _p(vidx), vidx.shape[0], _p(off), _p(span), int(layout), _p(keys))
    return keys


def morton3d_layout_batch(xyz, off, span, layout):
    """keys of explicit points in a frame with a layout; xyz are then vertex SUMS p1 + p2 + p3 (layout 0: centroids, morton.h:70-89)"""
    p = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    off = np.ascontiguousarray(off, dtype=np.float64); span = np.ascontiguousarray(span, dtype=np.float64)
    keys = np.zeros(p.shape[0], dtype=np.uint64)
    lib().orc_morton3d_layout_batch(_p(p), p.shape[0], _p(off), _p(span), int(layout), _p(keys))
    return keys


def sort_by_key(keys):
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    perm = np.zeros(k.shape[0], dtype=np.uint32)
    lib().orc_sort_by_key(_p(k), _p(perm), k.shape[0])
    return k, perm


def determine_range(keys, i, tiebreak=0):
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    a, b = C.c_int(), C.c_int()
    lib().orc_determine_range(_p(k), k.shape[0], i, tiebreak, C.byref(a), C.byref(b))
    return a.value, b.value


def find_split(keys, first, last, tiebreak=0):
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    return lib().orc_find_split(_p(k), k.shape[0], first, last, tiebreak)


def build_hierarchy(keys, tiebreak=1):
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    n = k.shape[0]
    left = np.zeros(max(n - 1, 0), dtype=np.int32); right = np.zeros(max(n - 1, 0), dtype=np.int32)
    parent = np.zeros(2 * n - 1, dtype=np.int32)
    rf = np.zeros(max(n - 1, 0), dtype=np.int32); rl = np.zeros(max(n - 1, 0), dtype=np.int32)
    wrong = C.c_uint32(0)
    lib().orc_build_hierarchy(_p(k), n, tiebreak, _p(left), _p(right), _p(parent), _p(rf), _p(rl), C.byref(wrong))
    return left, right, parent, rf, rl, wrong.value


def refit(verts, vidx, perm, left, right, parent):
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    perm = np.ascontiguousarray(perm, dtype=np.uint32)
    n = perm.shape[0]
    boxes = np.zeros((2 * n - 1, 6), dtype=np.float64)
    bounded = np.zeros(max(n - 1, 0), dtype=np.uint32)
    cc = np.zeros(2 * n - 1, dtype=np.uint32)
    lib().orc_refit(_p(verts), _p(vidx), _p(perm), n, _p(left), _p(right), _p(parent), _p(boxes), _p(bounded), _p(cc))
    return boxes, bounded, cc


def tri_contact(P, Q) -> int:
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(3, 3); Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(3, 3)
    rows = [np.ascontiguousarray(r) for r in (P[0], P[1], P[2], Q[0], Q[1], Q[2])]
    return lib().orc_tri_contact(*[_p(r) for r in rows])


def tri_contact_batch(verts, vidx, pairs, ids=None):
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
    out = np.zeros(pairs.shape[0], dtype=np.uint8)
    lib().orc_tri_contact_batch(_p(verts), _p(vidx), _p(ids), _p(pairs), pairs.shape[0], _p(out))
    return out


# ---- batch forms of the predicates on explicit operands (replayed against tests/golden/contact_ref.npz)
def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _cu32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def tri_contact_points(tri):
    tri = _c64(tri).reshape(-1, 18); out = np.zeros(tri.shape[0], dtype=np.int32)
    lib().orc_tri_contact_points_batch(_p(tri), tri.shape[0], _p(out)); return out


def helper_batch(verts, va, ida, vb, idb):
    verts, va, ida, vb, idb = _c64(verts), _cu32(va), _cu32(ida), _cu32(vb), _cu32(idb)
    out = np.zeros(va.shape[0], dtype=np.int32)
    lib().orc_helper_batch(_p(verts), _p(va), _p(ida), _p(vb), _p(idb), va.shape[0], _p(out)); return out


def neighbor_count_batch(va, vb):
    va, vb = _cu32(va), _cu32(vb); out = np.zeros(va.shape[0], dtype=np.int32)
    lib().orc_neighbor_count_batch(_p(va), _p(vb), va.shape[0], _p(out)); return out


def box_set_batch(verts, vidx):
    verts, vidx = _c64(verts), _cu32(vidx); out = np.zeros((vidx.shape[0], 6), dtype=np.float64)
    lib().orc_box_set_batch(_p(verts), _p(vidx), vidx.shape[0], _p(out)); return out


def box_merge_batch(a, b):
    a, b = _c64(a), _c64(b); out = np.zeros_like(a)
    lib().orc_box_merge_batch(_p(a), _p(b), a.shape[0], _p(out)); return out


def box_overlap_batch(a, b):
    a, b = _c64(a), _c64(b); out = np.zeros(a.shape[0], dtype=np.int32)
    lib().orc_box_overlap_batch(_p(a), _p(b), a.shape[0], _p(out)); return out


def project3_batch(v):
    v = _c64(v); out = np.zeros(v.shape[0], dtype=np.int32)
    lib().orc_project3_batch(_p(v), v.shape[0], _p(out)); return out


def project6_batch(v):
    v = _c64(v); out = np.zeros(v.shape[0], dtype=np.int32)
    lib().orc_project6_batch(_p(v), v.shape[0], _p(out)); return out


def cross_dot_batch(v):
    v = _c64(v); cr = np.zeros((v.shape[0], 3), dtype=np.float64); dt = np.zeros(v.shape[0], dtype=np.float64)
    lib().orc_cross_dot_batch(_p(v), v.shape[0], _p(cr), _p(dt)); return cr, dt


def find_collisions(verts, vidx, perm, left, right, boxes, ids=None, cap=1 << 22):
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    perm = np.ascontiguousarray(perm, dtype=np.uint32); boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
    pairs = np.zeros((cap, 2), dtype=np.uint32)
    st = OrcStats()
    lib().orc_find_collisions(_p(verts), _p(vidx), _p(ids), _p(perm), perm.shape[0], _p(left), _p(right), _p(boxes), _p(pairs), cap, C.byref(st))
    return pairs[:min(st.n_pairs, cap)], st


def find_collisions_queries(queries, verts, vidx, perm, left, right, boxes, ids=None, cap=1 << 22, vbase=0):
    """queries: structured array with fields v[9], id, vidx[3] (cd_query layout)."""
    qv = np.ascontiguousarray(queries["v"], dtype=np.float64)
    qx = np.ascontiguousarray(queries["vidx"], dtype=np.uint32)
    qi = np.ascontiguousarray(queries["id"], dtype=np.uint32)
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    perm = np.ascontiguousarray(perm, dtype=np.uint32); boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
    pairs = np.zeros((cap, 2), dtype=np.uint32)
    st = OrcStats()
    lib().orc_find_collisions_queries(_p(qv), _p(qx), _p(qi), qi.shape[0], _p(verts), _p(vidx), _p(ids), _p(perm), perm.shape[0],
                                      _p(left), _p(right), _p(boxes), int(vbase), _p(pairs), cap, C.byref(st))
    return pairs[:min(st.n_pairs, cap)], st


def brute_force(verts, vidx, ids=None, box_filter=True, cap=1 << 22):
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
    pairs = np.zeros((cap, 2), dtype=np.uint32)
    tested = C.c_uint64(0)
    n = lib().orc_brute_force(_p(verts), _p(vidx), _p(ids), vidx.shape[0], 1 if box_filter else 0, _p(pairs), cap, C.byref(tested))
    return pairs[:min(n, cap)], n, tested.value


def self_collide(verts, vidx, ids=None, off=REF_OFF, span=REF_SPAN, threads=1, cap=1 << 22, want_pairs=True):
    verts = np.ascontiguousarray(verts, dtype=np.float64); vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
    off = np.ascontiguousarray(off, dtype=np.float64); span = np.ascontiguousarray(span, dtype=np.float64)
    pairs = np.zeros((cap, 2), dtype=np.uint32) if (want_pairs and threads <= 1) else None
    st, tm = OrcStats(), OrcTimes()
    L = lib(omp=threads > 1)
    L.orc_self_collide(_p(verts), _p(vidx), _p(ids), vidx.shape[0], _p(off), _p(span), threads, _p(pairs), cap if pairs is not None else 0,
                       C.byref(st), C.byref(tm))
    return (pairs[:min(st.n_pairs, cap)] if pairs is not None else None), st, tm


def pipeline(verts, vidx, ids=None, off=REF_OFF, span=REF_SPAN, tiebreak=1, layout=0):
    """Stage-by-stage oracle run; returns a dict with every intermediate for parity tests.  layout: the key layout of an
    adaptive frame (auto_frame); 0 = the reference's interleave (morton.h:70-89)."""
    keys0 = centroid_morton_layout(verts, vidx, off, span, layout) if layout else centroid_morton(verts, vidx, off, span)
    keys, perm = sort_by_key(keys0)
    left, right, parent, rf, rl, wrong = build_hierarchy(keys, tiebreak)
    boxes, bounded, cc = refit(verts, vidx, perm, left, right, parent)
    pairs, st = find_collisions(verts, vidx, perm, left, right, boxes, ids)
    return dict(keys_unsorted=keys0, keys=keys, perm=perm, left=left, right=right, parent=parent, range_first=rf,
                range_last=rl, parent_wrong=wrong, boxes=boxes, bounded=bounded, child_count=cc, pairs=pairs, stats=st)


def pair_set(pairs) -> np.ndarray:
    """Canonical form for set comparison: rows sorted lexicographically, as uint64 keys."""
    p = np.asarray(pairs, dtype=np.uint64).reshape(-1, 2)
    return np.sort((p[:, 0] << np.uint64(32)) | p[:, 1])


# ---- ray tracer
def rt_hit(sphere_rec, ox, oy, shifts):
    s = np.ascontiguousarray(sphere_rec); sh = np.ascontiguousarray(shifts, dtype=np.int32)
    n = C.c_float(0)
    t = lib().orc_rt_hit(_p(s), C.c_float(ox), C.c_float(oy), C.byref(n), _p(sh))
    return t, n.value


def rt_render(spheres, shifts, dim, c_shift_x=0, c_shift_y=0, rows=None):
    s = np.ascontiguousarray(spheres); sh = np.ascontiguousarray(shifts, dtype=np.int32)
    img = np.zeros((dim, dim, 4), dtype=np.uint8)
    if rows is None:
        lib().orc_rt_render(_p(s), s.shape[0], _p(sh), dim, c_shift_x, c_shift_y, _p(img))
    else:
        lib().orc_rt_render_rows(_p(s), s.shape[0], _p(sh), dim, c_shift_x, c_shift_y, rows[0], rows[1], _p(img))
    return img


class RtAnim:
    """Oracle twin of the animation state kernels (sphere.cuh:50-118): rng[n,6] uint32, shifts[n,4] int32, angles[n] f64."""

    def __init__(self, n):
        self.n = n
        self.rng = np.zeros((n, 6), dtype=np.uint32); self.shifts = np.zeros((n, 4), dtype=np.int32); self.angles = np.zeros(n, dtype=np.float64)
        lib().orc_rt_anim_init(n, _p(self.rng), _p(self.shifts), _p(self.angles))

    def axis_move(self, shake_width=35):
        lib().orc_rt_anim_axis_move(self.n, _p(self.rng), _p(self.shifts), shake_width)

    def curve_move(self):
        lib().orc_rt_anim_curve_move(self.n, _p(self.shifts), _p(self.angles))

    def speed_angle(self, update_prob=1, max_speed=18):
        lib().orc_rt_anim_speed_angle(self.n, _p(self.rng), _p(self.shifts), _p(self.angles), update_prob, max_speed)


def xorwow_stream(seed, count):
    st = np.zeros(6, dtype=np.uint32)
    lib().orc_xorwow_init(_p(st), seed)
    return np.array([lib().orc_xorwow_next(_p(st)) for _ in range(count)], dtype=np.uint32), st
