"""ctypes binding of libmi355cd.so (include/mi355cd.h) -- plumbing for tests/ and bench.py.

This is NOT a second implementation: every method is one call through the C ABI.  There is no CPU
fallback; if the shared library is missing or no HIP device is present the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libmi355cd.so")

CD_OK, CD_OVERFLOW = 0, 1
CD_ERR_ARG, CD_ERR_ORDER, CD_ERR_NO_DEVICE, CD_ERR_INDEX = -1001, -1002, -1003, -1004
CD_FRAME_REFERENCE, CD_FRAME_AUTO, CD_FRAME_CUSTOM = 0, 1, 2
CD_ERR_SORT, CD_ERR_IO, CD_ERR_FORMAT = -1005, -1006, -1007
CD_OPT_TRAVERSAL, CD_OPT_QUERIES_PER_WAVE, CD_OPT_SORT_FULL, CD_OPT_STAGE_TIMING, CD_OPT_KERNEL_STAMPS, CD_OPT_GRAPH, CD_OPT_POLL = 0, 1, 2, 3, 4, 5, 6
CD_OPT_CELL_TABLE = 7
CD_OPT_ORDER_HINT = 8
# cd_debug_option keys (measurement hooks / test switches; not part of the mirrored interface)
CD_DBG_LDS_PAD, CD_DBG_EXACT_BLOCKS, CD_DBG_NO_SHARED_PATH, CD_DBG_DIAG, CD_DBG_STAGEWISE_BUILD, CD_DBG_SPLIT_CROSS = 0, 1, 2, 3, 4, 5
CD_DBG_SORT_WINDOWS, CD_DBG_GET_SORT_FORM = 7, 8
CD_DBG_POLL_SCAN, CD_DBG_GET_POLL_STALE, CD_DBG_GET_POLL_FALLBACKS, CD_DBG_GET_POLLED_STEPS, CD_DBG_GET_TREE_WAS_FUSED = 10, 11, 12, 13, 14
CD_DBG_GET_ORDER_STATE = 15
CD_DBG_GET_POLL_FB_WHY, CD_DBG_GET_POLL_MAX_WAIT_US = 16, 17
CD_DBG_REPORT_COPIES = 6
CD_DBG_STORE_QBOX = 9
CD_DBG_BIG_OFFSETS = 18

QUERY_DTYPE = np.dtype([("v", "<f8", (9,)), ("id", "<u4"), ("vidx", "<u4", (3,))])
assert QUERY_DTYPE.itemsize == 88


class CdStats(C.Structure):
    _fields_ = [("ms_morton", C.c_float), ("ms_sort", C.c_float), ("ms_hierarchy", C.c_float),
                ("ms_refit", C.c_float), ("ms_traverse", C.c_float), ("ms_check", C.c_float),
                ("traverse_launches", C.c_uint32), ("stack_overflows", C.c_uint32),
                ("n_pairs", C.c_uint64), ("pairs_tested", C.c_uint64), ("node_visits", C.c_uint64),
                ("wave_steps", C.c_uint64), ("candidates", C.c_uint64),
                ("ms_descend", C.c_float), ("ms_exact", C.c_float), ("sort_passes", C.c_uint32), ("ms_pipeline", C.c_float),
                ("ms_build_block", C.c_float), ("ms_descend_clock", C.c_float)]


# every symbol include/mi355cd.h declares (tests check the library exports exactly these)
CD_MULTI_SELF_PEER, CD_MULTI_TIMING, CD_MULTI_SELF_SLICE, CD_MULTI_CROSS_SERIAL, CD_MULTI_INJECT_FAILURE, CD_MULTI_PRIORITY_STREAM = 1, 2, 4, 8, 16, 32
CD_MULTI_INJECT_ALLOC_FAILURE = 64
CD_ERR_RCCL, CD_ERR_PEER, CD_ERR_INJECTED = -1008, -1009, -1010


class CdMultiInfo(C.Structure):
    _fields_ = [("world", C.c_uint32), ("rank", C.c_uint32), ("n_peers", C.c_uint32), ("host_syncs", C.c_uint32), ("attempts", C.c_uint32),
                ("failed_rank_plus1", C.c_uint32), ("sent_queries", C.c_uint64), ("recv_queries", C.c_uint64), ("local_pairs", C.c_uint64),
                ("cross_pairs", C.c_uint64), ("pairs_tested", C.c_uint64), ("query_cap", C.c_uint64),
                ("ms_tree", C.c_float), ("ms_allgather", C.c_float), ("ms_pack", C.c_float), ("ms_counts", C.c_float),
                ("ms_exchange", C.c_float), ("ms_local", C.c_float), ("ms_cross", C.c_float), ("pad1", C.c_float)]


EXPORTS = [
    "cd_load_obj", "cd_free_obj", "cd_create", "cd_destroy", "cd_update_vertices", "cd_set_morton_frame", "cd_get_morton_frame", "cd_set_morton_frame_layout", "cd_morton_sort",
    "cd_build_hierarchy", "cd_refit_boxes", "cd_check_internal", "cd_check_leaves",
    "cd_check_triangle_idx", "cd_find_collisions", "cd_build_tree", "cd_self_collide", "cd_sorted_pairs", "cd_collision_triangles", "cd_brute_force",
    "cd_test_pairs", "cd_export_keys", "cd_export_tree", "cd_get_stats", "cd_debug_counters", "cd_debug_records", "cd_num_triangles",
    "cd_set_option", "cd_set_vertex_id_base", "cd_root_box", "cd_pack_queries", "cd_find_collisions_queries", "cd_version",
    "cd_debug_option", "cd_debug_hint", "cd_debug_hint_set", "cd_morton3d_points", "cd_morton3d_points_layout", "cd_expand64_values", "cd_box_pairs", "cd_tri_contact_points", "cd_alloc_host_pairs", "cd_free_host_pairs",
    "cd_multi_unique_id", "cd_multi_create", "cd_multi_create_from_comm", "cd_multi_destroy", "cd_multi_set_flags", "cd_multi_step",
]

_lib = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    global _lib
    if _lib is None:
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 (same sonames as /opt/rocm's).  Two HIP
        # runtimes in one process do not work -- the second one finds no GPU -- so when torch is installed its copy is
        # loaded first and serves this library too.  (Plumbing only: nothing here computes with torch.)
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if _lib is not None:
        return _lib
    path = os.environ.get("MI355CD_LIB", path)          # A/B runs of two builds (tools/)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not built: run __graft_entry__.build() (there is no CPU fallback)")
    lib = C.CDLL(path)
    vp, u32p, u64p, i32p, dp = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    lib.cd_load_obj.argtypes = [C.c_char_p, C.POINTER(dp), u32p, C.POINTER(u32p), u32p, C.c_int]
    lib.cd_free_obj.argtypes = [dp, u32p]
    lib.cd_free_obj.restype = None
    lib.cd_create.argtypes = [C.POINTER(vp), vp, C.c_uint32, vp, vp, C.c_uint32]
    lib.cd_destroy.argtypes = [vp]
    lib.cd_destroy.restype = None
    lib.cd_update_vertices.argtypes = [vp, vp]
    lib.cd_set_morton_frame.argtypes = [vp, C.c_int, vp, vp]
    lib.cd_get_morton_frame.argtypes = [vp, vp, vp, u64p]
    lib.cd_set_morton_frame_layout.argtypes = [vp, vp, vp, C.c_uint64]
    lib.cd_morton_sort.argtypes = [vp]
    lib.cd_build_hierarchy.argtypes = [vp, u32p]
    lib.cd_refit_boxes.argtypes = [vp]
    lib.cd_check_internal.argtypes = [vp, vp]
    lib.cd_check_leaves.argtypes = [vp, vp]
    lib.cd_check_triangle_idx.argtypes = [vp, C.c_uint32, u32p]
    lib.cd_find_collisions.argtypes = [vp, vp, C.c_uint64, u64p]
    lib.cd_build_tree.argtypes = [vp]
    lib.cd_self_collide.argtypes = [vp, vp, C.c_uint64, u64p]
    lib.cd_sorted_pairs.argtypes = [vp, vp, C.c_uint64, u64p]
    lib.cd_collision_triangles.argtypes = [vp, vp, C.c_uint64, u64p]
    lib.cd_brute_force.argtypes = [vp, C.c_int, vp, C.c_uint64, u64p]
    lib.cd_test_pairs.argtypes = [vp, vp, C.c_uint64, vp]
    lib.cd_export_keys.argtypes = [vp, vp, vp]
    lib.cd_export_tree.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.cd_get_stats.argtypes = [vp, C.POINTER(CdStats)]
    lib.cd_debug_counters.argtypes = [vp, vp]
    lib.cd_debug_records.argtypes = [vp, vp, vp, vp]
    lib.cd_debug_hint.argtypes = [vp, vp, vp, vp]
    lib.cd_debug_hint_set.argtypes = [vp, vp]
    lib.cd_num_triangles.argtypes = [vp, u32p]
    lib.cd_set_option.argtypes = [vp, C.c_int, C.c_int64]
    lib.cd_set_vertex_id_base.argtypes = [vp, C.c_uint32]
    lib.cd_root_box.argtypes = [vp, vp]
    lib.cd_pack_queries.argtypes = [vp, vp, vp, C.c_uint64, u64p]
    lib.cd_find_collisions_queries.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, u64p]
    lib.cd_version.restype = C.c_char_p
    lib.cd_alloc_host_pairs.argtypes = [C.c_uint64, C.POINTER(u32p)]
    lib.cd_free_host_pairs.argtypes = [u32p]
    lib.cd_free_host_pairs.restype = None
    lib.cd_morton3d_points.argtypes = [vp, C.c_uint64, vp, vp, vp]
    lib.cd_morton3d_points_layout.argtypes = [vp, C.c_uint64, vp, vp, C.c_uint64, vp]
    lib.cd_expand64_values.argtypes = [vp, C.c_uint64, vp]
    lib.cd_debug_option.argtypes = [vp, C.c_int, C.c_int64, C.POINTER(C.c_int64)]
    lib.cd_box_pairs.argtypes = [vp, vp, C.c_uint64, vp, vp]
    lib.cd_tri_contact_points.argtypes = [vp, C.c_uint64, vp]
    lib.cd_multi_unique_id.argtypes = [vp]
    lib.cd_multi_create.argtypes = [C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_uint64, C.c_int]
    lib.cd_multi_create_from_comm.argtypes = [C.POINTER(vp), vp, vp, C.c_uint64, C.c_int]
    lib.cd_multi_destroy.argtypes = [vp]
    lib.cd_multi_destroy.restype = None
    lib.cd_multi_set_flags.argtypes = [vp, C.c_int]
    lib.cd_multi_step.argtypes = [vp, vp, C.c_uint64, u64p, C.POINTER(CdMultiInfo)]
    for name in EXPORTS:
        if name not in ("cd_destroy", "cd_version", "cd_free_obj", "cd_multi_destroy", "cd_free_host_pairs"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


class CdError(RuntimeError):
    def __init__(self, fn: str, rc: int):
        super().__init__(f"{fn} failed with status {rc}")
        self.rc = rc


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class CollisionDetector:
    """One cd_ctx.  Methods mirror the reference harness's stage order (main.cu:64-146)."""

    def __init__(self, verts: np.ndarray, vidx: np.ndarray, ids: np.ndarray | None = None):
        self.lib = load_library()
        self.verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        self.vidx = np.ascontiguousarray(vidx, dtype=np.uint32).reshape(-1, 3)
        self.ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
        self.nv, self.nt = self.verts.shape[0], self.vidx.shape[0]
        self._ctx = C.c_void_p()
        rc = self.lib.cd_create(C.byref(self._ctx), _ptr(self.verts), self.nv, _ptr(self.vidx), _ptr(self.ids), self.nt)
        if rc != CD_OK:
            self._ctx = C.c_void_p()
            raise CdError("cd_create", rc)

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.cd_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, fn, rc, allow=(CD_OK,)):
        if rc not in allow:
            raise CdError(fn, rc)
        return rc

    # ---- stages
    def set_morton_frame(self, mode=CD_FRAME_REFERENCE, offset=None, span=None):
        off = None if offset is None else np.ascontiguousarray(offset, dtype=np.float64)
        sp = None if span is None else np.ascontiguousarray(span, dtype=np.float64)
        self._chk("cd_set_morton_frame", self.lib.cd_set_morton_frame(self._ctx, mode, _ptr(off), _ptr(sp)))

    def get_morton_frame(self):
        """(offset[3], span[3], layout word) of the frame the last sort used (cd_get_morton_frame)."""
        off = np.zeros(3, dtype=np.float64); sp = np.zeros(3, dtype=np.float64); lay = C.c_uint64(0)
        self._chk("cd_get_morton_frame", self.lib.cd_get_morton_frame(self._ctx, _ptr(off), _ptr(sp), C.byref(lay)))
        return off, sp, int(lay.value)

    def set_morton_frame_layout(self, offset, span, layout: int):
        off = np.ascontiguousarray(offset, dtype=np.float64); sp = np.ascontiguousarray(span, dtype=np.float64)
        self._chk("cd_set_morton_frame_layout", self.lib.cd_set_morton_frame_layout(self._ctx, _ptr(off), _ptr(sp), int(layout)))

    def keep_auto_frame(self):
        """The frame CD_FRAME_AUTO computed in the last sort becomes the context's fixed frame (the AUTO pass over the triangles leaves the step)."""
        off, sp, lay = self.get_morton_frame()
        self.set_morton_frame_layout(off, sp, lay)
        return off, sp, lay

    def set_option(self, key: int, value: int):
        self._chk("cd_set_option", self.lib.cd_set_option(self._ctx, key, value))

    def debug_set(self, key: int, value: int):
        self._chk("cd_debug_option", self.lib.cd_debug_option(self._ctx, key, value, None))

    def debug_get(self, key: int) -> int:
        out = C.c_int64(0)
        self._chk("cd_debug_option", self.lib.cd_debug_option(self._ctx, key, 0, C.byref(out)))
        return out.value

    def update_vertices(self, verts):
        v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        assert v.shape[0] == self.nv
        self.verts = v
        self._chk("cd_update_vertices", self.lib.cd_update_vertices(self._ctx, _ptr(v)))

    def morton_sort(self):
        self._chk("cd_morton_sort", self.lib.cd_morton_sort(self._ctx))

    def build_hierarchy(self) -> int:
        w = C.c_uint32(0)
        self._chk("cd_build_hierarchy", self.lib.cd_build_hierarchy(self._ctx, C.byref(w)))
        return w.value

    def build_tree(self):
        self._chk("cd_build_tree", self.lib.cd_build_tree(self._ctx))

    def refit_boxes(self):
        self._chk("cd_refit_boxes", self.lib.cd_refit_boxes(self._ctx))

    def check_internal(self):
        out = np.zeros(5, dtype=np.uint32)
        self._chk("cd_check_internal", self.lib.cd_check_internal(self._ctx, _ptr(out)))
        return out

    def check_leaves(self):
        out = np.zeros(4, dtype=np.uint32)
        self._chk("cd_check_leaves", self.lib.cd_check_leaves(self._ctx, _ptr(out)))
        return out

    def check_triangle_idx(self, maxv: int) -> int:
        out = C.c_uint32(0)
        self._chk("cd_check_triangle_idx", self.lib.cd_check_triangle_idx(self._ctx, maxv, C.byref(out)))
        return out.value

    def _pairs_call(self, fn, name, cap, *pre, copy=True):
        n = C.c_uint64(0)
        # the output buffer is kept between calls (a 4 M-pair buffer is 32 MB; allocating it per step costs more
        # than the step); callers get a copy of the filled prefix -- or, with copy=False, a VIEW of the buffer the C
        # call wrote into, valid until the next call on this object (what a C caller has; a per-frame loop needs no more)
        buf = None
        if cap:
            if getattr(self, "_pairbuf", None) is None or self._pairbuf.shape[0] != cap:
                self._pairbuf = np.empty((cap, 2), dtype=np.uint32)
                self._pairptr = _ptr(self._pairbuf)
            buf = self._pairbuf
        rc = fn(self._ctx, *pre, self._pairptr if cap else None, cap, C.byref(n))
        self._chk(name, rc, allow=(CD_OK, CD_OVERFLOW))
        got = min(n.value, cap)
        if buf is None:
            return np.zeros((0, 2), dtype=np.uint32), n.value, rc
        return (buf[:got].copy() if copy else buf[:got]), n.value, rc

    def find_collisions(self, cap: int = 1 << 20):
        return self._pairs_call(self.lib.cd_find_collisions, "cd_find_collisions", cap)

    def self_collide(self, cap: int = 1 << 20, copy: bool = True):
        return self._pairs_call(self.lib.cd_self_collide, "cd_self_collide", cap, copy=copy)

    def self_collide_into(self, buf: np.ndarray):
        """cd_self_collide and cd_get_stats with every ctypes argument built ONCE (a per-frame loop: the marshalling of the
        generic path costs several microseconds a step).  buf: caller-owned uint32[cap, 2], reused.  Returns (n_pairs, rc);
        the pairs are buf[:min(n_pairs, cap)], the statistics self.fast_stats (a CdStats refreshed by every call)."""
        fc = getattr(self, "_fast", None)
        if fc is None or fc[0] is not buf:
            n = C.c_uint64(0)
            self.fast_stats = CdStats()
            fc = self._fast = (buf, _ptr(buf), C.c_uint64(buf.shape[0]), n, C.byref(n), C.byref(self.fast_stats),
                               self.lib.cd_self_collide, self.lib.cd_get_stats)
        rc = fc[6](self._ctx, fc[1], fc[2], fc[4])
        if rc != CD_OK and rc != CD_OVERFLOW:
            raise CdError("cd_self_collide", rc)
        fc[7](self._ctx, fc[5])
        return fc[3].value, rc

    def sorted_pairs(self, cap: int = 1 << 20):
        """Pair list of the last traversal, sorted by (a, b) on the device."""
        return self._pairs_call(self.lib.cd_sorted_pairs, "cd_sorted_pairs", cap)

    def collision_triangles(self, cap: int = 1 << 21):
        """Sorted distinct triangle IDs of the last traversal's pairs (main.cu:33-45), built on the device."""
        n = C.c_uint64(0)
        buf = np.empty(cap, dtype=np.uint32)
        rc = self.lib.cd_collision_triangles(self._ctx, _ptr(buf), cap, C.byref(n))
        self._chk("cd_collision_triangles", rc, allow=(CD_OK, CD_OVERFLOW))
        return buf[:min(n.value, cap)].copy(), n.value, rc

    def brute_force(self, box_filter: bool = True, cap: int = 1 << 20):
        return self._pairs_call(self.lib.cd_brute_force, "cd_brute_force", cap, 1 if box_filter else 0)

    def test_pairs(self, pairs: np.ndarray) -> np.ndarray:
        p = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        out = np.zeros(p.shape[0], dtype=np.uint8)
        self._chk("cd_test_pairs", self.lib.cd_test_pairs(self._ctx, _ptr(p), p.shape[0], _ptr(out)))
        return out

    # ---- read-back
    def export_keys(self):
        keys = np.zeros(self.nt, dtype=np.uint64)
        perm = np.zeros(self.nt, dtype=np.uint32)
        self._chk("cd_export_keys", self.lib.cd_export_keys(self._ctx, _ptr(keys), _ptr(perm)))
        return keys, perm

    def export_tree(self, with_boxes: bool = True):
        n = self.nt
        parent = np.zeros(2 * n - 1, dtype=np.int32)
        left = np.zeros(max(n - 1, 0), dtype=np.int32)
        right = np.zeros(max(n - 1, 0), dtype=np.int32)
        boxes = np.zeros((2 * n - 1, 6), dtype=np.float64) if with_boxes else None
        bounded = np.zeros(max(n - 1, 0), dtype=np.uint32) if with_boxes else None
        self._chk("cd_export_tree", self.lib.cd_export_tree(self._ctx, _ptr(parent), _ptr(left), _ptr(right), _ptr(boxes), _ptr(bounded)))
        return parent, left, right, boxes, bounded

    def stats(self) -> CdStats:
        s = CdStats()
        self._chk("cd_get_stats", self.lib.cd_get_stats(self._ctx, C.byref(s)))
        return s

    def debug_counters(self) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        self._chk("cd_debug_counters", self.lib.cd_debug_counters(self._ctx, out.ctypes.data))
        return out

    def debug_hint(self, with_order: bool = True, with_tri: bool = False):
        g = (self.nt + 63) // 64
        cost = np.zeros(g, dtype=np.uint32); order = np.zeros(g, dtype=np.uint32); tri = np.zeros(self.nt, dtype=np.uint8)
        self._chk("cd_debug_hint", self.lib.cd_debug_hint(self._ctx, _ptr(cost), _ptr(order) if with_order else None, _ptr(tri) if with_tri else None))
        return (cost, (order if with_order else None), tri) if with_tri else (cost, (order if with_order else None))

    def debug_hint_set(self, order: np.ndarray):
        o = np.ascontiguousarray(order, dtype=np.uint32)
        self._chk("cd_debug_hint_set", self.lib.cd_debug_hint_set(self._ctx, _ptr(o)))

    def debug_records(self):
        """(right halves u32[n, 8], left halves u32[n, 8], query boxes u32[n, 8], root split) of the current tree."""
        n = self.nt
        recs = np.zeros((2 * n, 8), dtype=np.uint32)
        qb = np.zeros((n, 8), dtype=np.uint32)
        root = C.c_int32(0)
        self._chk("cd_debug_records", self.lib.cd_debug_records(self._ctx, recs.ctypes.data, qb.ctypes.data, C.byref(root)))
        return recs[:n], recs[n:], qb, int(root.value)

    # ---- cross-rank pass
    def set_vertex_id_base(self, base: int):
        self._chk("cd_set_vertex_id_base", self.lib.cd_set_vertex_id_base(self._ctx, base))

    def root_box(self) -> np.ndarray:
        b = np.zeros(6, dtype=np.float64)
        self._chk("cd_root_box", self.lib.cd_root_box(self._ctx, _ptr(b)))
        return b

    def pack_queries_into(self, box, d_out_ptr: int, cap: int):
        """Compact overlapping leaves into a caller-owned DEVICE buffer (e.g. torch tensor data_ptr)."""
        b = np.ascontiguousarray(box, dtype=np.float64)
        n = C.c_uint64(0)
        rc = self.lib.cd_pack_queries(self._ctx, _ptr(b), C.c_void_p(d_out_ptr), cap, C.byref(n))
        self._chk("cd_pack_queries", rc, allow=(CD_OK, CD_OVERFLOW))
        return n.value, rc

    def find_collisions_queries(self, d_queries_ptr: int, nq: int, cap: int = 1 << 20):
        return self._pairs_call(self.lib.cd_find_collisions_queries, "cd_find_collisions_queries", cap,
                                C.c_void_p(d_queries_ptr), C.c_uint64(nq))


def load_obj(path: str, threads: int = 0):
    """cd_load_obj through the C ABI -> (verts float64[V,3], vidx uint32[N,3]).  Host only, needs no GPU."""
    lib = load_library()
    pv, pf = C.POINTER(C.c_double)(), C.POINTER(C.c_uint32)()
    nv, nt = C.c_uint32(0), C.c_uint32(0)
    rc = lib.cd_load_obj(path.encode(), C.byref(pv), C.byref(nv), C.byref(pf), C.byref(nt), threads)
    if rc != CD_OK:
        raise CdError("cd_load_obj", rc)
    try:
        verts = np.ctypeslib.as_array(pv, shape=(nv.value, 3)).copy()
        vidx = np.ctypeslib.as_array(pf, shape=(nt.value, 3)).copy()
    finally:
        lib.cd_free_obj(pv, pf)
    return verts, vidx


class HostPairs:
    """cd_alloc_host_pairs: a uint32[cap, 2] array in pinned host memory the library lets the GPU write straight into
    (.array; hand it to CollisionDetector.self_collide_into).  Freed by close() / the context manager."""

    def __init__(self, cap: int):
        self.lib = load_library()
        self._p = C.POINTER(C.c_uint32)()
        rc = self.lib.cd_alloc_host_pairs(cap, C.byref(self._p))
        if rc != CD_OK:
            raise CdError("cd_alloc_host_pairs", rc)
        self.array = np.ctypeslib.as_array(self._p, shape=(cap, 2))

    def close(self):
        if self._p:
            self.array = None
            self.lib.cd_free_host_pairs(self._p)
            self._p = C.POINTER(C.c_uint32)()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def morton3d_points(xyz, offset=None, span=None) -> np.ndarray:
    """morton.h:70-89 morton3D on explicit points, on the device (cd_morton3d_points)."""
    p = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    off = None if offset is None else np.ascontiguousarray(offset, dtype=np.float64)
    sp = None if span is None else np.ascontiguousarray(span, dtype=np.float64)
    keys = np.zeros(p.shape[0], dtype=np.uint64)
    rc = load_library().cd_morton3d_points(_ptr(p), p.shape[0], _ptr(off), _ptr(sp), _ptr(keys))
    if rc != CD_OK:
        raise CdError("cd_morton3d_points", rc)
    return keys


def morton3d_points_layout(xyz, offset, span, layout: int) -> np.ndarray:
    """The key of explicit points in a frame with a key layout (cd_morton3d_points_layout; layout 0 = morton3d_points)."""
    p = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    off = np.ascontiguousarray(offset, dtype=np.float64); sp = np.ascontiguousarray(span, dtype=np.float64)
    keys = np.zeros(p.shape[0], dtype=np.uint64)
    rc = load_library().cd_morton3d_points_layout(_ptr(p), p.shape[0], _ptr(off), _ptr(sp), int(layout), _ptr(keys))
    if rc != CD_OK:
        raise CdError("cd_morton3d_points_layout", rc)
    return keys


def expand64_values(v) -> np.ndarray:
    """morton.h:7-29 expand64Bits on explicit values, on the device (cd_expand64_values)."""
    a = np.ascontiguousarray(v, dtype=np.uint64).ravel()
    out = np.zeros(a.shape[0], dtype=np.uint64)
    rc = load_library().cd_expand64_values(_ptr(a), a.shape[0], _ptr(out))
    if rc != CD_OK:
        raise CdError("cd_expand64_values", rc)
    return out


def box_pairs(a, b, want_merged=True):
    """box.cuh:40-43 checkBoxOverlap and box.cuh:24-32 Box::merge on explicit boxes, on the device (cd_box_pairs)."""
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 6); b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, 6)
    ov = np.zeros(a.shape[0], dtype=np.uint8)
    mg = np.zeros_like(a) if want_merged else None
    rc = load_library().cd_box_pairs(_ptr(a), _ptr(b), a.shape[0], _ptr(ov), _ptr(mg))
    if rc != CD_OK:
        raise CdError("cd_box_pairs", rc)
    return ov, mg


def tri_contact_points(tri) -> np.ndarray:
    """tri_contact.cuh:19-78 checkTriangleContact on explicit vertex positions [n, 6, 3], on the device (cd_tri_contact_points)."""
    t = np.ascontiguousarray(tri, dtype=np.float64).reshape(-1, 18)
    out = np.zeros(t.shape[0], dtype=np.uint8)
    rc = load_library().cd_tri_contact_points(_ptr(t), t.shape[0], _ptr(out))
    if rc != CD_OK:
        raise CdError("cd_tri_contact_points", rc)
    return out


def version() -> str:
    return load_library().cd_version().decode()


def multi_unique_id() -> bytes:
    """ncclGetUniqueId through the library (128 bytes): made on one rank, handed to all."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.cd_multi_unique_id(buf)
    if rc != CD_OK:
        raise CdError("cd_multi_unique_id", rc)
    return buf.raw


class MultiStep:
    """The multi-GPU step of libmi355cd.so (cd_multi_*): RCCL collectives issued by the C++ side, one process per GPU."""

    def __init__(self, cd: "CollisionDetector", unique_id: bytes, rank: int, world: int, query_cap_per_peer: int = 0, flags: int = 0,
                 nccl_comm: int | None = None):
        """unique_id / rank / world: the library creates (and owns) the communicator.  nccl_comm: an existing ncclComm_t (as an
        integer address) whose rank and size are used instead; it stays the caller's."""
        self.lib = cd.lib
        self.cd = cd
        self._m = C.c_void_p()
        if nccl_comm is not None:
            rc = self.lib.cd_multi_create_from_comm(C.byref(self._m), cd._ctx, C.c_void_p(nccl_comm), query_cap_per_peer, flags)
            name = "cd_multi_create_from_comm"
        else:
            rc = self.lib.cd_multi_create(C.byref(self._m), cd._ctx, C.c_char_p(unique_id), rank, world, query_cap_per_peer, flags)
            name = "cd_multi_create"
        if rc != CD_OK:
            raise CdError(name, rc)
        self._pairs = None

    def step(self, cap: int = 1 << 22):
        if self._pairs is None or self._pairs.shape[0] < cap:
            self._pairs = np.empty((cap, 2), dtype=np.uint32)
        n = C.c_uint64(0)
        info = CdMultiInfo()
        rc = self.lib.cd_multi_step(self._m, self._pairs.ctypes.data, cap, C.byref(n), C.byref(info))
        if rc < 0:
            raise CdError("cd_multi_step", rc)
        return self._pairs[: min(n.value, cap)], int(n.value), rc, info

    def set_flags(self, flags: int):
        rc = self.lib.cd_multi_set_flags(self._m, flags)
        if rc != CD_OK:
            raise CdError("cd_multi_set_flags", rc)

    def close(self):
        if self._m:
            self.lib.cd_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
