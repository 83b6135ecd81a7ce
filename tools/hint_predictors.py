#!/usr/bin/env python3
"""Could a hint serve a mesh that MOVES?  The 1 M cloth pair, sheet B translated by SHIFT quads per frame.  For every frame f + 1 the descent is run (a) in the plain order, (b) in an order made
from frame f + 1's OWN wave times (measured in a first run of the same frame: what a perfect predictor would give), (c) from frame f's times carried over BY POSITION (what the library's hint
would do if cd_update_vertices did not drop it), (d) carried over BY TRIANGLE (every triangle remembers the class of the wave it was in; a new group takes the mean / the max of its members), (e) with the
library's own hint (by triangle, max: CD_OPT_ORDER_HINT).  Orders are installed with cd_debug_hint_set into a tree built by cd_build_tree (longest first per XCD list,
stable), cd_find_collisions traverses it, the descent's own clock is read.  usage: hint_predictors.py [FRAMES] [SHIFT ...]   GPU only."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import numpy as np
import mi355_synth as synth, mi355cd
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shifts = [float(x) for x in sys.argv[2:]] or [0.25, 1.0, 4.0]
verts, vidx = synth.cloth_pair(500)
quad = 2.9 / 500; h = verts.shape[0] // 2
buf = np.empty((1 << 22, 2), dtype=np.uint32)

def order_from(score, plain):
    """per XCD list (workgroups b = x mod 8), the groups of the plain order sorted by score, longest first, stable"""
    out = plain.copy()
    for x in range(8):
        lst = plain[x::8]
        out[x::8] = lst[np.argsort(-score[lst], kind="stable")]
    return out

with mi355cd.CollisionDetector(verts, vidx) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    n = vidx.shape[0]; g = (n + 63) // 64
    def upload(f, shift):
        v = verts.copy(); v[h:, 0] = np.float32(v[h:, 0] + np.float32(f * shift * quad)); cd.update_vertices(v)
    def times_by_group():
        _, _, tri = cd.debug_hint(with_order=False, with_tri=True); _, perm = cd.export_keys()
        return tri[perm[::64]].astype(np.float64), tri, perm
    def run_plain():
        cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 0); cd.build_tree(); cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 1)   # (a tree without a hint; the traversal still leaves its times)
        cd.find_collisions(cap=1 << 22); return cd.stats().ms_descend_clock * 1e3
    def run_with(order):
        cd.build_tree(); cd.debug_hint_set(order); cd.find_collisions(cap=1 << 22); return cd.stats().ms_descend_clock * 1e3
    def run_library():
        cd.build_tree(); cd.find_collisions(cap=1 << 22); return cd.stats().ms_descend_clock * 1e3                  # the library's own hint (by triangle, max)
    def half_vblock(b, nb):
        per = nb >> 3; v = (b & 7) * per + (b >> 3) if b < (per << 3) else b
        c = per // 4
        if c > 0 and b < c * 4 * 8:
            x, l = b & 7, b >> 3; sub, off = l // c, l % c
            v = (sub * 8 + x) * c + off
        elif c > 0: v = b
        return v
    plain = np.array([half_vblock(b, g) for b in range(g)], dtype=np.uint32)
    for shift in shifts:
        res = {k: [] for k in ("plain", "own times", "by position", "by triangle (mean)", "by triangle (max)", "the library's hint")}
        upload(0, shift); run_plain(); run_plain()
        t_prev, tri_prev, perm_prev = times_by_group()
        for f in range(1, frames + 1):
            upload(f, shift)
            res["plain"].append(run_plain())
            t_now, tri_now, perm_now = times_by_group()                 # this frame's times in the plain order
            m = np.concatenate([tri_prev[perm_now].astype(np.float64), np.zeros(g * 64 - n)]).reshape(g, 64)
            for name, score in (("own times", t_now), ("by position", t_prev), ("by triangle (mean)", m.mean(1)), ("by triangle (max)", m.max(1))):
                res[name].append(run_with(order_from(score, plain)))
            # the library's own hint needs the PREVIOUS frame's times with the triangles: put them back, then a fused build + traversal
            upload(f - 1, shift); run_plain(); upload(f, shift)
            res["the library's hint"].append(run_library())
            run_plain(); t_prev, tri_prev, perm_prev = times_by_group()
        print(f"sheet B moves {shift:5.2f} quads per frame, descent (device clock, median of {frames} frames): " + "  ".join(f"{k} {statistics.median(v):.1f}" for k, v in res.items()), flush=True)
