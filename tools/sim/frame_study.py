"""CPU study of Morton frames for meshes the reference's constants (morton.h:43-58) do not fit.

For a mesh: the keys of several frames -> the oracle's tree -> node visits per query (tree quality), and what the hybrid
sort needs of the same keys: the longest run of equal key bits 44..59 (k_local_sort's windows: <= 3072 small form, <= 6144
large form) and the longest run of equal high halves (the fix-up hop: <= 16).  The pair set and pairs tested must not move.

  per-axis   span[a] = max - min per axis, fixed x y z interleave: round 5's CD_FRAME_AUTO
  isotropic  span = the largest extent on all axes, fixed interleave
  cap k      the adaptive layout (oracle/cd_oracle.c, "the ADAPTIVE frame"): the 60 key bits dealt to the axes by extent in units
             of the triangles' own extent, an axis counting at most 2^k longer for its thin triangles.  cap 0: cubes by extent
             alone; cap 3: this round's CD_FRAME_AUTO

usage: python3 tools/sim/frame_study.py [mesh ...]     meshes: c4_320k c4_2M cloth250k soup100k soup300k ellipsoids sheets (default: all but c4_2M)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth  # noqa: E402
import oracle  # noqa: E402


def runs(keys_sorted, shift):
    t = keys_sorted >> np.uint64(shift)
    edge = np.flatnonzero(np.concatenate(([True], t[1:] != t[:-1], [True])))
    return int(np.diff(edge).max())


BASE = {}


def study(name, verts, vidx, ids, keys0):
    t0 = time.time()
    keys, perm = oracle.sort_by_key(keys0)
    left, right, parent, rf, rl, wrong = oracle.build_hierarchy(keys, 1)
    boxes, bounded, cc = oracle.refit(verts, vidx, perm, left, right, parent)
    pairs, st = oracle.find_collisions(verts, vidx, perm, left, right, boxes, ids)
    n = vidx.shape[0]
    same = BASE.setdefault("res", (st.n_pairs, st.pairs_tested)) == (st.n_pairs, st.pairs_tested)
    print(f"  {name:34s} visits/query {st.node_visits / n:6.2f}  max stack {st.max_stack:2d}"
          f"  | run(bits 44..59) {runs(keys, 44):6d}  run(48..63) {runs(keys, 48):6d}  run(high half) {runs(keys, 32):4d}  top bit {int(keys.max()).bit_length()}"
          f"  {'' if same else '  !!! RESULT DIFFERS'}  ({time.time() - t0:.0f} s)", flush=True)
    return st


def ellipsoid(nu, nv, centre, radii):
    u = np.linspace(0, np.pi, nu + 1)[:, None]; v = np.linspace(0, 2 * np.pi, nv + 1)[None, :]
    P = np.stack([radii[0] * np.sin(u) * np.cos(v), radii[1] * np.cos(u) * np.ones_like(v), radii[2] * np.sin(u) * np.sin(v)], -1) + np.asarray(centre)
    verts = P.reshape(-1, 3)
    i = np.arange(nu)[:, None]; j = np.arange(nv)[None, :]
    v00 = (i * (nv + 1) + j).ravel(); v10 = v00 + nv + 1; v01 = v00 + 1; v11 = v10 + 1
    tris = np.concatenate([np.stack([v00, v10, v11], -1), np.stack([v00, v11, v01], -1)]).astype(np.uint32)
    return verts, tris


def meshes(which):
    if "c4_320k" in which:
        v, t, i, lo, sp = synth.config4_merged(8, 100); yield "config 4 merged, 320 k", v, t, i
    if "c4_2M" in which:
        v, t, i, lo, sp = synth.config4_merged(8, 250); yield "config 4 merged, 2 M", v, t, i
    if "cloth250k" in which:
        v, t = synth.cloth_pair(250); yield "cloth pair, 250 k (reference frame fits)", v, t, None
    if "soup100k" in which:
        v, t = synth.soup(100000, e=0.02, seed=1234); yield "soup 100 k", v, t, None
    if "soup300k" in which:
        v, t = synth.soup(300000, e=0.014, seed=5); yield "soup 300 k", v, t, None
    if "ellipsoids" in which:
        a, ta = ellipsoid(300, 300, (0, 0, 0), (3.0, 1.0, 2.0)); b, tb = ellipsoid(300, 300, (0.8, 0.3, 0.2), (3.0, 1.0, 2.0))
        v = np.concatenate([a, b]).astype(np.float32).astype(np.float64); t = np.concatenate([ta, tb + np.uint32(a.shape[0])]).astype(np.uint32)
        yield "two ellipsoids 3:1:2, 360 k", v, np.ascontiguousarray(t), None
    if "sheets" in which:                                   # flat horizontal sheets on top of each other: every box flat along y
        vs, ts, off = [], [], 0
        for k in range(12):
            v, t = synth._sheet(120, 120, lambda X, Y: 0.0 * X + 0.01 * k, 0.0 + 0.003 * k, 4.0 + 0.003 * k, 0.0, 1.0)
            vs.append(v[:, [0, 2, 1]]); ts.append(t + np.uint32(off)); off += v.shape[0]
        v = np.concatenate(vs).astype(np.float32).astype(np.float64); yield "12 flat sheets, 345 k", v, np.ascontiguousarray(np.concatenate(ts).astype(np.uint32)), None


def main():
    which = sys.argv[1:] or ["c4_320k", "cloth250k", "soup100k", "soup300k", "ellipsoids", "sheets"]
    for name, verts, vidx, ids in meshes(which):
        BASE.clear()
        lo, hi, ss, sc = oracle.layout_stat(verts, vidx)
        span = (hi - lo) * (1.0 + 2.0 ** -20); span[span <= 0] = 1.0
        mean_ext = np.where(sc > 0, 2.0 ** (ss / np.maximum(sc, 1) / 256.0), 0.0)
        print(f"{name}: {vidx.shape[0]} triangles, centroid extents {hi - lo}, mean box extents {mean_ext}", flush=True)
        if "reference frame fits" in name:
            study("reference frame (morton.h:43-58)", verts, vidx, ids, oracle.centroid_morton(verts, vidx))
        study("per-axis (r05 AUTO)", verts, vidx, ids, oracle.centroid_morton(verts, vidx, lo, span))
        study("isotropic", verts, vidx, ids, oracle.centroid_morton(verts, vidx, lo, np.full(3, span.max())))
        for cap in (0, 2, 3, 4, 8):
            lay = oracle.frame_layout_cap(lo, hi, ss, sc, cap)
            study(f"cap {cap}: {oracle.layout_fields(lay)}", verts, vidx, ids, oracle.centroid_morton_layout(verts, vidx, lo, span, lay))


if __name__ == "__main__":
    main()
