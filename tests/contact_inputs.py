"""Input recipes of tests/golden/contact_ref.npz, shared by its generator (tests/golden/make_contact_ref.py, which feeds
them to the REFERENCE's tri_contact.cuh / box.cuh / triangle.cuh / vec3f.cuh compiled unmodified:
oracle/_ref/libref_contact.so) and by the tests that replay them through the oracle and the HIP path.

As in morton_inputs.py every recipe is a u64 LCG plus IEEE operations that are exact or correctly rounded, so the same
bytes come out on any box; the fixture holds a SHA-256 of every input array and the tests check it first.

Triangle-pair families (tri_pairs() -> f64[n, 6, 3] = P1 P2 P3 Q1 Q2 Q3 and a family id per pair):
  0 float     float32-valued vertices (what load_obj.h:38 produces), the two triangles near each other
  1 double    full-precision doubles, same geometry
  2 coplanar  both triangles in one axis-aligned plane (exactly coplanar: n1 . q == 0 decides, tri_contact.cuh:58-59)
  3 touching  Q1 is a vertex of P / the midpoint of an edge of P / a point inside P, on a dyadic lattice (exact)
  4 samepos   the triangles share one or two vertex POSITIONS (what cloth neighbours look like to the SAT once
              neighborCount let them through; degenerate edge x edge axes)
  5 degen     zero-area triangles: two equal vertices, three collinear vertices, a point
  6 lattice   all 18 coordinates small integers in [-4, 4]: ties on every axis
"""
from __future__ import annotations

import numpy as np

from morton_inputs import lcg, sha  # noqa: F401  (sha re-exported for the generator and the tests)

FAMILIES = ("float", "double", "coplanar", "touching", "samepos", "degen", "lattice")
FAMILY_SIZE = (1 << 18, 1 << 18, 1 << 17, 1 << 17, 1 << 17, 1 << 17, 1 << 17)      # 1 179 648 pairs
N_BOX = 1 << 20
N_INDEXED = 1 << 18
N_SMALL = 1 << 16
SAMPLE_STRIDE = 64


def unit(n: int, seed: int) -> np.ndarray:
    """n doubles in [0, 1) with 53 random bits each (exact)."""
    return (lcg(n, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def ints(n: int, seed: int, lo: int, hi: int) -> np.ndarray:
    """n integers in [lo, hi] from the high bits of the LCG."""
    return ((lcg(n, seed) >> np.uint64(33)) % np.uint64(hi - lo + 1)).astype(np.int64) + lo


def _near_pairs(n: int, seed: int) -> np.ndarray:
    """Two random triangles whose vertices lie in overlapping unit-ish boxes: roughly half of them intersect."""
    u = unit(n * 18, seed).reshape(n, 6, 3)
    centre = unit(n * 3, seed + 1).reshape(n, 1, 3) * 2.0 - 1.0
    shift = (unit(n * 3, seed + 2).reshape(n, 1, 3) - 0.5) * 0.8
    t = centre + (u - 0.5)
    t[:, 3:] += shift
    return t


def tri_pairs():
    fam, out = [], []
    for f, n in enumerate(FAMILY_SIZE):
        seed = 1000 + 17 * f
        if f == 0:
            t = _near_pairs(n, seed).astype(np.float32).astype(np.float64)
        elif f == 1:
            t = _near_pairs(n, seed)
        elif f == 2:
            t = _near_pairs(n, seed).astype(np.float32).astype(np.float64)
            axis = ints(n, seed + 3, 0, 2)
            plane = t[np.arange(n), 0, axis].copy()
            for a in range(3):
                m = axis == a
                t[m, :, a] = plane[m, None]
        elif f == 3:
            g = ints(n * 18, seed, -16, 16).reshape(n, 6, 3).astype(np.float64) * 0.125          # dyadic lattice
            kind = ints(n, seed + 3, 0, 3)
            w = ints(n * 2, seed + 4, 1, 3).reshape(n, 2).astype(np.float64)
            q1 = np.where((kind == 0)[:, None], g[:, 0],
                          np.where((kind == 1)[:, None], (g[:, 0] + g[:, 1]) * 0.5,
                                   np.where((kind == 2)[:, None], (g[:, 1] + g[:, 2]) * 0.5,
                                            (g[:, 0] * (8.0 - w[:, :1] - w[:, 1:]) + g[:, 1] * w[:, :1] + g[:, 2] * w[:, 1:]) * 0.125)))
            g[:, 3] = q1
            t = g
        elif f == 4:
            t = _near_pairs(n, seed).astype(np.float32).astype(np.float64)
            kind = ints(n, seed + 3, 0, 3)
            t[:, 3] = t[:, 0]                                              # Q1 == P1
            m = kind >= 1; t[m, 4] = t[m, 1]                               # and Q2 == P2 (shared edge)
            m = kind == 2; t[m, 4] = t[m, 2]                               # or Q2 == P3
            m = kind == 3; t[m, 5] = t[m, 2]                               # all three (the same triangle, other indices)
        elif f == 5:
            t = _near_pairs(n, seed).astype(np.float32).astype(np.float64)
            kind = ints(n, seed + 3, 0, 4)
            which = ints(n, seed + 4, 0, 1) * 3                            # degenerate P or Q
            r = np.arange(n)
            m = kind == 0; t[r[m], which[m] + 1] = t[r[m], which[m]]       # two equal vertices
            m = kind == 1; t[r[m], which[m] + 2] = (t[r[m], which[m]] + t[r[m], which[m] + 1]) * 0.5   # collinear (midpoint)
            m = kind == 2; t[r[m], which[m] + 1] = t[r[m], which[m]]; t[r[m], which[m] + 2] = t[r[m], which[m]]   # a point
            m = kind == 3; t[m, 1] = t[m, 0]; t[m, 4] = t[m, 3]            # both degenerate
            m = kind == 4; t[m, 3:] = t[m, :3]                             # identical triangles
        else:
            t = ints(n * 18, seed, -4, 4).reshape(n, 6, 3).astype(np.float64)
        out.append(np.ascontiguousarray(t)); fam.append(np.full(n, f, dtype=np.uint8))
    return np.ascontiguousarray(np.concatenate(out)), np.concatenate(fam)


N_NONFINITE = 1 << 16
SPECIALS = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e300, -1e300, 1e-300, 1e155, -1e155, 1.7e308, 5e-324, 1.0, -1.0])


def nonfinite_pairs() -> np.ndarray:
    """f64[N_NONFINITE, 6, 3]: near pairs in which 0 .. 18 coordinates are replaced by NaN, +-inf, +-0, huge, tiny and ordinary values, an eighth of them scaled as a whole
    by 1e150 / 1e-150 / 1e200 (overflow to inf - inf and underflow INSIDE the cross and dot products).  What tri_contact.cuh does with them is decided by mathop.cuh's
    NaN-asymmetric compare-selects: round 6's fixture (tests/golden/contact_nonfinite_ref.npz, made by the reference compiled here) pins it for the device's
    hardware-max / min form and its fall-back (cd_math.h, tri_contact_fast)."""
    n = N_NONFINITE
    t = _near_pairs(n, 7001).reshape(n, 18)
    how_many = np.array([0, 1, 2, 4, 9, 18, 3, 6])[ints(n, 7002, 0, 7)]
    where = ints(n * 18, 7003, 0, 17).reshape(n, 18)                      # positions drawn with repetition: "up to" how_many distinct ones
    what = SPECIALS[ints(n * 18, 7004, 0, len(SPECIALS) - 1)].reshape(n, 18)
    rows = np.arange(n)
    for k in range(18):
        on = how_many > k
        t[rows[on], where[on, k]] = what[on, k]
    scale = np.array([1e150, 1e-150, 1e200, 1.0])[ints(n // 8, 7005, 0, 3)]
    with np.errstate(over="ignore", invalid="ignore"):
        t[: n // 8] *= scale[:, None]
    return np.ascontiguousarray(t.reshape(n, 6, 3))


def box_pairs():
    """N_BOX box pairs {x1,x2,y1,y2,z1,z2}: half on a small integer lattice (touching faces, zero-thickness boxes, identical
    boxes: where the strict '> 0' of box.cuh:41 decides), a quarter float32-valued, a quarter full doubles."""
    n = N_BOX
    h = n // 2
    q = n // 4
    lo = ints(h * 6, 31, -6, 6).reshape(h, 2, 3).astype(np.float64)
    ext = ints(h * 6, 32, 0, 4).reshape(h, 2, 3).astype(np.float64)
    lat = np.stack([lo, lo + ext], axis=-1)                                # [h, 2, 3, 2]
    c = unit(h * 6, 33).reshape(h, 2, 3) * 2.0 - 1.0
    e = unit(h * 6, 34).reshape(h, 2, 3) * 0.7
    rnd = np.stack([c - e, c + e], axis=-1)
    rnd[:q] = rnd[:q].astype(np.float32).astype(np.float64)
    b = np.concatenate([lat, rnd]).reshape(n, 2, 6)
    return np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1])


def indexed_pairs():
    """N_INDEXED indexed triangle pairs over one vertex pool, for checkTriangleContactHelper (ID rule + vertex fetch) and
    Triangle::neighborCount: indices drawn from a window of 12 so shared indices (0..3 of them, repeats inside one
    triangle too) are common; IDs equal, reversed and ordered."""
    n = N_INDEXED
    pool = 4096
    verts = (unit(pool * 3, 51).reshape(pool, 3) * 2.0 - 1.0).astype(np.float32).astype(np.float64)
    verts[: pool // 4] = ints(pool // 4 * 3, 52, -3, 3).reshape(-1, 3).astype(np.float64)
    base = ints(n, 53, 0, pool - 13)
    va = (base[:, None] + ints(n * 3, 54, 0, 11).reshape(n, 3)).astype(np.uint32)
    vb = (base[:, None] + ints(n * 3, 55, 0, 11).reshape(n, 3)).astype(np.uint32)
    ida = ints(n, 56, 0, 7).astype(np.uint32)
    idb = ints(n, 57, 0, 7).astype(np.uint32)
    return np.ascontiguousarray(verts), np.ascontiguousarray(va), ida, np.ascontiguousarray(vb), idb


def small_vectors():
    """Operands for project3 (4 vectors), project6 (7 vectors) and cross / dot (2 vectors): half lattice, half doubles."""
    n = N_SMALL

    def mix(k, seed):
        a = unit(n * k * 3, seed).reshape(n, k, 3) * 2.0 - 1.0
        a[: n // 2] = ints(n // 2 * k * 3, seed + 1, -3, 3).reshape(n // 2, k, 3).astype(np.float64)
        return np.ascontiguousarray(a)
    return mix(4, 71), mix(7, 73), mix(2, 75)


def end_configs():
    """The meshes whose END RESULT (pair set + pairs tested) the fixture pins: name -> (verts, vidx, mode), mode as
    ref_pair_set() takes it (0 = plain O(N^2), 1 = sweep proposals)."""
    import mi355_synth as synth
    yield "soup100k", synth.soup(100_000, e=0.02, seed=1234), 0            # BASELINE config 2
    yield "cloth1M", synth.cloth_pair(500), 1                              # BASELINE config 3
    yield "soup1M", synth.soup(1_000_000, e=0.01, seed=1234), 1            # the bench line's second workload
    yield "cloth1M_double", cloth_pair_double(500), 1                      # config 3 with vertices NOT rounded to float32
    # SURVEY.md 8(d)'s recipe with the generator it names, std::mt19937_64(seed = 1234) (mi355_synth.soup_mt64; round 5)
    yield "soup100k_mt64", synth.soup_mt64(100_000, e=0.02, seed=1234), 0  # BASELINE config 2, the survey's generator
    yield "soup1M_mt64", synth.soup_mt64(1_000_000, e=0.01, seed=1234), 1


def cloth_pair_double(quads: int = 500):
    """BASELINE config 3's surfaces with full-double vertices (vec3f.cuh:14-23 stores FP64; only the loader rounds)."""
    import mi355_synth as synth
    return synth.cloth_pair(quads, round_f32=False)


def pair_keys(pairs) -> np.ndarray:
    """Sorted u64 keys (a << 32 | b) of a pair list: the canonical form whose bytes are hashed."""
    p = np.asarray(pairs, dtype=np.uint64).reshape(-1, 2)
    return np.sort((p[:, 0] << np.uint64(32)) | p[:, 1])
