"""Seeded synthetic inputs for the collision-detection and ray-tracing hot paths.

The reference ships no usable data set (its OBJ is missing from the tree), so the
workloads of BASELINE.json are generated here.  Every generator rounds vertex coordinates to float32
first and then widens to float64, mimicking the reference loader which parses `%f` into `float` and
stores `double` (load_obj.h:38,50-52).  Geometry lives inside the reference's hard-coded Morton frame
(morton.h:43-58) so CD_FRAME_REFERENCE keys are meaningful.

All generators are pure numpy and deterministic in (n, seed).
"""
from __future__ import annotations

import numpy as np

# the reference's hard-coded data-set bounds, morton.h:45,51,57
REF_OFF = np.array([0.004501, -0.476622, -0.381965], dtype=np.float64)
REF_SPAN = np.array([3.08, 0.76, 2.36], dtype=np.float64)
# generation box used by SURVEY.md 8(d): strictly inside the frame above
BOX_LO = np.array([0.05, -0.45, -0.35], dtype=np.float64)
BOX_HI = np.array([2.95, 0.25, 1.85], dtype=np.float64)


def _f32(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a.astype(np.float32).astype(np.float64))


def soup(n: int, e: float = 0.02, seed: int = 1234):
    """Random-triangle soup (BASELINE config 2 at n=100k, e=0.02; 1 M at e=0.01).

    Centroid uniform in the generation box, three private vertices = centroid + U(-e/2, e/2)^3,
    V = 3N (no shared vertices), ID = i.  Returns (verts[V,3] f64, vidx[N,3] u32).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    c = BOX_LO + (BOX_HI - BOX_LO) * rng.random((n, 3))
    d = (rng.random((n, 3, 3)) - 0.5) * e
    verts = _f32((c[:, None, :] + d).reshape(3 * n, 3))
    vidx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return verts, vidx


class MT19937_64:
    """std::mt19937_64 (the 64-bit Mersenne twister of Matsumoto & Nishimura, 2000; C++11 [rand.predef]) in numpy, a state block of 312 words at a
    time -- what SURVEY.md 8(d) names for the soups (`mt19937_64(seed=1234)`); numpy's own MT19937 is the 32-bit generator.  Known answer ([rand.predef]):
    the 10000th output of a default-seeded (5489) engine is 9981545732273789042 (tests/test_synth.py)."""
    NN, MM = 312, 156
    MATRIX, UM, LM = np.uint64(0xB5026F5AA96619E9), np.uint64(0xFFFFFFFF80000000), np.uint64(0x7FFFFFFF)

    def __init__(self, seed: int = 5489):
        mt = np.zeros(self.NN, dtype=np.uint64)
        x = seed & 0xFFFFFFFFFFFFFFFF
        for i in range(self.NN):
            if i:
                x = (6364136223846793005 * (x ^ (x >> 62)) + i) & 0xFFFFFFFFFFFFFFFF
            mt[i] = x
        self.mt = mt

    def _twist(self):
        mt, N, M = self.mt, self.NN, self.MM
        one = np.uint64(1)

        def mix(up, low, far):                                  # the recurrence for a run of positions whose three inputs are all available
            x = (up & self.UM) | (low & self.LM)
            return far ^ (x >> one) ^ np.where((x & one).astype(bool), self.MATRIX, np.uint64(0))
        mt[:N - M] = mix(mt[:N - M], mt[1:N - M + 1], mt[M:])                       # i in [0, 156): mt[i + 156] is still the old block's
        mt[N - M:N - 1] = mix(mt[N - M:N - 1], mt[N - M + 1:], mt[:M - 1])          # i in [156, 311): mt[i - 156] is the new block's
        mt[N - 1:] = mix(mt[N - 1:], mt[:1], mt[M - 1:M])                          # i = 311 wraps to the new mt[0]

    def raw(self, count: int) -> np.ndarray:
        """The next `count` 64-bit outputs."""
        blocks = (count + self.NN - 1) // self.NN
        out = np.empty(blocks * self.NN, dtype=np.uint64)
        for b in range(blocks):
            self._twist()
            out[b * self.NN:(b + 1) * self.NN] = self.mt
        x = out
        x ^= (x >> np.uint64(29)) & np.uint64(0x5555555555555555)
        x ^= (x << np.uint64(17)) & np.uint64(0x71D67FFFEDA60000)
        x ^= (x << np.uint64(37)) & np.uint64(0xFFF7EEE000000000)
        x ^= x >> np.uint64(43)
        # (a whole number of blocks is consumed: generators below draw everything they need in one call)
        return x[:count]

    def uniform(self, count: int, a: float, b: float) -> np.ndarray:
        """std::uniform_real_distribution<double>(a, b) as libstdc++ evaluates it on a 64-bit engine: generate_canonical<double, 53> is ONE draw,
        double(x) / 2^64 (the conversion rounds to nearest; a result of 1.0 is replaced by the double below it), then r * (b - a) + a."""
        r = self.raw(count).astype(np.float64) / 18446744073709551616.0
        r = np.where(r >= 1.0, np.nextafter(1.0, 0.0), r)
        return r * (b - a) + a


def soup_mt64(n: int, e: float = 0.02, seed: int = 1234):
    """SURVEY.md 8(d)'s soup recipe with the generator it names: std::mt19937_64(seed); ALL centroids first (x, y, z per triangle, each uniform in its
    interval of the generation box), then per triangle its three vertices' offsets (vertex-major, x y z), each uniform_real_distribution(-e/2, e/2);
    vertices rounded to float, V = 3N, ID = i.  (Round 5: the survey's recorded counts -- 1 326 pairs / 117 850 tested at 100 k, 16 795 / 1 222 266 at 1 M --
    are not reproduced by any draw order tried: the survey kept the description, not the code.  This order gives 1 326 / 117 666 and 16 992 / 1 224 320;
    DESIGN.md 3.)  Returns (verts[V,3] f64, vidx[N,3] u32)."""
    g = MT19937_64(seed)
    draws = g.raw(12 * n).astype(np.float64) / 18446744073709551616.0      # one engine, one stream: 3 n centroid draws, then 9 n offsets
    draws = np.where(draws >= 1.0, np.nextafter(1.0, 0.0), draws)
    c = draws[:3 * n].reshape(n, 3) * (BOX_HI - BOX_LO) + BOX_LO
    d = draws[3 * n:].reshape(n, 3, 3) * (e / 2 - (-e / 2)) + (-e / 2)
    verts = _f32((c[:, None, :] + d).reshape(3 * n, 3))
    vidx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return verts, vidx


def _sheet(nx: int, ny: int, zfun, x0, x1, y0, y1):
    """(nx x ny quads) -> (nx+1)*(ny+1) shared vertices, 2*nx*ny triangles."""
    xs = np.linspace(x0, x1, nx + 1)
    ys = np.linspace(y0, y1, ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    Z = zfun(X, Y)
    verts = np.stack([X, Y, Z], axis=-1).reshape(-1, 3)
    i = np.arange(nx)[:, None]
    j = np.arange(ny)[None, :]
    v00 = (i * (ny + 1) + j).ravel()
    v10 = v00 + (ny + 1)
    v01 = v00 + 1
    v11 = v10 + 1
    t1 = np.stack([v00, v10, v11], axis=-1)
    t2 = np.stack([v00, v11, v01], axis=-1)
    tris = np.empty((2 * nx * ny, 3), dtype=np.uint32)
    tris[0::2] = t1
    tris[1::2] = t2
    return verts, tris


def cloth_pair(quads: int = 500, x_offset: float = 0.0, round_f32: bool = True):
    """Cloth-vs-cloth (BASELINE config 3): two sheets of quads x quads quads, 2 triangles per quad.

    quads=500 -> 2 x 500 000 = 1 000 000 triangles, 2 x 501^2 vertices shared inside a sheet (so
    Triangle::neighborCount matters).  round_f32=False keeps the vertices as full doubles (vec3f.cuh:14-23 stores
    FP64; only the loader rounds, load_obj.h:38) -- the "double-coords" variant of the bench.  In the reference's y-up frame the sheets span x and z and
    undulate in y:  sheet A  y = -0.1 + 0.02 sin(6x) cos(6z);  sheet B the same surface phase-shifted
    and tilted so the two intersect along curves.  Returns (verts f64[V,3], vidx u32[N,3]).
    """
    x0, x1 = 0.06 + x_offset, 2.94 + x_offset
    z0, z1 = -0.34, 1.84

    def za(X, Z):
        return -0.1 + 0.02 * np.sin(6.0 * (X - x_offset)) * np.cos(6.0 * Z)

    def zb(X, Z):
        return -0.1 + 0.02 * np.sin(6.0 * (X - x_offset) + 0.9) * np.cos(6.0 * Z + 0.4) + 0.004 * ((X - x_offset) - 1.5)

    # build in (x, z) then place the height in y
    va, ta = _sheet(quads, quads, za, x0, x1, z0, z1)
    vb, tb = _sheet(quads, quads, zb, x0 + 0.0007, x1 + 0.0007, z0 + 0.0011, z1 + 0.0011)
    va = va[:, [0, 2, 1]]
    vb = vb[:, [0, 2, 1]]
    verts = np.concatenate([va, vb], axis=0)
    verts = _f32(verts) if round_f32 else np.ascontiguousarray(verts, dtype=np.float64)
    vidx = np.concatenate([ta, tb + np.uint32(va.shape[0])], axis=0).astype(np.uint32)
    return verts, np.ascontiguousarray(vidx)


def cloth_shard(rank: int, quads: int = 500, overlap: float = 0.10):
    """BASELINE config 4: rank r owns a copy of the config-3 geometry shifted along x so that
    neighbouring copies overlap by `overlap` of their width.  Vertex indices and triangle IDs are
    GLOBAL (rank * per-rank count offset) so neighborCount / the ID rule stay meaningful across ranks.
    Coordinates leave the reference's Morton frame for rank > 0, so use CD_FRAME_AUTO.
    Returns (verts, vidx_local, ids_global, vertex_offset)."""
    width = 2.88
    verts, vidx = cloth_pair(quads, x_offset=rank * width * (1.0 - overlap))
    n = vidx.shape[0]
    ids = (np.arange(n, dtype=np.uint64) + rank * n).astype(np.uint32)
    return verts, vidx, ids, rank * verts.shape[0]


def config4_merged(world: int = 8, quads: int = 500, overlap: float = 0.10):
    """BASELINE config 4's eight shards as ONE mesh (the single-GPU point a scaling curve is read against): the concatenation of
    cloth_shard(0..world-1) with global vertex indices and triangle IDs, and the Morton frame of the merged centroids
    (off = min, span = (max - min)(1 + 2^-20): what CD_FRAME_AUTO computes).  Returns (verts, vidx, ids, off, span)."""
    vs, ts, ids = [], [], []
    for r in range(world):
        v, t, i, vb = cloth_shard(r, quads, overlap)
        vs.append(v); ts.append(t + np.uint32(vb)); ids.append(i)
    verts = np.concatenate(vs); vidx = np.ascontiguousarray(np.concatenate(ts).astype(np.uint32)); ids = np.concatenate(ids).astype(np.uint32)
    lo = np.full(3, np.inf); hi = np.full(3, -np.inf)
    for a in range(0, vidx.shape[0], 1 << 20):                       # centroids as the library forms them, (p1 + p2 + p3) / 3, in slabs
        t = vidx[a:a + (1 << 20)]
        cen = (verts[t[:, 0]] + verts[t[:, 1]] + verts[t[:, 2]]) / 3
        lo = np.minimum(lo, cen.min(0)); hi = np.maximum(hi, cen.max(0))
    return verts, vidx, ids, lo, (hi - lo) * (1.0 + 2.0 ** -20)


def grids_obj_text(m: int = 32) -> str:
    """Plumbing input (BASELINE config 1): two interpenetrating m x m grids (~4 k triangles at m=32)
    written in the only OBJ dialect the reference loader accepts: `v x y z` and `f a/ta b/tb c/tc`
    with 1-based indices (load_obj.h:50,68)."""
    def za(X, Y):
        return 0.5 + 0.05 * np.sin(5.0 * X) * np.cos(7.0 * Y)

    def zb(X, Y):
        return 0.5 + 0.05 * np.cos(6.0 * X + 0.3) * np.sin(5.0 * Y) + 0.01

    va, ta = _sheet(m, m, za, 0.1, 2.9, -0.4, 0.2)
    vb, tb = _sheet(m, m, zb, 0.1003, 2.9003, -0.3996, 0.2004)
    verts = np.concatenate([va, vb], axis=0)
    tris = np.concatenate([ta, tb + np.uint32(va.shape[0])], axis=0)
    lines = ["# generated by mi355_synth.grids_obj_text"]
    for v in verts:
        lines.append("v %.6f %.6f %.6f" % (v[0], v[1], v[2]))
    for t in tris:
        lines.append("f %d/%d %d/%d %d/%d" % (t[0] + 1, t[0] + 1, t[1] + 1, t[1] + 1, t[2] + 1, t[2] + 1))
    return "\n".join(lines) + "\n"


def parse_obj_text(text: str):
    """Python restatement of the loader's parse step for tests (load_obj.h:41-103): float vertices,
    `f a/ta b/tb c/tc` faces, 1-based -> 0-based."""
    vs, fs = [], []
    for line in text.splitlines():
        if line.startswith("v "):
            p = line.split()
            vs.append([np.float32(p[1]), np.float32(p[2]), np.float32(p[3])])
        elif line.startswith("f "):
            p = line.split()[1:4]
            fs.append([int(q.split("/")[0]) - 1 for q in p])
    return (np.asarray(vs, dtype=np.float32).astype(np.float64).reshape(-1, 3),
            np.asarray(fs, dtype=np.uint32).reshape(-1, 3))


# ---------------------------------------------------------------- ray tracer
SPHERE_DTYPE = np.dtype([("r", "<f4"), ("b", "<f4"), ("g", "<f4"), ("radius", "<f4"),
                         ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("idx", "<i4")])   # sphere.cuh:28-32
assert SPHERE_DTYPE.itemsize == 32


def sphere_scene(n_spheres: int = 4096, dim: int = 4096, seed: int = 7):
    """BASELINE config 5: colour U[0,1)^3, x,y,z U[-dim/2, dim/2), radius U[8,28), idx = i
    (the reference's ranges, anime_ray.cu:168-175, scaled from its 1024-pixel image to `dim`).
    Returns (spheres[SPHERE_DTYPE], shifts int32[n,4]) with shifts as initSpheres leaves them
    (sphere.cuh:54-56): {0, 0, (i%5+1)*5, (i%2)*2-1}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    s = np.zeros(n_spheres, dtype=SPHERE_DTYPE)
    s["r"] = rng.random(n_spheres, dtype=np.float32)
    s["g"] = rng.random(n_spheres, dtype=np.float32)
    s["b"] = rng.random(n_spheres, dtype=np.float32)
    half = dim / 2.0
    s["x"] = (rng.random(n_spheres) * dim - half).astype(np.float32)
    s["y"] = (rng.random(n_spheres) * dim - half).astype(np.float32)
    s["z"] = (rng.random(n_spheres) * dim - half).astype(np.float32)
    s["radius"] = (rng.random(n_spheres) * 20.0 + 8.0).astype(np.float32)
    s["idx"] = np.arange(n_spheres, dtype=np.int32)
    shifts = np.zeros((n_spheres, 4), dtype=np.int32)
    i = np.arange(n_spheres)
    shifts[:, 2] = (i % 5 + 1) * 5
    shifts[:, 3] = (i % 2) * 2 - 1
    return s, shifts
