"""Input recipes of tests/golden/morton_ref.npz, shared by its generator (tests/golden/make_morton_ref.py, which feeds
them to the REFERENCE's morton.h compiled unmodified: oracle/_ref/libref_morton.so) and by the tests that replay them
through the oracle and the HIP path.

Every recipe is integer arithmetic plus IEEE operations that are exact or correctly rounded (u64 LCG, int -> double,
one division / multiplication by a power of two, float32 rounding), so the same bytes come out on any box; the fixture
also holds a SHA-256 of every input array and the tests check it before they trust the expected outputs.
"""
from __future__ import annotations

import hashlib

import numpy as np

REF_OFF = np.array([0.004501, -0.476622, -0.381965], dtype=np.float64)     # morton.h:45,51,57
REF_SPAN = np.array([3.08, 0.76, 2.36], dtype=np.float64)

N_EXPAND = 1 << 17          # >= 10^5 expand64Bits inputs
N_POINTS = 1 << 20          # >= 10^6 morton3D points
N_FULL = 1 << 16            # points whose keys are stored in full (the rest: digest + every 64th key)
SAMPLE_STRIDE = 64


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def lcg(n: int, seed: int) -> np.ndarray:
    """Knuth's MMIX LCG on uint64 (wraps mod 2^64 in numpy), one value per step, as a closed loop-free form is not needed:
    n is at most a few million."""
    a = np.uint64(6364136223846793005); c = np.uint64(1442695040888963407)
    out = np.empty(n, dtype=np.uint64)
    # jump-free vectorised form: x_k = a^k x_0 + c (a^k - 1)/(a - 1); computed by doubling blocks
    out[0] = np.uint64(seed)
    filled = 1
    mul, add = a, c                       # the affine map of ONE step
    with np.errstate(over="ignore"):
        while filled < n:
            m = min(filled, n - filled)
            out[filled:filled + m] = out[:m] * mul + add      # out[k + filled] = step^filled(out[k])
            filled += m
            add = add * mul + add                            # compose the map with itself: x -> mul (mul x + add) + add
            mul = mul * mul
    return out


def expand_inputs() -> np.ndarray:
    """All 2^16 low patterns shifted through the 21 live bits, every single bit, all-ones masks, and random 64-bit
    words (bits above 20 must be ignored, morton.h:15)."""
    r = lcg(N_EXPAND, 12345)
    v = r.copy()
    k = np.arange(1 << 16, dtype=np.uint64)
    v[:1 << 16] = (k << np.uint64(5)) | (k >> np.uint64(11))          # every 16-bit pattern across bits 5..20 / 0..4
    v[1 << 16:(1 << 16) + 64] = np.uint64(1) << np.arange(64, dtype=np.uint64)
    v[(1 << 16) + 64:(1 << 16) + 128] = (np.uint64(1) << np.arange(64, dtype=np.uint64)) - np.uint64(1)
    v[(1 << 16) + 128] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return v


def frame_points() -> np.ndarray:
    """N_POINTS points strictly inside the reference's frame (so morton.h:78's assert holds):
      * the first quarter: float32-valued coordinates (what load_obj.h:38 produces), uniform over the frame;
      * the second quarter: full-precision doubles;
      * the third quarter: CELL BOUNDARIES -- for random cells k the double nearest to off + span * k / 2^20 and its
        two neighbours (nextafter either way), per axis: where ((c - off) / span) * 2^20 truncates is decided by the
        last bit of the division and of the product;
      * the last quarter: points hugging the frame's low and high faces (norm just above 0, just below 1)."""
    n = N_POINTS
    r = lcg(3 * n, 987654321).reshape(n, 3)
    u = (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))          # [0, 1), exact
    lo = REF_OFF + REF_SPAN * (1.0 / 1024)
    p = lo + u * (REF_SPAN * (1022.0 / 1024))
    q = n // 4
    p[:q] = p[:q].astype(np.float32).astype(np.float64)
    cell = ((r[2 * q:3 * q] >> np.uint64(40)) % np.uint64((1 << 20) - 2)).astype(np.float64) + 1.0
    edge = REF_OFF + REF_SPAN * (cell / 1048576.0)
    which = ((r[2 * q:3 * q] >> np.uint64(8)) % np.uint64(3)).astype(np.int64)
    edge = np.where(which == 0, np.nextafter(edge, -np.inf), np.where(which == 2, np.nextafter(edge, np.inf), edge))
    p[2 * q:3 * q] = edge
    tiny = ((r[3 * q:] >> np.uint64(44)).astype(np.float64) + 1.0) * (1.0 / (1 << 40))      # (0, 2^-20]
    side = ((r[3 * q:] >> np.uint64(9)) & np.uint64(1)).astype(bool)
    p[3 * q:] = np.where(side, REF_OFF + REF_SPAN * (1.0 - tiny) * (1.0 - 1.0 / (1 << 30)), REF_OFF + REF_SPAN * tiny)
    return np.ascontiguousarray(p)


def cloth_centroids(quads: int = 500):
    """Centroids of BASELINE config 3 (mi355_synth.cloth_pair(500)) in the reference's operand order,
    load_obj.h:89-101: (p1 + p2 + p3) / 3 per axis.  Returns (centroids f64[N,3], verts, vidx)."""
    import mi355_synth as synth
    verts, vidx = synth.cloth_pair(quads)
    p1, p2, p3 = verts[vidx[:, 0]], verts[vidx[:, 1]], verts[vidx[:, 2]]
    return np.ascontiguousarray((p1 + p2 + p3) / 3), verts, vidx
