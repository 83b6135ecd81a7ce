"""Pins the CPU oracle (oracle/cd_oracle.c, oracle/rt_oracle.c) to the reference.

What the reference offers for this path and how each is used here:
  * check.cuh:19-27 testFunc key set {1,2,4,5,19,24,25,30}: the reference never records its output; the
    answers below were produced by the reference's own determineRange/findSplit source when the survey
    executed it, and are re-derived by hand in test_kat_by_hand.
  * morton.h:15-20 masks and the two morton3D anchors recorded in SURVEY.md (8a row a4, Appendix A).
  * Sphere::hit anchor recorded in SURVEY.md Appendix A.
  * resources/MyResult.txt and resources/flag-2000-changed.txt (copied verbatim as data fixtures): the
    input OBJ is missing from the reference tree, so they cannot be replayed; they pin the output
    convention (smaller ID first, sets agree between the GPU run and the independent CPU tool).
  * the reference's structural self-check expectations (resources/cleanResult.png).
  * an independent O(N^2) brute force (check.cuh:117-141 restated) must equal the tree traversal.
"""
import os
import re

import numpy as np
import pytest

import mi355_synth as synth
import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_expand64_kat():
    assert oracle.expand64(0x1fffff) == 0x1249249249249249
    assert oracle.expand64(0xfffff) == 0x249249249249249
    assert oracle.expand64(1) == 1
    assert oracle.expand64(0) == 0
    assert oracle.expand64(0xffffffffffe00000) == 0       # only the low 21 bits survive (morton.h:15)
    # every bit lands on a multiple of 3
    for b in range(21):
        assert oracle.expand64(1 << b) == 1 << (3 * b)


def test_morton3d_anchors():
    assert oracle.morton3d(1.0, 0.0, 0.5) == 384255804010903211
    cx, cy, cz = 0.004501 + 3.08 / 2, -0.476622 + 0.76 / 2, -0.381965 + 2.36 / 2
    assert oracle.morton3d(cx, cy, cz) == 1008806316530991104
    # interleave order (xx<<2)|(yy<<1)|zz, morton.h:86
    off = np.zeros(3); span = np.full(3, 1048576.0)
    assert oracle.morton3d(1.0, 0.0, 0.0, off, span) == 4
    assert oracle.morton3d(0.0, 1.0, 0.0, off, span) == 2
    assert oracle.morton3d(0.0, 0.0, 1.0, off, span) == 1


KAT_KEYS = np.array([1, 2, 4, 5, 19, 24, 25, 30], dtype=np.uint64)          # check.cuh:21
KAT_RANGES = [(0, 7), (0, 1), (2, 3), (0, 3), (4, 7), (5, 7), (5, 6)]
KAT_SPLITS = [3, 0, 2, 1, 4, 6, 5]


@pytest.mark.parametrize("tiebreak", [0, 1])
def test_range_split_kat(tiebreak):
    # check.cuh:22-23: determineRange(nums, 8, 6) -> (5,6), findSplit -> 5
    assert oracle.determine_range(KAT_KEYS, 6, tiebreak) == (5, 6)
    assert oracle.find_split(KAT_KEYS, 5, 6, tiebreak) == 5
    for i in range(7):
        r = oracle.determine_range(KAT_KEYS, i, tiebreak)
        assert r == KAT_RANGES[i]
        assert oracle.find_split(KAT_KEYS, r[0], r[1], tiebreak) == KAT_SPLITS[i]


def test_kat_by_hand():
    """Independent derivation of the same table: the radix tree over the 5-bit keys
    00001 00010 00100 00101 10011 11000 11001 11110 -- node i's range is the maximal run around i that
    shares the longer of its two neighbour prefixes, the split is where the top differing bit flips."""
    keys = [int(k) for k in KAT_KEYS]

    def lcp(a, b):
        return 64 - (a ^ b).bit_length()

    for i in range(7):
        dl = lcp(keys[i], keys[i - 1]) if i > 0 else -1
        dr = lcp(keys[i], keys[i + 1])
        d = 1 if dr - dl >= 0 else -1
        dmin = dl if d == 1 else dr
        j = i
        while 0 <= j + d < 8 and lcp(keys[i], keys[j + d]) > dmin:
            j += d
        first, last = min(i, j), max(i, j)
        assert (first, last) == KAT_RANGES[i]
        common = lcp(keys[first], keys[last])
        split = max(s for s in range(first, last) if lcp(keys[first], keys[s]) > common)
        assert split == KAT_SPLITS[i]


def test_literal_and_tiebreak_modes_agree_on_unique_keys():
    rng = np.random.default_rng(0)
    keys = np.unique(rng.integers(0, 1 << 62, 5000, dtype=np.uint64))
    a = oracle.build_hierarchy(keys, tiebreak=0)
    b = oracle.build_hierarchy(keys, tiebreak=1)
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)
    assert a[5] == b[5] == 0


def test_duplicate_keys_tiebreak_builds_valid_tree():
    keys = np.sort(np.array([5, 5, 5, 5, 7, 7, 9, 9, 9, 9, 9, 12], dtype=np.uint64))
    left, right, parent, rf, rl, wrong = oracle.build_hierarchy(keys, tiebreak=1)
    n = len(keys)
    assert wrong == 0
    assert (parent == -1).sum() == 1 and parent[0] == -1
    # every node except the root is the child of exactly one internal node
    seen = np.zeros(2 * n - 1, dtype=int)
    for i in range(n - 1):
        seen[left[i]] += 1; seen[right[i]] += 1
    assert seen[0] == 0 and (seen[1:] == 1).all()


def test_hit_anchor():
    s = np.zeros(1, dtype=oracle.SPHERE_DTYPE)
    s["x"], s["y"], s["z"], s["radius"], s["idx"] = 10, -3, 100, 20, 0
    t, n = oracle.rt_hit(s, 15.0, 2.0, np.zeros(4, dtype=np.int32))
    assert "%.6f" % t == "118.708282"
    assert "%.9f" % n == "0.935414314"
    t, n = oracle.rt_hit(s, 40.0, 2.0, np.zeros(4, dtype=np.int32))
    assert t == np.float32(-2e10)


def _parse_pairs(path, pat):
    out = []
    with open(path, errors="replace") as f:
        for line in f:
            m = re.search(pat, line)
            if m:
                out.append((int(m.group(1)), int(m.group(2))))
    return out


def test_reference_result_files_agree():
    """resources/MyResult.txt (the author's GPU run) and resources/flag-2000-changed.txt (an external CPU
    tool) list the same 20 pairs; every pair is written smaller ID first (tri_contact.cuh:81)."""
    gpu = _parse_pairs(os.path.join(GOLD, "ref_MyResult.txt"), r"^(\d{9}) - (\d{9})")
    cpu = _parse_pairs(os.path.join(GOLD, "ref_flag-2000-changed.txt"), r"#self contact found at \((\d+), (\d+)\)")
    assert len(gpu) == 20 and len(cpu) == 20
    assert set(gpu) == set(cpu)
    assert all(a < b for a, b in gpu)


@pytest.mark.parametrize("n,e,seed", [(2000, 0.08, 1), (6000, 0.05, 2)])
def test_tree_traversal_equals_brute_force(n, e, seed):
    verts, vidx = synth.soup(n, e, seed)
    r = oracle.pipeline(verts, vidx)
    bf_pairs, bf_n, bf_tested = oracle.brute_force(verts, vidx, box_filter=True)
    assert r["stats"].n_pairs == bf_n > 0
    assert np.array_equal(oracle.pair_set(r["pairs"]), oracle.pair_set(bf_pairs))
    assert r["stats"].pairs_tested == bf_tested             # leaf AABB hits == box-overlapping ordered pairs
    # literal checkDirectComp (no box gate) finds the same contacts on a soup: contact implies AABB overlap
    lit_pairs, lit_n, _ = oracle.brute_force(verts, vidx, box_filter=False)
    assert lit_n == bf_n and np.array_equal(oracle.pair_set(lit_pairs), oracle.pair_set(bf_pairs))
    assert (r["pairs"][:, 0] < r["pairs"][:, 1]).all()


def test_structural_counters_match_reference_expectations():
    """resources/cleanResult.png: wrongParentNum 0; internal: nullParent 1 (root), rest 0; leaf: all 0."""
    verts, vidx = synth.cloth_pair(20)
    r = oracle.pipeline(verts, vidx)
    n = vidx.shape[0]
    assert r["parent_wrong"] == 0
    import ctypes as C
    out5 = np.zeros(5, dtype=np.uint32); out4 = np.zeros(4, dtype=np.uint32)
    init = np.ones(2 * n - 1, dtype=np.uint8)
    L = oracle.lib()
    L.orc_check_internal(n, oracle._p(r["left"]), oracle._p(r["right"]), oracle._p(r["parent"]), oracle._p(r["bounded"]), oracle._p(init), oracle._p(out5))
    L.orc_check_leaves(n, oracle._p(r["parent"]), oracle._p(r["perm"]), oracle._p(vidx), verts.shape[0], oracle._p(init), oracle._p(out4))
    assert out5.tolist() == [1, 0, 0, 0, 0]
    assert out4.tolist() == [0, 0, 0, 0]
    assert L.orc_check_triangle_idx(n, oracle._p(r["perm"]), oracle._p(vidx), verts.shape[0]) == 0
    assert L.orc_check_triangle_idx(n, oracle._p(r["perm"]), oracle._p(vidx), 10) > 0
    assert (r["bounded"] == 2).all()
    assert r["child_count"][0] == 2 * n - 1                 # bvh.cuh:279 on the root


def test_boxes_contain_children_and_root_is_scene_box():
    verts, vidx = synth.soup(3000, 0.05, 4)
    r = oracle.pipeline(verts, vidx)
    b = r["boxes"]
    assert np.allclose(b[0, 0::2], verts.min(0)) and np.allclose(b[0, 1::2], verts.max(0))
    for i in range(len(r["left"])):
        for c in (r["left"][i], r["right"][i]):
            assert (b[i, 0::2] <= b[c, 0::2]).all() and (b[i, 1::2] >= b[c, 1::2]).all()


def test_sort_is_stable_and_matches_numpy():
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 1 << 40, 20000, dtype=np.uint64)
    keys[::7] = keys[0]                                     # many duplicates
    k, perm = oracle.sort_by_key(keys)
    want = np.argsort(keys, kind="stable")
    assert np.array_equal(perm, want.astype(np.uint32))
    assert np.array_equal(k, keys[want])


def test_sat_simple_cases():
    A = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    through = [[0.2, 0.2, -1], [0.2, 0.2, 1], [0.3, 0.9, 1]]      # pierces A
    far = [[0, 0, 5], [1, 0, 5], [0, 1, 5]]                       # parallel, 5 above
    touch = [[1, 0, 0], [2, 0, 0], [1, 1, 0]]                     # shares the point (1,0,0): contact is non-strict
    coplanar_apart = [[3, 3, 0], [4, 3, 0], [3, 4, 0]]
    assert oracle.tri_contact(A, through) == 1
    assert oracle.tri_contact(A, far) == 0
    assert oracle.tri_contact(A, touch) == 1                      # vec3f.cuh:288-289 uses >, touching counts
    assert oracle.tri_contact(A, coplanar_apart) == 0


def test_xorwow_restatement_structure():
    """The XORWOW generator behind the animation kernels (oracle/rt_oracle.c; random stream: parity unpinned, see the
    header there).  What CAN be checked without cuRAND: the recurrence is Marsaglia's xorwow (period structure: the
    Weyl counter advances by 362437 per draw and is added to the xorshift word), seeds give distinct streams, and the
    state after k draws is reproducible."""
    a, sa = oracle.xorwow_stream(0, 8)
    b, sb = oracle.xorwow_stream(0, 8)
    c, _ = oracle.xorwow_stream(1, 8)
    assert np.array_equal(a, b) and np.array_equal(sa, sb) and not np.array_equal(a, c)
    # independent restatement in Python integers
    def stream(seed, k):
        M = 0xFFFFFFFF
        s0 = (seed & M) ^ 0xaad26b49; s1 = ((seed >> 32) & M) ^ 0xf7dcefdd
        t0 = (1099087573 * s0) & M; t1 = (2591861531 * s1) & M
        d = (6615241 + t1 + t0) & M
        v = [(123456789 + t0) & M, 362436069 ^ t0, (521288629 + t1) & M, 88675123 ^ t1, (5783321 + t0) & M]
        out = []
        for _ in range(k):
            t = v[0] ^ (v[0] >> 2)
            v = v[1:] + [((v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1))) & M]
            d = (d + 362437) & M
            out.append((v[4] + d) & M)
        return out
    assert a.tolist() == stream(0, 8) and c.tolist() == stream(1, 8)
    big, _ = oracle.xorwow_stream(0x1234567890, 4)
    assert big.tolist() == stream(0x1234567890, 4)


def test_animation_oracle_basics():
    an = oracle.RtAnim(10)
    assert an.shifts[:, 2].tolist() == [5, 10, 15, 20, 25, 5, 10, 15, 20, 25]          # sphere.cuh:56
    assert an.shifts[:, 3].tolist() == [-1, 1] * 5                                     # sphere.cuh:57
    an.curve_move()                                                                    # angle 0: x += speed, y += 0
    assert an.shifts[:, 0].tolist() == [5, 10, 15, 20, 25, 5, 10, 15, 20, 25] and not an.shifts[:, 1].any()
    assert np.allclose(np.abs(an.angles), 3.1415926535898 / 12)
    an.axis_move(35)
    assert ((an.shifts[:, :2] >= 0) & (an.shifts[:, :2] < 35)).all()


# ---------------------------------------------------------------------------------------------------------------------
# Reference-COMPILED pin of rows a3 / a4 (and the key half of a5): tests/golden/morton_ref.npz holds outputs of the
# reference's own morton.h (expand64Bits :7-29, normX/Y/Z :43-58, morton3D :70-89), compiled unmodified into
# oracle/_ref/libref_morton.so in the build container (oracle/Makefile, tests/golden/make_morton_ref.py).  The inputs
# are recipes (tests/morton_inputs.py) whose SHA-256 the fixture records.
import morton_inputs as mi  # noqa: E402


@pytest.fixture(scope="module")
def morton_ref():
    return np.load(os.path.join(GOLD, "morton_ref.npz"))


def test_ref_compiled_expand64(morton_ref):
    v = mi.expand_inputs()
    assert mi.sha(v) == str(morton_ref["expand_in_sha"])
    want = morton_ref["expand_out"]
    assert v.size >= 100000 and want.shape == v.shape
    assert np.array_equal(oracle.expand64_batch(v), want)
    assert [oracle.expand64(int(x)) for x in v[:256]] == want[:256].tolist()          # the scalar entry point too


def test_ref_compiled_anchors(morton_ref):
    # the two anchors SURVEY.md recorded, as the reference's own object code computes them
    assert morton_ref["anchor_keys"].tolist() == [384255804010903211, 1008806316530991104]
    a = morton_ref["anchor_points"]
    assert [oracle.morton3d(*a[0]), oracle.morton3d(*a[1])] == morton_ref["anchor_keys"].tolist()


def test_ref_compiled_morton3d_points(morton_ref):
    """All 2^20 in-frame points (float-valued, full doubles, cell boundaries +- 1 ulp, frame faces) through orc_morton3d:
    the head in full, every 64th key, and the SHA-256 of all keys equal the reference-compiled ones."""
    pts = mi.frame_points()
    assert pts.shape[0] >= 1000000 and mi.sha(pts) == str(morton_ref["points_in_sha"])
    keys = oracle.morton3d_batch(pts)
    assert np.array_equal(keys[:mi.N_FULL], morton_ref["points_keys_head"])
    assert np.array_equal(keys[::mi.SAMPLE_STRIDE], morton_ref["points_keys_sample"])
    assert mi.sha(keys) == str(morton_ref["points_keys_sha"])
    assert int(keys.max()) < (1 << 60) and np.unique(keys >> np.uint64(57)).size == 8      # in-frame keys: 60 bits, every top octant hit


def test_ref_compiled_config3_keys(morton_ref):
    """BASELINE config 3: centroids (load_obj.h:89-101 operand order) -> morton3D, all 1 000 000 keys, through the oracle's
    own centroid + key loop (orc_centroid_morton), compared by SHA-256 with the reference-compiled keys; and sorted."""
    cen, verts, vidx = mi.cloth_centroids(500)
    assert mi.sha(verts) == str(morton_ref["cloth_verts_sha"]) and mi.sha(vidx) == str(morton_ref["cloth_vidx_sha"])
    keys, ocen = oracle.centroid_morton(verts, vidx, want_centroids=True)
    assert mi.sha(ocen) == str(morton_ref["cloth_centroids_sha"])          # the oracle's centroids == the ones the reference keys were made from
    assert np.array_equal(keys[::mi.SAMPLE_STRIDE], morton_ref["cloth_keys_sample"])
    assert mi.sha(keys) == str(morton_ref["cloth_keys_sha"])
    sk, _ = oracle.sort_by_key(keys)
    assert np.array_equal(sk[::mi.SAMPLE_STRIDE], morton_ref["cloth_sorted_sample"])
    assert mi.sha(sk) == str(morton_ref["cloth_sorted_sha"])
    assert [int(sk[0]), int(sk[-1])] == morton_ref["cloth_first_last"].tolist()


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libref_morton.so")),
                    reason="oracle/_ref exists only in the build container (the reference does not travel)")
def test_fixture_is_what_the_reference_build_produces_now(morton_ref, tmp_path):
    """Build container only: the committed fixture equals a fresh run of the reference-compiled library."""
    import ctypes as C
    L = C.CDLL(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libref_morton.so"))
    pts = mi.frame_points()
    k = np.zeros(pts.shape[0], dtype=np.uint64)
    L.ref_morton3D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]; L.ref_morton3D.restype = None
    L.ref_morton3D(pts.ctypes.data_as(C.c_void_p), pts.shape[0], k.ctypes.data_as(C.c_void_p))
    assert mi.sha(k) == str(morton_ref["points_keys_sha"])


# ---------------------------------------------------------------- pins against the REFERENCE'S OWN OBJECT CODE: the exact test
# tests/golden/contact_ref.npz holds what the reference's tri_contact.cuh:19-87, box.cuh:13-43, triangle.cuh:18-30,
# vec3f.cuh:118-125,257-291 (+ mathop.cuh:17-44), compiled UNMODIFIED by g++ against the genuine <cuda_runtime.h> this image
# ships, return for the recipes of tests/contact_inputs.py -- and, for BASELINE configs 2 and 3 (+ the 1 M soup and the
# full-double cloth), the END RESULT: the pair set and the pairs-tested count that follow from those predicates alone
# (collision.cuh:31-44 is only their sequence; the tree cannot change the set).  Generator: tests/golden/make_contact_ref.py.
import contact_inputs as ci  # noqa: E402


@pytest.fixture(scope="module")
def contact_ref():
    return np.load(os.path.join(GOLD, "contact_ref.npz"))


def _unbits(packed, n):
    return np.unpackbits(packed)[:n].astype(np.int32)


def test_ref_compiled_layouts(contact_ref):
    assert contact_ref["sizes"].tolist() == [24, 56, 56]              # vec3f, Triangle, Box as the reference's compiler lays them out


def test_ref_compiled_tri_contact(contact_ref):
    """checkTriangleContact on 1 179 648 pairs in seven families (float-valued, full doubles, exactly coplanar, touching,
    shared vertex positions, degenerate, integer lattice): every answer equals the reference-compiled one."""
    tri, fam = ci.tri_pairs()
    assert tri.shape[0] >= 1000000 and ci.sha(tri) == str(contact_ref["tri_in_sha"])
    want = _unbits(contact_ref["tri_contact_bits"], tri.shape[0])
    got = oracle.tri_contact_points(tri)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, (bad[:8], fam[bad[:8]])
    assert [int(got[fam == f].sum()) for f in range(len(ci.FAMILIES))] == contact_ref["tri_contact_per_family"].tolist()
    for f in range(len(ci.FAMILIES)):                                  # every family decides both ways somewhere (or is all-contact by construction)
        assert got[fam == f].any()
    assert (got[fam == 0] == 0).any() and (got[fam == 2] == 0).any() and (got[fam == 5] == 0).any() and (got[fam == 6] == 0).any()


def test_ref_compiled_tri_contact_on_non_finite_coordinates():
    """Round 6: 65 536 pairs with NaN, +-inf, +-0, huge and tiny coordinates (contact_inputs.nonfinite_pairs; mathop.cuh:17-44's compare-selects are NaN-asymmetric, and
    what tri_contact.cuh returns for such a pair follows from them): the restatement gives the reference-compiled verdict on every one
    (tests/golden/contact_nonfinite_ref.npz, generator make_contact_nonfinite_ref.py).  The device's hardware-max / min SAT and its fall-back are held to the same
    fixture in tests/test_cd_gpu.py."""
    ref = np.load(os.path.join(GOLD, "contact_nonfinite_ref.npz"))
    tri = ci.nonfinite_pairs()
    n = tri.shape[0]
    assert n == ci.N_NONFINITE and ci.sha(tri) == str(ref["tri_in_sha"])
    want = _unbits(ref["tri_contact_bits"], n)
    got = oracle.tri_contact_points(tri)
    assert np.array_equal(got, want) and int(want.sum()) == int(ref["contacts"])
    nan = np.isnan(tri).any(axis=(1, 2))
    assert int(nan.sum()) == int(ref["pairs_with_a_nan"]) and want[nan].any() and (want[nan] == 0).any()       # pairs with a NaN decide both ways


def test_ref_compiled_helper_and_neighbor_count(contact_ref):
    verts, va, ida, vb, idb = ci.indexed_pairs()
    blob = np.concatenate([verts.view(np.uint8).ravel(), va.view(np.uint8).ravel(), ida.view(np.uint8), vb.view(np.uint8).ravel(), idb.view(np.uint8)])
    assert ci.sha(blob) == str(contact_ref["indexed_in_sha"])
    n = va.shape[0]
    assert np.array_equal(oracle.helper_batch(verts, va, ida, vb, idb), _unbits(contact_ref["helper_bits"], n))
    nc = oracle.neighbor_count_batch(va, vb)
    assert np.array_equal(nc, contact_ref["neighbor"].astype(np.int32))
    assert set(np.unique(nc).tolist()) >= {0, 1, 2, 3, 4}             # repeats inside a triangle push the count past 3 (triangle.cuh:19-29)
    assert (ida >= idb).any() and (ida < idb).any()                    # the ID rule, tri_contact.cuh:81, on both sides


def test_ref_compiled_boxes(contact_ref):
    """Box::set / Box::merge bit patterns and checkBoxOverlap on 2^20 pairs (touching faces, zero thickness, identical)."""
    verts, va, ida, vb, idb = ci.indexed_pairs()
    n = va.shape[0]
    bs = oracle.box_set_batch(verts, np.concatenate([va, vb]))
    assert np.array_equal(bs[:4096].view(np.uint64), contact_ref["box_set_head"].view(np.uint64)) and ci.sha(bs) == str(contact_ref["box_set_sha"])
    bm = oracle.box_merge_batch(bs[:n], bs[n:])
    assert np.array_equal(bm[:4096].view(np.uint64), contact_ref["box_merge_head"].view(np.uint64)) and ci.sha(bm) == str(contact_ref["box_merge_sha"])
    a, b = ci.box_pairs()
    assert a.shape[0] >= 1000000 and ci.sha(np.concatenate([a, b])) == str(contact_ref["box_in_sha"])
    want = _unbits(contact_ref["box_overlap_bits"], a.shape[0])
    got = oracle.box_overlap_batch(a, b)
    assert np.array_equal(got, want)
    touching = ((a[:, 1] == b[:, 0]) | (b[:, 1] == a[:, 0])) & (a[:, 1] > a[:, 0]) & (b[:, 1] > b[:, 0])
    assert touching.sum() > 10000 and not got[touching].any()          # a shared face is NOT an overlap (strict '>', box.cuh:41)
    flat = (a[:, 0] == a[:, 1]) | (a[:, 2] == a[:, 3]) | (a[:, 4] == a[:, 5])
    assert flat.sum() > 10000                                          # zero-thickness boxes are in the set ...
    assert np.array_equal(oracle.box_overlap_batch(a[flat], a[flat]), np.zeros(int(flat.sum()), dtype=np.int32))   # ... and never overlap themselves


def test_ref_compiled_vec3f_helpers(contact_ref):
    v4, v7, v2 = ci.small_vectors()
    assert ci.sha(np.concatenate([v4.ravel(), v7.ravel(), v2.ravel()])) == str(contact_ref["small_in_sha"])
    assert np.array_equal(oracle.project3_batch(v4), _unbits(contact_ref["project3_bits"], v4.shape[0]))
    assert np.array_equal(oracle.project6_batch(v7), _unbits(contact_ref["project6_bits"], v7.shape[0]))
    cr, dt = oracle.cross_dot_batch(v2)
    assert np.array_equal(cr.view(np.uint64), contact_ref["cross"].view(np.uint64)) and np.array_equal(dt.view(np.uint64), contact_ref["dot"].view(np.uint64))


def check_end_result(ref, name, pairs, pairs_tested):
    """pairs / pairs_tested of one run against the reference-compiled END RESULT of config `name` (shared with the GPU tests)."""
    keys = ci.pair_keys(pairs)
    assert keys.size == int(ref[name + "_count"]), (name, keys.size, int(ref[name + "_count"]))
    assert int(pairs_tested) == int(ref[name + "_tested"]), (name, int(pairs_tested), int(ref[name + "_tested"]))
    assert np.array_equal(keys[::ci.SAMPLE_STRIDE], ref[name + "_sample"])
    if name + "_pairs" in ref.files:
        assert np.array_equal(keys, ref[name + "_pairs"])
    assert ci.sha(keys) == str(ref[name + "_pairs_sha"])


def end_mesh(name):
    for n, mesh, _ in ci.end_configs():
        if n == name:
            return mesh
    raise KeyError(name)


@pytest.mark.parametrize("name", ["soup100k", "cloth1M", "soup1M", "cloth1M_double", "soup100k_mt64", "soup1M_mt64"])
def test_ref_compiled_end_result(contact_ref, name):
    """The oracle's whole pipeline (Morton build + traversal, orc_self_collide) returns the reference-compiled pair set and
    pairs-tested count on BASELINE config 2 (100 k soup; the reference side is a plain O(N^2) over its predicates),
    config 3 (1 M cloth), the 1 M soup and the full-double cloth."""
    verts, vidx = end_mesh(name)
    assert ci.sha(verts) == str(contact_ref[name + "_verts_sha"]) and ci.sha(vidx) == str(contact_ref[name + "_vidx_sha"])
    pairs, st, _ = oracle.self_collide(verts, vidx)
    assert st.overflow == 0
    check_end_result(contact_ref, name, pairs, st.pairs_tested)


def test_ref_compiled_end_result_brute_force_agrees(contact_ref):
    """The oracle's own O(N^2) (check.cuh:117-141 restated, box filter on) against the same fixture: config 2's first 20 000
    triangles cannot be cut out of the fixture, so this runs the full 100 k only through the tree (above) and checks here that
    brute force == tree on a prefix, closing the triangle oracle-tree == reference-compiled == oracle-brute."""
    verts, vidx = end_mesh("soup100k")
    sub = vidx[:12000]
    bp, bn, bt = oracle.brute_force(verts, sub)
    tp, st, _ = oracle.self_collide(verts, sub)
    assert np.array_equal(ci.pair_keys(bp), ci.pair_keys(tp)) and bt == st.pairs_tested


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libref_contact.so")),
                    reason="oracle/_ref exists only in the build container (the reference does not travel)")
def test_contact_fixture_is_what_the_reference_build_produces_now(contact_ref):
    """Build container only: the committed fixture equals a fresh run of the reference-compiled library (tri_contact, all pairs)."""
    import ctypes as C
    L = C.CDLL(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libref_contact.so"))
    tri, _ = ci.tri_pairs()
    r = np.zeros(tri.shape[0], dtype=np.int32)
    L.ref_tri_contact.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]; L.ref_tri_contact.restype = None
    L.ref_tri_contact(tri.ctypes.data_as(C.c_void_p), tri.shape[0], r.ctypes.data_as(C.c_void_p))
    assert ci.sha(r.astype(np.uint8)) == str(contact_ref["tri_contact_sha"])
    a, b = ci.box_pairs()
    o = np.zeros(a.shape[0], dtype=np.int32)
    L.ref_box_overlap.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]; L.ref_box_overlap.restype = None
    L.ref_box_overlap(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), a.shape[0], o.ctypes.data_as(C.c_void_p))
    assert np.array_equal(np.packbits(o.astype(np.uint8)), contact_ref["box_overlap_bits"])
