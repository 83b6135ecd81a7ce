"""GPU parity tests for the collision path: every stage of libmi355cd.so (through the C ABI) against
the CPU oracle on the same seeded inputs.  Bar: bit-exact (integer keys / indices / tree links, FP64
box bits, pair index SETS -- the reference's own output order is an atomicAdd race, collision.cuh:40)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import mi355_synth as synth
import mi355cd
import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


VARIANTS = [0, 1, 3]        # CD_OPT_TRAVERSAL: lane-private FP64 descent / per-lane fp32 descent from the root + exact kernel /
                            # half traversal: bottom-up chain of right siblings, the subtrees it hits descended by a second kernel, + exact kernel
SHALLOW_LAUNCHES = 2        # kernels of the default traversal's first pass: descent + exact


def _check_visits(st, ref_stats, variant):
    """Variant 0 walks exactly the oracle's nodes; variant 2 descends with conservative fp32 boxes, so it may visit a
    few more internal nodes (never fewer); variant 1 also walks the root-to-first-leaf path once per wave and counts
    those (cheap, scalar-fetched) steps per lane, so its count is of the same order only -- a diagnostic, not a result.
    Variant 3 visits only what lies to the right of each query (about half of it) plus one record per chain hop."""
    if variant == 0:
        assert st.node_visits == ref_stats.node_visits
    elif variant == 1:
        assert 0.5 * ref_stats.node_visits <= st.node_visits <= ref_stats.node_visits * 2 + 64
    elif variant >= 3:
        assert 0 < st.node_visits <= ref_stats.node_visits * 2 + 64
    else:
        assert ref_stats.node_visits <= st.node_visits <= ref_stats.node_visits * 1.02 + 16


def _stagewise(verts, vidx, ids=None, frame=mi355cd.CD_FRAME_REFERENCE, off=None, span=None, variant=1):
    cd = mi355cd.CollisionDetector(verts, vidx, ids)
    cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
    cd.set_morton_frame(frame, off, span)
    cd.morton_sort()
    keys, perm = cd.export_keys()
    wrong = cd.build_hierarchy()
    cd.refit_boxes()
    parent, left, right, boxes, bounded = cd.export_tree()
    return cd, dict(keys=keys, perm=perm, wrong=wrong, parent=parent, left=left, right=right, boxes=boxes, bounded=bounded)


def _assert_tree_equal(g, r):
    assert np.array_equal(g["keys"], r["keys"])
    assert np.array_equal(g["perm"], r["perm"])
    assert g["wrong"] == r["parent_wrong"] == 0
    assert np.array_equal(g["left"], r["left"]) and np.array_equal(g["right"], r["right"])
    assert np.array_equal(g["parent"], r["parent"])
    assert np.array_equal(g["boxes"].view(np.uint64), r["boxes"].view(np.uint64))     # FP64 bit patterns
    assert (g["bounded"] == 2).all()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n,e,seed", [(1000, 0.1, 11), (100_000, 0.02, 1234)])
def test_soup_every_stage_matches_oracle(n, e, seed, variant):
    """BASELINE config 2 (100 k random-triangle soup) and a small one."""
    verts, vidx = synth.soup(n, e, seed)
    cd, g = _stagewise(verts, vidx, variant=variant)
    r = oracle.pipeline(verts, vidx)
    _assert_tree_equal(g, r)
    assert cd.check_internal().tolist() == [1, 0, 0, 0, 0]          # cleanResult.png expectations
    assert cd.check_leaves().tolist() == [0, 0, 0, 0]
    assert cd.check_triangle_idx(verts.shape[0]) == 0
    assert cd.check_triangle_idx(5) == int((vidx >= 5).sum())
    pairs, npairs, rc = cd.find_collisions(cap=1 << 20)
    st = cd.stats()
    assert rc == 0 and npairs == r["stats"].n_pairs
    assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
    assert st.pairs_tested == r["stats"].pairs_tested
    _check_visits(st, r["stats"], variant)
    assert (pairs[:, 0] < pairs[:, 1]).all()
    cd.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_cloth_pair_shared_vertices(variant):
    """Config-3 geometry at 1/6 scale (80 k triangles): shared vertices exercise neighborCount."""
    verts, vidx = synth.cloth_pair(100)
    cd, g = _stagewise(verts, vidx, variant=variant)
    r = oracle.pipeline(verts, vidx)
    _assert_tree_equal(g, r)
    pairs, npairs, rc = cd.find_collisions(cap=1 << 20)
    assert rc == 0 and npairs == r["stats"].n_pairs > 0
    assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
    assert cd.stats().pairs_tested == r["stats"].pairs_tested
    # fused call gives the same set
    pairs2, n2, rc2 = cd.self_collide(cap=1 << 20)
    assert rc2 == 0 and np.array_equal(oracle.pair_set(pairs2), oracle.pair_set(pairs))
    cd.close()


def test_auto_frame_and_custom_ids():
    """CD_FRAME_AUTO (the adaptive frame, cd_math.h): frame and key layout as the oracle's restatement forms them, bit for bit; keys, tree, boxes and pairs of that
    frame; the frame kept (cd_get_morton_frame -> cd_set_morton_frame_layout) gives the same keys; morton.h:70-89's interleave in the same offset / span does not."""
    verts, vidx = synth.soup(20000, 0.04, 5)
    verts = verts * 37.0 + 1000.0                           # far outside the reference frame
    ids = (np.arange(vidx.shape[0], dtype=np.uint32)[::-1] * 3 + 7).astype(np.uint32)
    off, span, lay = oracle.auto_frame(verts, vidx)
    cd, g = _stagewise(verts, vidx, ids, frame=mi355cd.CD_FRAME_AUTO)
    goff, gspan, glay = cd.get_morton_frame()
    assert glay == lay and np.array_equal(goff, off) and np.array_equal(gspan, span)
    r = oracle.pipeline(verts, vidx, ids, off=off, span=span, layout=lay)
    _assert_tree_equal(g, r)
    pairs, npairs, rc = cd.find_collisions()
    assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])) and npairs > 0
    # the frame kept == what auto computed
    cd2 = mi355cd.CollisionDetector(verts, vidx, ids)
    cd2.set_morton_frame_layout(off, span, lay)
    cd2.morton_sort()
    assert np.array_equal(cd2.export_keys()[0], g["keys"]) and cd2.get_morton_frame()[2] == lay
    # the reference's interleave in the same offset / span: other keys (layout 0), the same pairs
    cd3, g3 = _stagewise(verts, vidx, ids, frame=mi355cd.CD_FRAME_CUSTOM, off=off, span=span)
    assert cd3.get_morton_frame()[2] == 0 and not np.array_equal(g3["keys"], g["keys"])
    assert np.array_equal(g3["keys"], oracle.pipeline(verts, vidx, ids, off=off, span=span)["keys"])
    p3, n3, _ = cd3.find_collisions()
    assert np.array_equal(oracle.pair_set(p3), oracle.pair_set(pairs))
    # words that are not layouts are refused
    for bad in (1, (1 << 63) | 0 | (0 << 2) | (1 << 4), oracle.layout_word((0, 1, 2), 30, 10, 5), oracle.layout_word((0, 1, 2), 0, 0, 21), oracle.layout_word((0, 1, 2), 0, 31, 0) ,
                oracle.layout_word((0, 1, 3), 0, 0, 20), oracle.layout_word((0, 1, 2), 0, 0, 20) | (1 << 40)):
        with pytest.raises(mi355cd.CdError):
            cd2.set_morton_frame_layout(off, span, bad)
    with mi355cd.CollisionDetector(verts, vidx, ids) as fresh:
        with pytest.raises(mi355cd.CdError) as e:
            fresh.get_morton_frame()
        assert e.value.rc == mi355cd.CD_ERR_ORDER
    cd.close(); cd2.close(); cd3.close()


def _flat_sheets(k=6, q=40):
    vs, ts, o = [], [], 0
    for s in range(k):
        v, t = synth._sheet(q, q, lambda X, Y: 0.0 * X + 0.01 * s, 0.003 * s, 4.0 + 0.003 * s, 0.0, 1.0)
        vs.append(v[:, [0, 2, 1]]); ts.append(t + np.uint32(o)); o += v.shape[0]
    return np.ascontiguousarray(np.concatenate(vs).astype(np.float32).astype(np.float64)), np.ascontiguousarray(np.concatenate(ts).astype(np.uint32))


@pytest.mark.parametrize("mesh", ["thin_long", "flat_sheets", "soup", "line", "one_point", "full_double", "huge_coordinates"])
def test_auto_frame_is_the_oracles_on_every_shape_of_mesh(mesh):
    """The statistic is summed as integers and min / max are exact: whatever order the device reduces in, frame and layout equal the sequential restatement's.
    thin_long: config 4's shape; flat_sheets: every box flat along y (no statistic there: the cap); line: two axes without extent; one_point: none."""
    ids = None
    if mesh == "thin_long": verts, vidx, ids, _, _ = synth.config4_merged(8, 24)
    elif mesh == "flat_sheets": verts, vidx = _flat_sheets()
    elif mesh == "soup": verts, vidx = synth.soup(30000, 0.03, 9)
    elif mesh == "line":
        n = 5000; x = np.arange(3 * n, dtype=np.float64) * 0.25
        verts = np.stack([x, np.full(3 * n, 2.0), np.full(3 * n, -1.0)], 1); vidx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    elif mesh == "one_point":
        verts = np.tile(np.array([[1.0, 2.0, 3.0]]), (300, 1)); vidx = np.arange(300, dtype=np.uint32).reshape(100, 3)
    elif mesh == "full_double": verts, vidx = synth.cloth_pair(60, round_f32=False)
    else:
        verts, vidx = synth.soup(20000, 0.04, 5); verts = verts * 1e12 - 3e13
    off, span, lay = oracle.auto_frame(verts, vidx)
    with mi355cd.CollisionDetector(verts, vidx, ids) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_AUTO)
        pairs, n, rc = cd.self_collide(cap=1 << 20)
        goff, gspan, glay = cd.get_morton_frame()
        assert glay == lay and np.array_equal(goff, off) and np.array_equal(gspan, span), (oracle.layout_fields(glay), oracle.layout_fields(lay))
        keys, perm = cd.export_keys()
        r = oracle.pipeline(verts, vidx, ids, off=off, span=span, layout=lay)
        assert np.array_equal(keys, r["keys"]) and np.array_equal(perm, r["perm"])
        assert rc == 0 and n == r["stats"].n_pairs and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
        assert cd.stats().pairs_tested == r["stats"].pairs_tested


def test_adaptive_keys_of_explicit_points_match_the_oracle():
    """cd_morton3d_points_layout (what k_morton computes per triangle in a frame with a layout; the points are vertex SUMS there) against the restatement: layouts of
    every shape, sums inside, on cell boundaries +- 1 ulp, outside the frame, NaN; and layout 0 against the reference-pinned morton3D itself."""
    rng = np.random.default_rng(77)
    words = [oracle.layout_word((0, 2, 1), 3, 3, 17), oracle.layout_word((2, 0, 1), 0, 3, 18), oracle.layout_word((1, 0, 2), 60, 0, 0), oracle.layout_word((0, 1, 2), 0, 30, 0),
             oracle.layout_word((2, 1, 0), 10, 10, 10), oracle.layout_word((1, 2, 0), 5, 2, 7), oracle.layout_word((0, 1, 2), 0, 0, 20), oracle.layout_word((1, 2, 0), 0, 0, 0)]
    for w in words:
        off = rng.normal(size=3) * 5; span = np.exp(rng.normal(size=3) * 2)
        (A, B, C), nA, p, t = oracle.layout_fields(w)
        pts = (rng.random((40000, 3)) * span + off) * 3
        nb = min(nA + p + t, 40)
        edge = ((rng.integers(0, 1 << nb, size=(20000, 3)) / float(1 << nb)) * span + off) * 3      # near cell boundaries of axis A's grid (and whatever they are on the others)
        edge = np.concatenate([edge, np.nextafter(edge, np.inf), np.nextafter(edge, -np.inf)])
        out = np.concatenate([(off - span * rng.random((500, 3))) * 3, (off + span * (1 + rng.random((500, 3)))) * 3, np.full((4, 3), np.nan), np.array([[1e300, -1e300, 0.0]])])
        allp = np.concatenate([pts, edge, out])
        got = mi355cd.morton3d_points_layout(allp, off, span, w)
        assert np.array_equal(got, oracle.morton3d_layout_batch(allp, off, span, w)), oracle.layout_fields(w)
        assert int(got.max()) < 1 << 60
    pts = rng.random((100000, 3)) * oracle.REF_SPAN + oracle.REF_OFF
    assert np.array_equal(mi355cd.morton3d_points_layout(pts, oracle.REF_OFF, oracle.REF_SPAN, 0), mi355cd.morton3d_points(pts))
    assert np.array_equal(mi355cd.morton3d_points(pts), oracle.morton3d_batch(pts))


def test_thin_long_mesh_sorts_in_two_passes_and_walks_a_better_tree():
    """a thin, long mesh (config 4 merged, 21 x 0.05 x 2.2, 320 k triangles) in CD_FRAME_AUTO -- the oracle's pair set and pairs tested, the
    hybrid sort in its FIRST form (two global passes, small windows), and a tree that costs at least a quarter fewer node visits than round 5's per-axis frame
    (same call, CD_FRAME_CUSTOM with the per-axis spans = morton.h:70-89's interleave).  The frame kept afterwards: same keys, same result, no AUTO pass."""
    verts, vidx, ids, lo, span5 = synth.config4_merged(8, 100)
    r = oracle.pipeline(verts, vidx, ids, off=lo, span=span5)
    want = oracle.pair_set(r["pairs"])
    with mi355cd.CollisionDetector(verts, vidx, ids) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_AUTO)
        pairs, n, rc = cd.self_collide(cap=1 << 20)
        st = cd.stats()
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), want) and st.pairs_tested == r["stats"].pairs_tested
        assert st.sort_passes == 2 and cd.debug_get(mi355cd.CD_DBG_GET_SORT_FORM) == 0
        visits_auto = st.node_visits
        keys_auto = cd.export_keys()[0]
        off, sp, lay = cd.keep_auto_frame()
        (A, B, C), nA, p, t = oracle.layout_fields(lay)
        assert (A, B, C) == (0, 2, 1) and nA + 2 * p + 3 * t == 60 and p <= 4
        pairs, n, rc = cd.self_collide(cap=1 << 20)
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), want) and np.array_equal(cd.export_keys()[0], keys_auto) and cd.stats().sort_passes == 2
        cd.set_morton_frame(mi355cd.CD_FRAME_CUSTOM, lo, span5)
        pairs, n, rc = cd.self_collide(cap=1 << 20)
        st5 = cd.stats()
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), want) and st5.pairs_tested == st.pairs_tested
        assert visits_auto < 0.8 * st5.node_visits, (visits_auto, st5.node_visits)


def test_pair_set_is_frame_independent():
    verts, vidx = synth.soup(30000, 0.03, 9)
    sets = []
    for frame in (mi355cd.CD_FRAME_REFERENCE, mi355cd.CD_FRAME_AUTO):
        with mi355cd.CollisionDetector(verts, vidx) as cd:
            cd.set_morton_frame(frame)
            pairs, n, rc = cd.self_collide()
            sets.append(oracle.pair_set(pairs))
    assert np.array_equal(sets[0], sets[1]) and len(sets[0]) > 0


def test_duplicate_and_degenerate_triangles():
    """Identical triangles (duplicate Morton keys -- the case the reference's builder breaks on,
    load_obj.h:109-115), zero-area triangles, and a triangle listed twice with shared indices."""
    verts, vidx = synth.soup(500, 0.2, 21)
    v2 = np.concatenate([verts, verts[:300]], axis=0)                    # copies of the first 100 triangles
    dup = (np.arange(300, dtype=np.uint32) + verts.shape[0]).reshape(100, 3)
    degenerate = np.array([[0, 0, 1], [5, 5, 5]], dtype=np.uint32)       # zero-area
    same_idx = vidx[:50].copy()                                          # same vertex indices -> neighbours, never reported
    vi = np.concatenate([vidx, dup, degenerate, same_idx], axis=0)
    cd, g = _stagewise(v2, vi)
    r = oracle.pipeline(v2, vi)
    assert len(np.unique(g["keys"])) < len(g["keys"])                    # duplicates really present
    _assert_tree_equal(g, r)
    pairs, n, rc = cd.find_collisions()
    assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])) and n > 0
    bf, bn, _ = oracle.brute_force(v2, vi, box_filter=True)
    assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(bf))
    assert cd.check_internal().tolist() == [1, 0, 0, 0, 0]
    cd.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n", [1, 2, 3, 64, 65, 4097])
def test_tiny_and_ragged_sizes(n, variant):
    verts, vidx = synth.soup(max(n, 1), 0.5, 100 + n)
    verts, vidx = verts[: 3 * n], vidx[:n]
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
        pairs, cnt, rc = cd.self_collide()
        st = cd.stats()
        if n == 1:
            assert cnt == 0
            return
        r = oracle.pipeline(verts, vidx)
        assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
        assert st.pairs_tested == r["stats"].pairs_tested
        parent, left, right, boxes, bounded = cd.export_tree()
        assert np.array_equal(left, r["left"]) and np.array_equal(right, r["right"]) and np.array_equal(parent, r["parent"])


@pytest.mark.parametrize("qpw", [64, 128, 1024, 65536])
def test_wave_chunk_size_does_not_change_results(qpw):
    verts, vidx = synth.cloth_pair(50)
    r = oracle.pipeline(verts, vidx)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_option(mi355cd.CD_OPT_QUERIES_PER_WAVE, qpw)
        pairs, n, rc = cd.self_collide()
        assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
        assert cd.stats().pairs_tested == r["stats"].pairs_tested
        with pytest.raises(mi355cd.CdError):
            cd.set_option(mi355cd.CD_OPT_QUERIES_PER_WAVE, 100)


def test_dense_collisions_fill_the_candidate_queue():
    """Many leaf hits per query (big overlapping triangles): the wave queue drains repeatedly mid-descent."""
    verts, vidx = synth.soup(6000, 0.6, 77)
    r = oracle.pipeline(verts, vidx)
    assert r["stats"].pairs_tested > 50 * 6000
    for variant in VARIANTS:
        with mi355cd.CollisionDetector(verts, vidx) as cd:
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
            pairs, n, rc = cd.self_collide(cap=1 << 22)
            assert rc == 0 and n == r["stats"].n_pairs
            assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
            assert cd.stats().pairs_tested == r["stats"].pairs_tested


def test_survivors_that_overflow_their_shards_make_the_step_grow_the_buffer_and_redo():
    """The half traversal's survivors leave as 32-byte SAT-ready pairs from the END of their shard of the candidate buffer, what still needs the FP64 box test as
    8-byte candidates from its front (cd_traverse.h, FatPair).  2.1 M survivors x 4 slots against the 2^20 slots a 12 000-triangle context starts with: every
    shard overflows, nothing is written past one, the host grows the buffer and redoes the pass -- the oracle's pairs, and again without growth in the next step.
    A mesh with a cell table sends both kinds through one shard."""
    verts, vidx = synth.soup(12000, 0.6, 77)
    for scale in (1.0, 1.0 + 2.0 ** -30):                               # (x (1 + 2^-30): no coordinate is an fp32 value any more -- cell table, candidates that are not certain)
        v = verts * scale
        r = oracle.pipeline(v, vidx)
        with mi355cd.CollisionDetector(v, vidx) as cd:
            for step in range(3):
                pairs, n, rc = cd.self_collide(cap=1 << 20)
                st = cd.stats()
                assert rc == 0 and n == r["stats"].n_pairs and st.pairs_tested == r["stats"].pairs_tested
                assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
                assert 4 * st.candidates > (1 << 20) and 2 * st.candidates <= r["stats"].pairs_tested + 2 * 12000
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, 1)                   # the descent from the root hands its survivors over as flagged 8-byte candidates
            p1, n1, _ = cd.find_collisions(cap=1 << 20)
            assert n1 == n and np.array_equal(oracle.pair_set(p1), oracle.pair_set(r["pairs"])) and cd.stats().pairs_tested == r["stats"].pairs_tested


def test_capacity_overflow_and_stage_order():
    verts, vidx = synth.soup(20000, 0.05, 3)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        with pytest.raises(mi355cd.CdError) as e:
            cd.build_hierarchy()
        assert e.value.rc == mi355cd.CD_ERR_ORDER
        with pytest.raises(mi355cd.CdError):
            cd.find_collisions()
        full, n, rc = cd.self_collide(cap=1 << 16)
        assert rc == 0 and n > 10
        part, n2, rc2 = cd.find_collisions(cap=10)
        assert rc2 == mi355cd.CD_OVERFLOW and n2 == n and part.shape[0] == 10
        assert set(map(tuple, part.tolist())) <= set(map(tuple, full.tolist()))
        none, n3, rc3 = cd.find_collisions(cap=0)                         # count only
        assert n3 == n and rc3 == mi355cd.CD_OVERFLOW
    bad = vidx.copy(); bad[7, 1] = verts.shape[0]
    with pytest.raises(mi355cd.CdError) as e:
        mi355cd.CollisionDetector(verts, bad)
    assert e.value.rc == mi355cd.CD_ERR_INDEX


def _comb(codes, big_first=False):
    """Two tiny triangles per Morton code (decoded to grid cells of a unit-cell frame) + one triangle
    spanning everything (big_first: with a centroid below the frame, so that it sorts FIRST -- key 0)."""
    tris = []
    for code in codes:
        c = np.zeros(3)
        for p in range(60):
            if (code >> p) & 1:
                c[{2: 0, 1: 1, 0: 2}[p % 3]] += float(1 << (p // 3))
        c += 0.5
        for s in (0.0, 0.02):
            tris.append([c + [s, 0, 0], c + [s + 0.2, 0.1, 0], c + [s, 0.1, 0.2]])
    tris.append([[-9.0e6 if big_first else -1.0] * 3, [4.0e6, -1.0, -1.0], [-1.0, 4.0e6, 4.0e6]])
    verts = np.asarray(tris, dtype=np.float64).reshape(-1, 3)
    vidx = np.arange(verts.shape[0], dtype=np.uint32).reshape(-1, 3)
    return verts, vidx


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("mirrored", [False, True])
def test_deep_tree_uses_the_deferred_stack_path(mirrored, variant):
    """60 two-triangle clusters forming a 60-level comb, plus one triangle overlapping all of them.
    mirrored=False: the chain hangs on the LEFT child, so this library's descend-left/push-right order
    needs ~60 pending entries -- more than its 32-entry LDS stack -> deferred (query, subtree) items.
    mirrored=True: the chain hangs on the RIGHT, which is the orientation that overflows the REFERENCE's
    unchecked 32-entry stack (push L, push R, pop R; collision.cuh:21,47,64,69)."""
    off = np.zeros(3); span = np.full(3, 1048576.0)                       # coordinate == grid cell
    if not mirrored:
        codes = [1 << (59 - k) for k in range(60)]
    else:
        codes = [(1 << 60) - (1 << (60 - k)) for k in range(60)]
    verts, vidx = _comb(codes)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_CUSTOM, off, span)
        cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
        pairs, n, rc = cd.self_collide()
        st = cd.stats()
    r = oracle.pipeline(verts, vidx, off=off, span=span)
    if mirrored:
        assert r["stats"].max_stack > 32
    elif variant >= 3:
        pass                                                             # (the half traversal's overflow case: next test)
    else:
        assert st.stack_overflows > 0 and st.traverse_launches == (4 if variant == 1 else 2)
    assert n == r["stats"].n_pairs > 0
    assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
    assert st.pairs_tested == r["stats"].pairs_tested
    _check_visits(st, r["stats"], variant)


def test_half_traversal_chain_overflows_into_the_deep_pass():
    """Variant 3: the all-overlapping triangle sorts first, so the chain of right siblings of ITS leaf holds all 60
    clusters of the left-hanging comb -- more than a lane's 12 LDS stack entries -> deferred (query, subtree) items that
    the deep pass continues with the half traversal's counting."""
    off = np.zeros(3); span = np.full(3, 1048576.0)
    verts, vidx = _comb([1 << (59 - k) for k in range(60)], big_first=True)
    r = oracle.pipeline(verts, vidx, off=off, span=span)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_CUSTOM, off, span)
        pairs, n, rc = cd.self_collide()
        st = cd.stats()
        assert st.stack_overflows > 0 and st.traverse_launches == SHALLOW_LAUNCHES + 2
        assert n == r["stats"].n_pairs > 0 and st.pairs_tested == r["stats"].pairs_tested
        assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
        for variant in (0, 1):
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
            p2, n2, _ = cd.find_collisions()
            assert np.array_equal(oracle.pair_set(p2), oracle.pair_set(pairs)) and cd.stats().pairs_tested == st.pairs_tested


def test_half_traversal_instance_for_trees_beyond_2_27_leaves_gives_the_same_step():
    """k_descend_half forms record addresses from a 32-bit byte offset up to 2^27 leaves and from 64-bit indices beyond (cd_traverse.h, HALF_SMALL_N).  No test
    mesh is that large: CD_DBG_BIG_OFFSETS runs the 64-bit instance on trees of any size -- the same pairs, counters and node visits, with and without a cell table,
    in the polled graph step and the stage-wise call."""
    for verts, vidx in (synth.cloth_pair(60), synth.soup(50_000, 0.02, 5), (synth.cloth_pair(40)[0] * (1 + 2.0 ** -30) + 0.1, synth.cloth_pair(40)[1])):
        r = oracle.pipeline(verts, vidx)
        with mi355cd.CollisionDetector(verts, vidx) as cd:
            got = []
            for big in (0, 1, 0):
                cd.debug_set(mi355cd.CD_DBG_BIG_OFFSETS, big)
                for _ in range(3):
                    pairs, n, rc = cd.self_collide()
                st = cd.stats()
                assert rc == 0 and n == r["stats"].n_pairs and st.pairs_tested == r["stats"].pairs_tested
                assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
                p2, n2, _ = cd.find_collisions()
                assert np.array_equal(oracle.pair_set(p2), oracle.pair_set(r["pairs"]))
                got.append((st.node_visits, st.candidates))
            assert got[0] == got[1] == got[2]


def test_exact_test_kernel_one_million_pairs():
    """tri_contact (17-axis SAT) + neighbour gate + ID rule on 1 M explicit pairs, bit-match vs oracle."""
    verts, vidx = synth.cloth_pair(60)
    n = vidx.shape[0]
    rng = np.random.default_rng(17)
    a = rng.integers(0, n, 1_000_000, dtype=np.int64)
    half = n // 2
    # partner: same quad neighbourhood on the OTHER sheet (mix of hits and near misses) or a near neighbour on the same sheet
    other = (a + half) % n + rng.integers(-3, 4, a.shape[0])
    same = a + rng.integers(-2, 3, a.shape[0])
    b = np.where(rng.random(a.shape[0]) < 0.7, other, same) % n
    pairs = np.stack([a, b], axis=1).astype(np.uint32)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        got = cd.test_pairs(pairs)
    want = oracle.tri_contact_batch(verts, vidx, pairs)
    assert np.array_equal(got, want)
    assert 1000 < int(want.sum()) < 900_000


def test_gpu_brute_force_agrees_with_traversal():
    verts, vidx = synth.soup(20000, 0.05, 8)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        tp, tn, _ = cd.self_collide()
        st = cd.stats()
        bp, bn, _ = cd.brute_force(box_filter=True)
        lp, ln, _ = cd.brute_force(box_filter=False)
    assert tn == bn == ln > 0
    assert np.array_equal(oracle.pair_set(tp), oracle.pair_set(bp))
    assert np.array_equal(oracle.pair_set(tp), oracle.pair_set(lp))
    # the GPU all-pairs pass against the oracle's own all-pairs loop (check.cuh:117-141 restated), not only via the tree
    op, on, otested = oracle.brute_force(verts, vidx, box_filter=True)
    assert on == bn and np.array_equal(oracle.pair_set(bp), oracle.pair_set(op))
    op2, on2, _ = oracle.brute_force(verts, vidx, box_filter=False)
    assert on2 == ln and np.array_equal(oracle.pair_set(lp), oracle.pair_set(op2))


def test_golden_pair_sets():
    """Committed fixtures (tests/golden/make_golden.py): sorted pair lists produced by the oracle."""
    g = np.load(os.path.join(GOLD, "cd_golden.npz"))
    for name, gen in (("soup_5k", lambda: synth.soup(5000, 0.06, 42)), ("cloth_30", lambda: synth.cloth_pair(30))):
        verts, vidx = gen()
        with mi355cd.CollisionDetector(verts, vidx) as cd:
            pairs, n, rc = cd.self_collide()
            st = cd.stats()
        assert np.array_equal(oracle.pair_set(pairs), g[name + "_pairs"])
        assert st.pairs_tested == int(g[name + "_tested"])


def test_update_vertices_rebuilds():
    verts, vidx = synth.cloth_pair(40)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        p0, n0, _ = cd.self_collide()
        v2 = verts.copy(); v2[verts.shape[0] // 2:, 1] += 0.5          # lift sheet B clear of sheet A
        cd.update_vertices(v2)
        p1, n1, _ = cd.self_collide()
        assert n0 > 0 and n1 == 0
        cd.update_vertices(verts)
        p2, n2, _ = cd.self_collide()
        assert np.array_equal(oracle.pair_set(p2), oracle.pair_set(p0))


def test_steady_state_steps_clean_their_own_scratch():
    """From the second fused step on there is no memset in front of the pipeline: the sort scratch is left zeroed by the
    previous step's k_build_block, the small counters / traversal counters are zeroed by this step's own kernels
    (ZeroPlan, cd_build.h).  Whatever happens between two steps -- other vertices, stage-wise calls, the brute-force
    checker, another traversal variant, a sort that raises a flag and is redone -- every step must give the oracle's
    pairs and its pairs-tested count."""
    va, ta = synth.soup(60_000, 0.03, 77)
    vb = va + np.random.default_rng(3).normal(0.0, 0.004, va.shape)
    ra, rb = oracle.pipeline(va, ta), oracle.pipeline(vb, ta)
    vo = va + np.array([2.5, 0.0, 0.0])                                    # half of the centroids outside the reference's Morton frame
    ro = oracle.pipeline(vo, ta)

    def step(cd, r):
        pairs, n, rc = cd.self_collide(cap=1 << 20)
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
        assert cd.stats().pairs_tested == r["stats"].pairs_tested

    with mi355cd.CollisionDetector(va, ta) as cd:
        for _ in range(3): step(cd, ra)
        cd.update_vertices(vb)
        for _ in range(2): step(cd, rb)
        cd.morton_sort(); cd.build_hierarchy(); cd.refit_boxes()           # stage-wise calls dirty the scratch their own way
        pairs, n, rc = cd.find_collisions(cap=1 << 20)
        assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(rb["pairs"]))
        for _ in range(2): step(cd, rb)
        bf, nbf, _ = cd.brute_force(True, cap=1 << 20)
        assert np.array_equal(oracle.pair_set(bf), oracle.pair_set(rb["pairs"]))
        step(cd, rb)
        for variant in (0, 1, 3):
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
            step(cd, rb); step(cd, rb)
        cd.update_vertices(vo)                                             # k_morton raises the flag, the step is redone with the other digits
        for _ in range(3): step(cd, ro)
        cd.update_vertices(va)
        for _ in range(2): step(cd, ra)
        cd.build_tree()                                                    # the fused build alone, then a traversal of its own
        pairs, n, rc = cd.find_collisions(cap=1 << 20)
        assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(ra["pairs"]))
        step(cd, ra)


def test_order_hint_is_a_permutation_and_changes_nothing_but_the_order():
    """CD_OPT_ORDER_HINT (default on): the half traversal of a fused call takes its groups of 64 leaves longest-first per XCD, by the
    previous traversal's wave times, which every wave leaves with its TRIANGLES (so the hint survives a mesh that moves and sorts
    differently).  The order must be a permutation of the groups whatever the times were -- this mesh's, the same mesh moved, a mesh
    with other contact curves, a stage-wise traversal in between -- and pairs, pairs tested and every other counter must be what they
    are without it."""
    va, ta = synth.cloth_pair(130)                                          # 67 600 triangles: 1 057 groups, lists of 132 / 133 groups per XCD
    vb = va.copy(); vb[va.shape[0] // 2:, 0] += 0.4                        # sheet B shifted: the contact curves are elsewhere
    vc = va.copy(); h = va.shape[0] // 2
    vc[h:, 0] = 3.0 - va[h:, 0]; vc[h:, 1] = -0.2 - va[h:, 1]             # sheet B mirrored in x and y: where the sheets meet has nothing to do with before
    ra, rb, rc_ = oracle.pipeline(va, ta), oracle.pipeline(vb, ta), oracle.pipeline(vc, ta)

    def step(cd, r):
        pairs, n, rc = cd.self_collide(cap=1 << 21)
        st = cd.stats()
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])) and st.pairs_tested == r["stats"].pairs_tested
        return st

    def check_arrays(cd):
        cost, order, tri = cd.debug_hint(with_tri=True)
        assert np.array_equal(np.sort(order), np.arange(order.shape[0], dtype=np.uint32)) and cost.max() < 32 and tri.max() < 32
        for x in range(8):
            assert np.all(np.diff(cost[order[x::8]].astype(np.int64)) <= 0)   # every XCD's list: the highest score first
        keys, perm = cd.export_keys()
        pad = np.concatenate([tri[perm], np.zeros(order.shape[0] * 64 - perm.shape[0], dtype=np.uint8)]).reshape(-1, 64)
        return cost, pad

    with mi355cd.CollisionDetector(va, ta) as cd:
        assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 0            # nothing built before the first fused build
        st1 = step(cd, ra)
        assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 2            # built by the first step's build from no times at all: the plain order
        cost, pad = check_arrays(cd)
        assert cost.max() == 0 and cd.debug_hint(with_tri=True)[2].min() >= 1 and np.all(pad.max(1)[:-1] == pad.min(1)[:-1])   # ... and the traversal left every triangle its wave's class
        st2 = step(cd, ra); st3 = step(cd, ra)
        assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 1            # the waves of a cloth pair do not all take the same time
        assert st2.node_visits == st1.node_visits == st3.node_visits and st2.candidates == st1.candidates
        cd.update_vertices(vb)                                              # the mesh moved: its triangles carry the times into whatever groups they sort into now
        for _ in range(2):
            step(cd, rb); assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 1
        prev_tri = cd.debug_hint(with_tri=True)[2]
        cd.update_vertices(vc)                                              # a mesh whose contact curves lie elsewhere altogether
        step(cd, rc_)
        cost, pad = check_arrays(cd)                                        # (pad: THIS traversal's classes by sorted position)
        keys, perm = cd.export_keys()
        want = np.concatenate([prev_tri[perm], np.zeros(cost.shape[0] * 64 - perm.shape[0], dtype=np.uint8)]).reshape(-1, 64).max(1)
        assert np.array_equal(cost, want.astype(np.uint32))                 # a group's score = the max of what its triangles brought along
        for _ in range(2): step(cd, rc_)
        cd.morton_sort(); cd.build_hierarchy(); cd.refit_boxes()            # the stage-wise API builds no hint: its traversal runs in the plain order
        assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 0
        pairs, n, rc = cd.find_collisions(cap=1 << 21)
        assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(rc_["pairs"]))
        cd.update_vertices(va)
        step(cd, ra)
        cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 0)                         # off: the plain order, nothing built, nothing remembered
        st_off = step(cd, ra)
        assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 0
        assert st_off.node_visits == st1.node_visits and st_off.candidates == st1.candidates
        cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 1)
        step(cd, ra); step(cd, ra)
        assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 1
        # an order installed from outside (cd_debug_hint_set): any permutation gives the same pairs; a non-permutation is refused
        cd.build_tree()
        g = (ta.shape[0] + 63) // 64
        cd.debug_hint_set(np.random.default_rng(1).permutation(g).astype(np.uint32))
        pairs, n, rc = cd.find_collisions(cap=1 << 21)
        assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(ra["pairs"])) and cd.stats().pairs_tested == ra["stats"].pairs_tested
        with pytest.raises(mi355cd.CdError):
            cd.debug_hint_set(np.zeros(g, dtype=np.uint32))
        for variant in (1, 0, 3):                                           # other traversals neither read nor write it
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant); step(cd, ra)
    # a mesh of less than one group per XCD, and one group exactly
    for nt in (64, 65, 500):
        v, t = synth.soup(nt, 0.2, 9)
        r = oracle.pipeline(v, t)
        with mi355cd.CollisionDetector(v, t) as cd:
            for _ in range(3): step(cd, r)
            assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) in ((0,) if nt <= 512 else (1, 2))   # (one block: no cross kernel, no hint)


def test_order_hint_of_a_large_tree_is_sorted_chunk_by_chunk():
    """A tree of more than 1 M leaves (round 5: k_cross_fused serves every size) with CD_OPT_ORDER_HINT 2: an XCD list of the order hint holds more than
    ORDER_MAX_ITEMS = 2048 groups and is sorted chunk by chunk of 2048 list positions, a workgroup each.  order[] is a permutation, every chunk of every list has its highest score first,
    a chunk holds exactly the groups half_vblock puts there, and pairs / pairs tested are the oracle's, step after step."""
    verts, vidx = synth.cloth_pair(800)                                     # 2 560 000 triangles: 40 000 groups, lists of 5 000 = chunks of 2048 + 2048 + 904
    r = oracle.pipeline(verts, vidx)
    want = oracle.pair_set(r["pairs"])
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        pairs, n, rc = cd.self_collide(cap=1 << 21)
        assert rc == 0 and cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 0                 # by default a tree of this size runs in the plain order: no hint is built
        cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 2)
        plain = None
        for it in range(3):
            pairs, n, rc = cd.self_collide(cap=1 << 21)
            assert rc == 0 and np.array_equal(oracle.pair_set(pairs), want) and cd.stats().pairs_tested == r["stats"].pairs_tested, it
            cost, order, tri = cd.debug_hint(with_tri=True)
            assert np.array_equal(np.sort(order), np.arange(order.shape[0], dtype=np.uint32)) and cost.max() < 32
            if it == 0:
                assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 2 and cost.max() == 0      # no times yet: the plain order (half_vblock's)
                plain = order.copy()
            else:
                assert cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE) == 1 and cost.max() > cost.min()
            for x in range(8):
                lst, base = order[x::8], plain[x::8]
                for c0 in range(0, lst.shape[0], 2048):
                    assert np.all(np.diff(cost[lst[c0:c0 + 2048]].astype(np.int64)) <= 0), (it, x, c0)
                    assert np.array_equal(np.sort(lst[c0:c0 + 2048]), np.sort(base[c0:c0 + 2048])), (it, x, c0)


def test_order_hint_on_random_meshes_that_move():
    """Steps on meshes whose vertices are jittered between steps (cd_update_vertices): every step takes the order hint its fused build made from the
    times the triangles brought along from the step before -- whatever they sort into now -- and must give the oracle's pairs and pairs tested."""
    rng = np.random.default_rng(2024)
    for case in range(6):
        if case % 2 == 0:
            verts, vidx = synth.soup(int(rng.choice([3000, 40_000, 90_000])), float(rng.choice([0.02, 0.05])), int(rng.integers(1 << 30)))
        else:
            verts, vidx = synth.cloth_pair(int(rng.choice([30, 90, 140])))
        with mi355cd.CollisionDetector(verts, vidx) as cd:
            states = []
            for stepno in range(4):
                v = verts if stepno == 0 else np.float32(verts + rng.normal(0.0, 0.003 * stepno, verts.shape)).astype(np.float64)
                cd.update_vertices(v)
                r = oracle.pipeline(v, vidx)
                pairs, n, rc = cd.self_collide(cap=1 << 21)
                assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])) and cd.stats().pairs_tested == r["stats"].pairs_tested, (case, stepno)
                states.append(cd.debug_get(mi355cd.CD_DBG_GET_ORDER_STATE))
            assert all(st in (0, 1, 2) for st in states) and (vidx.shape[0] <= 512 or states[-1] in (1, 2)), (case, states)


def test_cd_main_harness_on_generated_obj(tmp_path):
    """BASELINE config 1, plumbing: the C++ harness (main.cu twin) on a generated OBJ in the reference's dialect."""
    text = synth.grids_obj_text(32)
    path = tmp_path / "grids.obj"
    path.write_text(text)
    exe = os.path.join(ROOT, "gpu-computing-course_amd", "bin", "cd_main")
    out = subprocess.run([exe, str(path), "--brute"], check=True, capture_output=True, text=True, timeout=300).stdout
    verts, vidx = synth.parse_obj_text(text)
    r = oracle.pipeline(verts, vidx)
    got = sorted((int(a), int(b)) for a, b in __import__("re").findall(r"^(\d{7}) - (\d{7})$", out, flags=__import__("re").M))
    want = sorted(map(tuple, r["pairs"].tolist()))
    assert got == want and len(want) > 0
    assert "wrongParentNum = 0" in out
    assert "nullParentnum = 1, wrongBoundCount=0, nullChildCount=0, notInternalCount=0, uninitBoxCount=0" in out
    assert "nullParentnum = 0, nullTriangle=0, notLeafCount=0, illegalBoxCount=0" in out
    assert "illegal triangle vidx num = 0" in out
    assert f"contact count = {len(want)}" in out
    assert f"First morton code: {int(r['keys'][0])}, last morton code: {int(r['keys'][-1])}" in out


@pytest.mark.parametrize("variant", VARIANTS)
def test_cross_rank_pass_two_contexts_one_gpu(variant):
    """The multi-GPU step's device side on ONE GPU: two object shards in two contexts (HipEngine, the product
    engine), root boxes exchanged by hand, cd_pack_queries -> device buffer -> cd_find_collisions_queries on the
    peer.  Union of local + cross pairs must equal the single-tree oracle on the merged mesh, no duplicates."""
    import torch
    import mi355_multi as multi
    dev = torch.device("cuda", 0)
    shards = [synth.cloth_shard(r, 40, overlap=0.10) for r in range(2)]
    engines = [multi.HipEngine(v, t, i, dev, vertex_id_base=vb) for (v, t, i, vb) in shards]
    for e in engines:
        e.cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
    got = []
    roots = []
    for e in engines:
        pairs, n, tested = e.self_collide(1 << 20)
        got.append(pairs)
        roots.append(e.root_box())
    assert multi.boxes_overlap(roots[0], roots[1])
    sent = 0
    for me, peer in ((0, 1), (1, 0)):
        q = engines[me].pack_queries(roots[peer])                  # my leaves overlapping the peer's root
        sent += q.numel() // multi.QUERY_BYTES
        cross, n, tested = engines[peer].find_collisions_queries(q, 1 << 20)
        got.append(cross)
    got = np.concatenate(got, axis=0)
    verts = np.concatenate([s[0] for s in shards]); vidx = np.concatenate([s[1] + np.uint32(s[3]) for s in shards])
    ids = np.concatenate([s[2] for s in shards])
    cen = (verts[vidx[:, 0]] + verts[vidx[:, 1]] + verts[vidx[:, 2]]) / 3
    ref = oracle.pipeline(verts, vidx, ids, off=cen.min(0), span=(cen.max(0) - cen.min(0)) * (1 + 2.0 ** -20))
    gs = oracle.pair_set(got)
    assert len(gs) == len(np.unique(gs))
    assert np.array_equal(gs, oracle.pair_set(ref["pairs"]))
    assert sent > 0 and len(gs) > 0
    for e in engines:
        e.close()


def _cross_pass_by_hand(shards, variant=3):
    """Two (or more) shards in their own contexts, root boxes exchanged by hand, every shard's leaves that overlap a peer's root packed and
    traversed by that peer: returns the union of local + cross pairs and the queries sent."""
    import torch
    import mi355_multi as multi
    dev = torch.device("cuda", 0)
    engines = [multi.HipEngine(v, t, i, dev, vertex_id_base=vb) for (v, t, i, vb) in shards]
    got, roots, sent = [], [], 0
    for e in engines:
        e.cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
        pairs, n, tested = e.self_collide(1 << 21)
        got.append(pairs); roots.append(e.root_box())
    for me in range(len(engines)):
        for peer in range(len(engines)):
            if me == peer or not multi.boxes_overlap(roots[me], roots[peer]):
                continue
            q = engines[me].pack_queries(roots[peer])
            sent += q.numel() // multi.QUERY_BYTES
            cross, n, tested = engines[peer].find_collisions_queries(q, 1 << 21)
            got.append(cross)
    for e in engines:
        e.close()
    return np.concatenate(got, axis=0), sent


@pytest.mark.parametrize("coords", ["float", "double", "double-coarse-cells"])
def test_cross_pass_between_meshes_whose_boxes_tie_exactly(coords):
    """Queries from ANOTHER mesh are not in a mesh's cell table (cd_bvh.h): their comparisons treat an fp32 tie as "maybe" unless both
    sides are fp32 values.  Two contexts hold THE SAME geometry (the second a copy shifted by exactly one grid step along x, other
    IDs and vertex indices): every box face of one coincides with a face of the other -- ties on every level of both trees --
    and the coincident triangles are coplanar contacts.  float: fp32 values (ties are exact touches); double: full doubles;
    double-coarse-cells: the same moved to x + 4096, where an fp32 cell is 4.9e-4 wide and holds many distinct doubles (most cells
    ambiguous).  Union of local + cross pairs == the oracle on the merged mesh, no duplicates."""
    quads = 36
    v0, t0 = synth.cloth_pair(quads, round_f32=(coords == "float"))
    step = (2.94 - 0.06) / quads
    v1 = v0.copy(); v1[:, 0] += step
    if coords == "float":
        v1 = v1.astype(np.float32).astype(np.float64)
    if coords == "double-coarse-cells":
        v0 = v0 + np.array([4096.0, 0.0, 0.0]); v1 = v1 + np.array([4096.0, 0.0, 0.0])
    n = t0.shape[0]
    ids0 = np.arange(n, dtype=np.uint32); ids1 = (np.arange(n, dtype=np.uint32)[::-1] + n).astype(np.uint32)
    shards = [(v0, t0, ids0, 0), (v1, t0.copy(), ids1, v0.shape[0])]
    got, sent = _cross_pass_by_hand(shards)
    verts = np.concatenate([v0, v1]); vidx = np.concatenate([t0, t0 + np.uint32(v0.shape[0])]); ids = np.concatenate([ids0, ids1])
    cen = (verts[vidx[:, 0]] + verts[vidx[:, 1]] + verts[vidx[:, 2]]) / 3
    ref = oracle.pipeline(verts, vidx, ids, off=cen.min(0), span=(cen.max(0) - cen.min(0)) * (1 + 2.0 ** -20))
    gs = oracle.pair_set(got)
    assert len(gs) == len(np.unique(gs)) and np.array_equal(gs, oracle.pair_set(ref["pairs"]))
    assert sent > n and len(gs) > 1000


@pytest.mark.parametrize("kind", ["cloth-shifted", "soup-shifted", "cells-of-many", "tiny-scale"])
def test_meshes_with_many_ambiguous_cells_match_the_oracle(kind):
    """The cell table's hard cases, each against the oracle (pair set, pairs_tested, fused records byte-equal to the stage-wise build's,
    whose internal boxes take the table's answers value by value) and, where small enough, the brute force:
      cloth-shifted   the double cloth at x, y, z + 1024: cells 1.2e-4 wide, nearly every cell holds several distinct doubles;
      soup-shifted    a double soup at + 65536 (cells 7.8e-3: most boxes lie inside ONE cell per axis);
      cells-of-many   vertices snapped to a lattice of 2^-30 around 1.0 (exactly 2^7 doubles per fp32 cell, ties AND sub-cell overlaps);
      tiny-scale      a double soup scaled by 2^-140 (fp32 subnormals; squares of differences near the FP64 underflow)."""
    rng = np.random.default_rng(5)
    if kind == "cloth-shifted":
        verts, vidx = synth.cloth_pair(70, round_f32=False); verts = verts + 1024.0
    elif kind == "soup-shifted":
        verts, vidx = synth.soup(30_000, 0.05, 9); verts = verts * (1.0 + 1e-9) + 65536.0
    elif kind == "cells-of-many":
        verts, vidx = synth.soup(20_000, 0.05, 3)
        verts = 1.0 + np.round(verts * 2.0 ** 10) * 2.0 ** -30                      # a lattice of 2^-30 in [1, 1.003]: fp32 ulp at 1.0 is 2^-23
    else:
        verts, vidx = synth.soup(20_000, 0.05, 4); verts = (verts * (1.0 + 1e-12)) * 2.0 ** -140
    cen = (verts[vidx[:, 0]] + verts[vidx[:, 1]] + verts[vidx[:, 2]]) / 3
    off = cen.min(0); span = (cen.max(0) - off) * (1 + 2.0 ** -20)
    r = oracle.pipeline(verts, vidx, off=off, span=span)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_AUTO)
        visits = {}
        for table in (1, 0, 1):                                     # CD_OPT_CELL_TABLE 0: no table, every inexact coordinate rounded outward (rounds 1-3)
            cd.set_option(mi355cd.CD_OPT_CELL_TABLE, table)
            for variant in VARIANTS:
                cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
                for step in range(2):
                    pairs, n, rc = cd.self_collide(cap=1 << 21)
                    assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])), (table, variant, step)
                    assert cd.stats().pairs_tested == r["stats"].pairs_tested, (table, variant, step)
                if variant == 3:
                    visits[table] = cd.stats().node_visits
        assert visits[1] <= visits[0]                               # the table can only make boxes tighter
        bp, bn, _ = cd.brute_force(True, cap=1 << 21)
        assert np.array_equal(oracle.pair_set(bp), oracle.pair_set(r["pairs"]))
    _assert_fused_records_equal_stagewise(verts, vidx, auto_frame=True)


def test_update_vertices_switches_the_cell_table_on_and_off():
    """The cell table follows the vertices: a context created on fp32-valued vertices (no table), updated to full doubles (table built),
    to doubles in coarse cells, and back -- stream path and graph replay -- gives the oracle's pairs every time."""
    vf, vidx = synth.cloth_pair(90)
    vd, _ = synth.cloth_pair(90, round_f32=False)
    seq = [("float", vf), ("double", vd), ("double+512", vd + 512.0), ("float", vf), ("double", vd)]
    with mi355cd.CollisionDetector(vf, vidx) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_AUTO)
        for graph in (0, 1):
            cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0); cd.set_option(mi355cd.CD_OPT_GRAPH, graph)
            for name, v in seq:
                cd.update_vertices(v)
                cen = (v[vidx[:, 0]] + v[vidx[:, 1]] + v[vidx[:, 2]]) / 3
                r = oracle.pipeline(v, vidx, off=cen.min(0), span=(cen.max(0) - cen.min(0)) * (1 + 2.0 ** -20))
                for step in range(3):
                    pairs, n, rc = cd.self_collide(cap=1 << 20)
                    assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])), (graph, name, step)
                    assert cd.stats().pairs_tested == r["stats"].pairs_tested, (graph, name, step)


def test_config4_topology_eight_shards_on_one_gpu():
    """BASELINE config 4's topology on the one GPU there is: EIGHT object shards (40-quad cloth pairs, neighbours
    overlapping by 10 % along x) in eight contexts -- the middle shards have TWO peers -- root boxes exchanged by hand
    as the all-gather would, every shard packs for every overlapping peer, every peer traverses what it receives.
    Union of local + cross pairs == the single-tree oracle on the merged mesh, no duplicates; every cross pair is
    reported by exactly one side (tri_contact.cuh:81 on external queries)."""
    import torch
    import mi355_multi as multi
    dev = torch.device("cuda", 0)
    W = 8
    shards = [synth.cloth_shard(r, 40, overlap=0.10) for r in range(W)]
    engines = [multi.HipEngine(v, t, i, dev, vertex_id_base=vb) for (v, t, i, vb) in shards]
    got, roots, tested = [], [], 0
    for e in engines:
        pairs, n, t = e.self_collide(1 << 20)
        got.append(pairs); roots.append(e.root_box()); tested += t
    peers = {r: [s for s in range(W) if s != r and multi.boxes_overlap(roots[r], roots[s])] for r in range(W)}
    assert peers[0] == [1] and peers[7] == [6] and all(peers[r] == [r - 1, r + 1] for r in range(1, 7))
    cross_total = 0
    for me in range(W):
        for peer in peers[me]:
            q = engines[me].pack_queries(roots[peer])               # my leaves overlapping the peer's root
            assert q.numel() > 0
            cross, n, t = engines[peer].find_collisions_queries(q, 1 << 20)
            got.append(cross); cross_total += n; tested += t
    got = np.concatenate(got, axis=0)
    verts = np.concatenate([s[0] for s in shards]); vidx = np.concatenate([s[1] + np.uint32(s[3]) for s in shards])
    ids = np.concatenate([s[2] for s in shards])
    cen = (verts[vidx[:, 0]] + verts[vidx[:, 1]] + verts[vidx[:, 2]]) / 3
    ref = oracle.pipeline(verts, vidx, ids, off=cen.min(0), span=(cen.max(0) - cen.min(0)) * (1 + 2.0 ** -20))
    gs = oracle.pair_set(got)
    assert len(gs) == len(np.unique(gs))
    assert np.array_equal(gs, oracle.pair_set(ref["pairs"]))
    assert cross_total > 0
    # a (query, leaf) AABB hit across ranks is met from both sides, like inside one tree: the sharded count is the single tree's
    assert tested == ref["stats"].pairs_tested
    for e in engines:
        e.close()


def test_multi_step_c_abi_one_rank_self_peer():
    """cd_multi_* (the multi-GPU step in C++ over RCCL) on a ONE-rank communicator in CD_MULTI_SELF_PEER mode: the rank
    all-gathers its root, packs its own leaves for itself, exchanges counts and records with itself through
    ncclSend / ncclRecv and traverses them as external queries -- every phase of the step runs.  The cross pass must
    report exactly the local pair set again (q.id < leaf.id, no shared vertex, same SAT) and meet exactly as many
    overlapping (query, leaf) boxes as the local pass (each query meets its own leaf too)."""
    verts, vidx = synth.cloth_pair(60)
    ids = (np.arange(vidx.shape[0], dtype=np.uint32)[::-1] * 2 + 5).astype(np.uint32)
    r = oracle.pipeline(verts, vidx, ids)
    with mi355cd.CollisionDetector(verts, vidx, ids) as cd:
        uid = mi355cd.multi_unique_id()
        assert len(uid) == 128
        with mi355cd.MultiStep(cd, uid, 0, 1, query_cap_per_peer=64, flags=mi355cd.CD_MULTI_SELF_PEER | mi355cd.CD_MULTI_TIMING) as ms:
            for it in range(3):                                      # the first step grows the (deliberately tiny) slabs collectively
                pairs, n, rc, info = ms.step(cap=1 << 20)
                assert rc == 0
                nl = r["stats"].n_pairs
                assert info.world == 1 and info.rank == 0 and info.n_peers == 1
                assert info.local_pairs == nl and info.cross_pairs == nl and n == 2 * nl
                assert info.sent_queries == info.recv_queries == vidx.shape[0]
                assert np.array_equal(oracle.pair_set(pairs[:nl]), oracle.pair_set(r["pairs"]))
                assert np.array_equal(oracle.pair_set(pairs[nl:]), oracle.pair_set(r["pairs"]))
                assert info.pairs_tested == 2 * r["stats"].pairs_tested
                assert info.attempts == (2 if it == 0 else 1) and info.query_cap >= vidx.shape[0]
                assert info.host_syncs == (3 if it == 0 else 2)
                assert info.ms_tree > 0 and info.ms_pack > 0 and info.ms_exchange >= 0 and info.ms_local > 0 and info.ms_cross > 0
        # without the self peer a one-rank step is the plain self-collision
        with mi355cd.MultiStep(cd, mi355cd.multi_unique_id(), 0, 1) as ms:
            pairs, n, rc, info = ms.step(cap=1 << 20)
            assert rc == 0 and n == r["stats"].n_pairs and info.cross_pairs == 0 and info.n_peers == 0 and info.host_syncs == 2
            assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])) and info.pairs_tested == r["stats"].pairs_tested
        # over a communicator the CALLER owns (cd_multi_create_from_comm): made here with RCCL's own ncclCommInitRank
        import ctypes as C
        import torch
        rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

        class NcclId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid2 = NcclId()
        assert rccl.ncclGetUniqueId(C.byref(uid2)) == 0
        comm = C.c_void_p()
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclId, C.c_int]
        assert rccl.ncclCommInitRank(C.byref(comm), 1, uid2, 0) == 0
        with mi355cd.MultiStep(cd, b"", 0, 1, flags=mi355cd.CD_MULTI_SELF_PEER, nccl_comm=comm.value) as ms:
            pairs, n, rc, info = ms.step(cap=1 << 20)
            nl = r["stats"].n_pairs
            assert rc == 0 and info.world == 1 and info.rank == 0 and info.local_pairs == nl and info.cross_pairs == nl
            assert np.array_equal(oracle.pair_set(pairs[nl:]), oracle.pair_set(r["pairs"]))
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        assert rccl.ncclCommDestroy(comm) == 0                       # still the caller's to destroy
        # the context is still usable through the single-GPU entry points
        p2, n2, rc2 = cd.self_collide()
        assert np.array_equal(oracle.pair_set(p2), oracle.pair_set(r["pairs"]))


@pytest.mark.parametrize("quads,qcap,xs", [
    (40, 0, "0,0.9"),                                # two ranks, 10 % overlap
    (30, 64, "0,0.9,1.8"),                           # chain of three, slabs deliberately tiny: the capacity grows COLLECTIVELY in step 1
    (30, 0, "0,0.9,5.0"),                            # rank 2 overlaps nobody: it skips the exchange, the others do not wait for it
    (24, 0, "0,0.2,0.4,0.6"),                        # everybody overlaps everybody: every slab and every receive offset in use
    (24, 0, "0,0.9,1.8,2.7,3.6,4.5,5.4,6.3"),        # BASELINE config 4's topology: eight ranks in a row
    ("12,60,20", 0, "0,0.9,1.8"),                    # shards of UNEQUAL size with the default capacity (nt / 8 + 1024 differs per rank: 1 168 / 4 624 / 1 424):
                                                     # agreed at creation, or ranks 0 and 2 would grow their slabs while rank 1 moves on to send / receive
], ids=["w2", "w3-grow", "w3-isolated", "w4-all-to-all", "w8-chain", "w3-unequal-shards"])
def test_multi_step_world_gt_1_over_the_loopback_transport(quads, qcap, xs):
    """cd_multi_step with world 2 .. 8 on the one GPU there is: ranks are host threads with a context each, and the
    library is pointed (MI355CD_RCCL_LIBRARY) at tests/loopback_rccl, an in-process stand-in for the ten RCCL calls it
    uses -- RCCL itself refuses two ranks on one device.  The union of all ranks' pairs must be the single-tree oracle's
    set without duplicates, the summed pairs_tested the single tree's, peers / counts / capacities consistent on all
    ranks.  Runs in a child process: which RCCL a process uses is decided once."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "multi_loopback_driver.py"), str(quads), str(qcap), "3", xs],
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    res = json.loads(lines[-1])
    assert res["ok"] and p.returncode == 0, res
    W = len(xs.split(","))
    assert res["world"] == W and res["want_pairs"] > 0
    for it, st in enumerate(res["steps"]):
        if it == 0 and qcap:
            assert st["attempts"] == 2, st                       # 64 records per slab cannot hold a 10 % overlap
        if it > 0:
            assert st["attempts"] == 1, st                       # capacity is kept: no second growth
        assert st["host_syncs"] == [st["attempts"] + 1] * W, st    # one per attempt of the first half + one for both traversal passes
        if "5.0" in xs:
            assert st["sent"][2] == st["recv"][2] == 0 and st["cross"][2] == 0
        assert sum(st["cross"]) > 0
        if quads == "12,60,20":
            assert st["query_cap"] == 60 * 60 * 4 // 8 + 1024, st       # the largest rank's default, on every rank ("same_capacity_everywhere" is among the checks)


@pytest.mark.parametrize("world_xs,inject", [("0,0.9", "1:1"), ("0,0.9,1.8,2.7", "2:0"), ("0,0.9,1.8", "0:2")], ids=["w2-rank1-step1", "w4-rank2-step0", "w3-rank0-step2"])
def test_multi_step_fails_on_every_rank_together(world_xs, inject):
    """A rank that fails locally (CD_MULTI_INJECT_FAILURE: before its own pipeline starts) still joins the step's collectives and
    publishes its error in the count matrix: EVERY rank returns from that step -- the rank itself CD_ERR_INJECTED, the others
    CD_ERR_PEER -- before any send / receive is posted, nobody blocks (the driver's threads are joined with a time-out), and the
    steps before and after it give the oracle's pair set."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "multi_loopback_driver.py"), "30", "0", "4", world_xs, inject],
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    res = json.loads(lines[-1])
    assert res["ok"] and p.returncode == 0, res
    step = int(inject.split(":")[1])
    assert res["steps"][step].get("injected") and res["steps"][step]["checks"]["all_ranks_failed_together"]
    good = [st for st in res["steps"] if not st.get("injected")]
    assert len(good) == 3 and all(all(st["checks"].values()) for st in good), res


def test_multi_step_slab_allocation_failure_in_the_growth_round_fails_together_and_recovers():
    """ The per-peer capacity starts too small (64 records), so step 0 has to grow the slabs -- and on rank 1 that allocation
    fails (CD_MULTI_INJECT_ALLOC_FAILURE).  The new buffers are allocated before the old ones are released: rank 1 returns the error,
    rank 0 CD_ERR_PEER, nobody posts a send or a receive; the NEXT step sees slab_cap < qcap on rank 1, allocates again before anything is
    packed, and gives the oracle's pair set with the cross pairs in it -- nothing silently missing, nothing received into a null buffer."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "multi_loopback_driver.py"), "30", "64", "3", "0,0.9", "1:0:alloc"],
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    res = json.loads(lines[-1])
    assert res["ok"] and p.returncode == 0, res
    assert res["steps"][0].get("injected") and res["steps"][0]["checks"]["all_ranks_failed_together"], res
    for st in res["steps"][1:]:
        assert all(st["checks"].values()) and sum(st["cross"]) > 0 and st["query_cap"] > 64, st


@pytest.mark.parametrize("spec,steps", [("1:create", 1), ("0:2:orphan", 3), ("2:1:orphan", 2)], ids=["creation-allocation-fails-on-rank1", "context-destroyed-under-rank0", "context-destroyed-under-rank2-of-3"])
def test_multi_ranks_that_cannot_step_still_join_the_collectives(spec, steps):
    """(round-3 advisor leftovers)  Creation: a rank whose allocations fail still joins the agreement on the per-peer capacity (with 0) -- what it
    needs for that is allocated first -- so every rank's cd_multi_create returns.  Stepping: a rank whose context was destroyed under its cd_multi joins the step's two
    all-gathers with CD_ERR_ORDER in its status word instead of returning at the door: it gets CD_ERR_ORDER, its peers CD_ERR_PEER, nobody waits (the driver's
    threads are joined with a time-out)."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    xs = "0,0.9,1.8" if spec.startswith("2:") else "0,0.9"
    p = subprocess.run([sys.executable, os.path.join(here, "multi_loopback_driver.py"), "24", "0", str(steps), xs, spec], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    res = json.loads(lines[-1])
    assert res["ok"] and p.returncode == 0, res


def _assert_fused_records_equal_stagewise(verts, vidx, ids=None, split_cross_too=True, auto_frame=False):
    got = {}
    # 1: the fused build as it runs by default (round 6: WITHOUT storing qbox[] -- the cross nodes take leaf boxes out of the records, the query boxes compared here
    #    come from k_fill_qbox on request); 2: with k_cross_meta + k_cross_records (CD_DBG_SPLIT_CROSS: the three-launch form of the cross stage, which reads qbox[]);
    # 3: the default build with qbox[] stored by k_build_block (CD_DBG_STORE_QBOX)
    for fused in (1, 0, 2, 3):
        with mi355cd.CollisionDetector(verts, vidx, ids) as cd:
            if auto_frame:
                cd.set_morton_frame(mi355cd.CD_FRAME_AUTO)
            cd.debug_set(mi355cd.CD_DBG_STAGEWISE_BUILD, 0 if fused else 1)
            cd.debug_set(mi355cd.CD_DBG_SPLIT_CROSS, 1 if fused == 2 else 0)
            cd.debug_set(mi355cd.CD_DBG_STORE_QBOX, 1 if fused == 3 else 0)
            cd.build_tree()
            assert cd.debug_get(mi355cd.CD_DBG_GET_TREE_WAS_FUSED) == (1 if fused else 0)      # the build that was asked for is the build that ran
            got[fused] = cd.debug_records() + (cd.root_box(),)
    if split_cross_too:
        _compare_records(vidx.shape[0], got[2], got[0])
    _compare_records(vidx.shape[0], got[1], got[0])
    _compare_records(vidx.shape[0], got[3], got[0])


def _compare_records(n, a, b):
    (rr, rl, qb, root, rbox), (rr0, rl0, qb0, root0, rbox0) = a, b
    assert (n == 1 or root == root0) and np.array_equal(qb, qb0) and np.array_equal(rbox, rbox0)   # (one leaf: no record, no root name)
    used = np.zeros(n, dtype=bool)                          # records are named by split: n - 1 of the n slots are in use
    if n > 1:                                               # (unused slots hold whatever the allocation held: walk from the root)
        frontier = np.array([root0])
        while frontier.size:
            used[frontier] = True
            ch = np.concatenate([rr0[frontier, 6], rl0[frontier, 6]]).view(np.int32)
            frontier = ch[ch >= 0]
    assert used.sum() == max(n - 1, 0)
    for a, b in ((rr, rr0), (rl, rl0)):
        assert np.array_equal(a[used][:, :7], b[used][:, :7])
    # last | CERTAIN flags, first | EXACT flags: the low 30 bits always; bit 30 / 31 only where the left / right child is a leaf
    assert np.array_equal(rr[used][:, 7] & 0x3fffffff, rr0[used][:, 7] & 0x3fffffff)
    assert np.array_equal(rl[used][:, 7] & 0x3fffffff, rl0[used][:, 7] & 0x3fffffff)
    leafL = rl0[used][:, 6].view(np.int32) < 0
    leafR = rr0[used][:, 6].view(np.int32) < 0
    for w, w0 in ((rr, rr0), (rl, rl0)):
        assert np.array_equal((w[used][:, 7] >> 30 & 1)[leafL], (w0[used][:, 7] >> 30 & 1)[leafL])
        assert np.array_equal((w[used][:, 7] >> 31 & 1)[leafR], (w0[used][:, 7] >> 31 & 1)[leafR])
    # EXACT implies CERTAIN, leaf by leaf
    assert not ((rl[used][:, 7] >> 30 & 1)[leafL] & ~(rr[used][:, 7] >> 30 & 1)[leafL]).any()
    assert not ((rl[used][:, 7] >> 31 & 1)[leafR] & ~(rr[used][:, 7] >> 31 & 1)[leafR]).any()


@pytest.mark.parametrize("kind", ["cloth-float", "soup-double", "mixed", "tiny", "one-block", "two-blocks", "three-blocks", "ragged", "duplicates", "long-ranges",
                                  "1024-blocks", "2048-blocks", "4096-blocks", "16384-blocks-three-empty-spans"])
def test_fused_build_writes_the_records_of_the_stagewise_build(kind):
    """The fused build (cd_build.h: hierarchy from adjacent deltas, fp32 segment trees, cross nodes by k_cross_fused -- beyond 2048 blocks
    with the upper levels from k_top_publish / k_top_publish_upper -- or, on request, by k_cross_meta + k_cross_records) against the
    stage-wise one (k_hierarchy + the FP64 refit, key 104): the traversal records must be the same bytes -- child boxes
    rounded outward, child links, range ends, the root's name, and the exact-in-fp32 bits of LEAF children (those of
    internal children are not read by any kernel and not compared) -- and so must the fp32 query boxes."""
    if kind == "cloth-float":
        verts, vidx = synth.cloth_pair(90)
        verts = verts.astype(np.float32).astype(np.float64)
    elif kind == "soup-double":
        verts, vidx = synth.soup(40000, 0.03, 77)
    elif kind == "mixed":
        verts, vidx = synth.soup(30000, 0.04, 78)
        verts[: len(verts) // 2] = verts[: len(verts) // 2].astype(np.float32).astype(np.float64)
    elif kind == "tiny":
        verts, vidx = synth.soup(3, 0.5, 5)
    elif kind == "one-block":
        verts, vidx = synth.soup(512, 0.2, 6)
    elif kind == "two-blocks":                      # the level above the blocks is the root: nothing above it to keep in LDS
        verts, vidx = synth.soup(600, 0.2, 7)
    elif kind == "three-blocks":                    # four block slots, the last one empty
        verts, vidx = synth.soup(1300, 0.15, 8)
    elif kind == "2048-blocks":                     # 1954 blocks: the largest tree k_cross_fused takes, two nodes of the lowest LDS level per thread
        verts, vidx = synth.soup(1_000_000, 0.005, 14)
    elif kind == "duplicates":                      # equal keys: delta falls through to the index tie-break (64 + clz), runs of equal deltas
        v0, t0 = synth.soup(3000, 0.05, 10)
        verts, vidx = v0, np.concatenate([t0, t0, t0[:1500]], axis=0)
    elif kind == "long-ranges":                     # > 65535 leaves under the top nodes: the whole-wave form of the cross queries
        verts, vidx = synth.soup(200_000, 0.01, 11)
    elif kind == "1024-blocks":                     # 586 blocks of 512 leaves: four block boxes per thread in the top levels
        verts, vidx = synth.soup(300_000, 0.01, 12)
    elif kind == "4096-blocks":                     # 2149 blocks: beyond what one workgroup folds -- spans of 2048 blocks, then their roots (k_top_publish, k_top_publish_upper)
        verts, vidx = synth.soup(1_100_000, 0.005, 13)
    elif kind == "16384-blocks-three-empty-spans":  # 9180 blocks in 16384 slots: eight spans of 2048, the fifth partly filled, the last three empty; an XCD list of the order hint in 2 chunks
        verts, vidx = synth.soup(4_700_000, 0.004, 15)
    else:
        verts, vidx = synth.soup(512 * 7 + 1, 0.05, 9)
    _assert_fused_records_equal_stagewise(verts, vidx)


def test_readers_of_the_query_boxes_get_them_on_request():
    """Round 6: the default fused build does not store qbox[] (k_build_block is bound by its stores at size; nothing on the half traversal's path reads them).  Whoever
    does read them afterwards gets them filled on request: the from-the-root descent (variant switched AFTER the build), the packer, external queries, the deep pass."""
    verts, vidx = synth.cloth_pair(70)
    r = oracle.pipeline(verts, vidx)
    want = oracle.pair_set(r["pairs"])
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        pairs, n, rc = cd.self_collide(cap=1 << 18)
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), want) and cd.stats().pairs_tested == r["stats"].pairs_tested
        for variant in (1, 0, 3):
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
            p2, n2, rc2 = cd.find_collisions(cap=1 << 18)
            assert rc2 == 0 and np.array_equal(oracle.pair_set(p2), want) and cd.stats().pairs_tested == r["stats"].pairs_tested, variant
    # the packer and external queries (the mesh against itself: the ID rule lets each pair through once = the same set); no cd_multi attached: nothing told the build
    import torch
    import mi355_multi as multi
    e = multi.HipEngine(verts, vidx, None, torch.device("cuda", 0), frame=mi355cd.CD_FRAME_REFERENCE)
    pairs, n, tested = e.self_collide(1 << 18)
    q = e.pack_queries(e.root_box())                        # every leaf's box overlaps the root's... unless it is flat: at least most of them
    assert q.numel() // multi.QUERY_BYTES > 0.9 * vidx.shape[0]
    p3, n3, t3 = e.find_collisions_queries(q, 1 << 18)
    assert np.array_equal(oracle.pair_set(p3), want)
    e.close()
    # the deep pass behind a half traversal (a comb whose chain overflows a lane's stack)
    off = np.zeros(3); span = np.full(3, 1048576.0)
    cv, ct = _comb([1 << (59 - k) for k in range(60)], big_first=True)
    rc_ = oracle.pipeline(cv, ct, off=off, span=span)
    with mi355cd.CollisionDetector(cv, ct) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_CUSTOM, off, span)
        pairs, n, rc = cd.self_collide()
        assert cd.stats().stack_overflows > 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(rc_["pairs"])) and cd.stats().pairs_tested == rc_["stats"].pairs_tested


def test_device_pair_post_processing():
    """SURVEY 8f row 2 (main.cu:33-45,149-154): sorted pair list and the set of colliding triangle IDs, on the device."""
    verts, vidx = synth.soup(30000, 0.05, 12)
    ids = (np.arange(vidx.shape[0], dtype=np.uint32) * 7 + 3).astype(np.uint32)      # non-trivial IDs
    with mi355cd.CollisionDetector(verts, vidx, ids) as cd:
        pairs, n, rc = cd.self_collide(cap=1 << 20)
        assert rc == 0 and n > 100
        sp, ns, rcs = cd.sorted_pairs(cap=1 << 20)
        tri, nt, rct = cd.collision_triangles()
        assert rcs == 0 and rct == 0 and ns == n
        want = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
        assert np.array_equal(sp, want)
        assert np.array_equal(tri, np.unique(pairs)) and nt == len(np.unique(pairs))
        # truncated traversal -> the post-processing refuses instead of returning a partial set
        cd.find_collisions(cap=10)
        assert cd.sorted_pairs()[2] == mi355cd.CD_OVERFLOW
        # nothing colliding
    v2, t2 = synth.soup(2000, 0.001, 13)
    with mi355cd.CollisionDetector(v2, t2) as cd:
        pairs, n, rc = cd.self_collide()
        assert n == 0
        assert cd.sorted_pairs()[1] == 0 and cd.collision_triangles()[1] == 0


def test_config3_full_size_one_million_cloth_matches_oracle():
    """BASELINE config 3 at its full size (2 x 500x500 quads = 1 000 000 triangles): the pair SET, pairs_tested and the
    whole tree equal the CPU oracle's (the oracle needs about a second here)."""
    verts, vidx = synth.cloth_pair(500)
    assert vidx.shape[0] == 1_000_000
    r = oracle.pipeline(verts, vidx)
    cd, g = _stagewise(verts, vidx)
    _assert_tree_equal(g, r)
    pairs, n, rc = cd.find_collisions(cap=1 << 22)
    st = cd.stats()
    cd.close()
    assert rc == 0 and n == r["stats"].n_pairs > 10_000
    assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
    assert st.pairs_tested == r["stats"].pairs_tested


def test_eight_million_soup_variants_agree():
    """8 M triangles on one GPU (the whole config-4 data volume): the default path (fused build, half traversal), the split fp32
    descent + exact kernel and the lane-private FP64 traversal are independent implementations and must report the same
    pair set and the same pairs_tested -- and, since round 3, the ORACLE's at this size; the sort must leave the oracle's keys
    and permutation."""
    n = 8_000_000
    verts, vidx = synth.soup(n, 0.005, 99)
    sets, tested = [], []
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        for variant in (3, 1, 0):                            # 3: the default path (15 625 blocks: the fused build, its upper levels from k_top_publish / k_top_publish_upper; half traversal)
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
            pairs, cnt, rc = cd.self_collide(cap=1 << 22)
            assert rc == 0 and cnt > 1000
            sets.append(oracle.pair_set(pairs)); tested.append(cd.stats().pairs_tested)
        keys, perm = cd.export_keys()
        assert cd.check_internal().tolist() == [1, 0, 0, 0, 0] and cd.check_leaves().tolist() == [0, 0, 0, 0]
    assert np.array_equal(sets[0], sets[1]) and np.array_equal(sets[0], sets[2]) and tested[0] == tested[1] == tested[2]
    # ... and the oracle itself at this size (a few seconds of CPU): pair set, pairs-tested count, keys and permutation
    r = oracle.pipeline(verts, vidx)
    assert np.array_equal(sets[0], oracle.pair_set(r["pairs"])) and tested[0] == r["stats"].pairs_tested
    assert np.array_equal(keys, r["keys"]) and np.array_equal(perm, r["perm"])
    assert (keys[1:] >= keys[:-1]).all()
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))
    # spot-check 2 000 reported pairs and 2 000 random non-reported neighbours with the oracle's exact test
    p = (sets[0][:: max(1, len(sets[0]) // 2000)])
    pr = np.stack([(p >> np.uint64(32)).astype(np.uint32), (p & np.uint64(0xffffffff)).astype(np.uint32)], axis=1)
    assert oracle.tri_contact_batch(verts, vidx, pr).all()


@pytest.mark.parametrize("n", [512 * 4096, 512 * 4096 + 1, 544 * 4096 - 7, 577 * 4096 + 123])
def test_first_sort_pass_with_chunk_totals_at_its_boundaries(n):
    """Round 5: beyond 512 sort tiles the first onesweep pass takes a tile's offsets from chunk totals (k_tile_chunks: the per-tile digit counts of k_morton summed
    over chunks of 32 tiles) + the rows of its own chunk, instead of every earlier tile's row.  Exactly 512 tiles (the direct sums still), 513 (one chunk of one
    tile behind sixteen full ones), a last chunk that is full but for a ragged tile, a chunk of one tile + a ragged one: keys and permutation are the oracle's
    stable sort in every case."""
    verts, vidx = synth.soup(n, 0.004, 41)
    r_keys, r_perm = oracle.sort_by_key(oracle.centroid_morton(verts, vidx))
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.morton_sort()
        keys, perm = cd.export_keys()
        assert cd.stats().sort_passes == 2
    assert np.array_equal(keys, r_keys) and np.array_equal(perm, r_perm)


def test_window_sort_forms_agree_and_the_small_one_hands_long_runs_to_the_large_one():
    """Round 5: beyond 2 M keys the in-LDS window sort runs windows of 2048 keys in 512-thread workgroups (two a CU) instead of 4096 in 1024 (one a CU).
    Both forms, forced on meshes on either side of that size (CD_DBG_SORT_WINDOWS), give the oracle's keys and permutation; a run of equal top key bits that is
    too long for the small form (3072) but not for the large one (6144) makes the library redo the sort with the large form -- still two global passes."""
    for n, seed in ((300_000, 51), (1_500_000, 52), (9_000, 55)):
        verts, vidx = synth.soup(n, 0.004, seed)
        r_keys, r_perm = oracle.sort_by_key(oracle.centroid_morton(verts, vidx))
        for form in (1, 2, 0):
            with mi355cd.CollisionDetector(verts, vidx) as cd:
                cd.debug_set(mi355cd.CD_DBG_SORT_WINDOWS, form)
                cd.morton_sort()
                keys, perm = cd.export_keys()
                assert cd.stats().sort_passes == 2, (n, form)
            assert np.array_equal(keys, r_keys) and np.array_equal(perm, r_perm), (n, form)
    # 4500 triangles whose centroids share ONE cell of the top 16 key bits (6 bits of x, 5 of y and z: 0.048 x 0.024 x 0.074 of the reference's frame) but are
    # spread over it (their high key HALVES differ: that is another limit, FIX_MAX), inside a 2.2 M soup
    verts, vidx = synth.soup(2_200_000, 0.004, 53)
    rng = np.random.default_rng(54)
    k = 4500
    cell = synth.REF_SPAN / np.array([64.0, 32.0, 32.0])
    c0 = synth.REF_OFF + (np.array([30, 15, 14]) + 0.5) * cell
    cl = (c0 + (rng.random((k, 1, 3)) - 0.5) * 0.6 * cell + (rng.random((k, 3, 3)) - 0.5) * 1e-3).reshape(-1, 3)
    verts[: 3 * k] = cl.astype(np.float32).astype(np.float64)
    keys0 = oracle.centroid_morton(verts, vidx)
    top = np.sort(keys0 >> np.uint64(44))
    runs = np.diff(np.concatenate([[0], np.flatnonzero(np.diff(top)) + 1, [top.shape[0]]]))
    assert 3072 < runs.max() <= 6144, runs.max()
    r_keys, r_perm = oracle.sort_by_key(keys0)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        for _ in range(2):                                                   # (the second call starts with the large form: the context remembers)
            cd.morton_sort()
            keys, perm = cd.export_keys()
            assert cd.stats().sort_passes == 2
            assert np.array_equal(keys, r_keys) and np.array_equal(perm, r_perm)
        r = oracle.pipeline(verts, vidx)
        pairs, npairs, rc = cd.self_collide(cap=1 << 23)
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])) and cd.stats().pairs_tested == r["stats"].pairs_tested


def test_sort_returns_to_its_first_form_when_the_mesh_is_back_inside_the_frame():
    """A centroid outside the reference's Morton frame sets key bits beyond 59, which the default hybrid sort (global digits = bits 44..59) does not cover: the step
    is redone on bits 48..63 and the context stays with that form -- but, since round 5, only for 64 sorts at a time when THAT was the reason (a mesh that leaves the
    frame may come back; bench.py's first moving-mesh leg ran its last 14 frames in the second form): it then tries the first form again.  A run that is too long for
    the first form stays a reason for good.  Pairs and counters are the oracle's all the way."""
    verts, vidx = synth.cloth_pair(90)
    out = verts.copy(); out[verts.shape[0] // 2:, 0] += 0.2                                  # sheet B beyond x = 3.0845: keys beyond 2^60
    r_in, r_out = oracle.pipeline(verts, vidx), oracle.pipeline(out, vidx)
    assert (r_out["keys"] >> np.uint64(60)).max() > 0 and (r_in["keys"] >> np.uint64(60)).max() == 0

    def step(cd, r):
        pairs, n, rc = cd.self_collide(cap=1 << 20)
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])) and cd.stats().pairs_tested == r["stats"].pairs_tested

    with mi355cd.CollisionDetector(verts, vidx) as cd:
        step(cd, r_in); assert cd.debug_get(mi355cd.CD_DBG_GET_SORT_FORM) == 0
        cd.update_vertices(out)
        for _ in range(3):
            step(cd, r_out); assert cd.debug_get(mi355cd.CD_DBG_GET_SORT_FORM) == 1
        keys, perm = cd.export_keys()
        assert np.array_equal(keys, r_out["keys"]) and np.array_equal(perm, r_out["perm"])
        for _ in range(70):                                                          # still outside: every 64th sort tries the first form, fails, and is redone
            step(cd, r_out)
        assert cd.debug_get(mi355cd.CD_DBG_GET_SORT_FORM) == 1
        cd.update_vertices(verts)                                                     # back inside
        forms = []
        for _ in range(70):
            step(cd, r_in); forms.append(cd.debug_get(mi355cd.CD_DBG_GET_SORT_FORM))
        assert forms[0] == 1 and forms[-1] == 0 and forms == sorted(forms, reverse=True)   # one switch back, within 64 sorts
        keys, perm = cd.export_keys()
        assert np.array_equal(keys, r_in["keys"]) and np.array_equal(perm, r_in["perm"])


def test_half_key_sort_equals_full_sort_and_falls_back():
    """CD_OPT_SORT_FULL: the default hybrid (2 global passes on the top 16 key bits + in-LDS sort of run-aligned
    windows + stable fix-up of equal-high-half runs), the half-key form (4 global passes + fix-up) and all 8 passes
    must give the same keys AND permutation (= the oracle's stable sort); a run too long for one form makes the
    library redo the sort with the next one by itself."""
    verts, vidx = synth.soup(200_000, 0.01, 31)
    r_keys, r_perm = oracle.sort_by_key(oracle.centroid_morton(verts, vidx))
    for opt, passes in ((0, 2), (2, 4), (1, 8)):
        with mi355cd.CollisionDetector(verts, vidx) as cd:
            cd.set_option(mi355cd.CD_OPT_SORT_FULL, opt)
            cd.morton_sort()
            keys, perm = cd.export_keys()
            assert cd.stats().sort_passes == passes
            assert np.array_equal(keys, r_keys) and np.array_equal(perm, r_perm)
    # centroids OUTSIDE the reference's Morton frame set key bits 60..62 (morton.h:7 keeps 21 bits per axis), which the default
    # hybrid form (global digits = key bits 44..59) does not sort by: it must notice and redo with the digits at bits 48..63
    ov, ot = synth.soup(50_000, 0.02, 32)
    ov = ov + np.array([2.5, 0.0, 0.0])                         # frame: x in [0.0045, 3.0845)
    ok_, op_ = oracle.sort_by_key(oracle.centroid_morton(ov, ot))
    assert (ok_ >> np.uint64(60)).max() > 0 and (ok_ >> np.uint64(60)).min() == 0
    with mi355cd.CollisionDetector(ov, ot) as cd:
        cd.morton_sort()
        keys, perm = cd.export_keys()
        assert cd.stats().sort_passes == 2 and np.array_equal(keys, ok_) and np.array_equal(perm, op_)
    # the cloth's keys form long runs of equal top-16 bits (hundreds of keys): the window logic of the hybrid form
    cv, ct = synth.cloth_pair(150)
    ck, cp = oracle.sort_by_key(oracle.centroid_morton(cv, ct))
    with mi355cd.CollisionDetector(cv, ct) as cd:
        cd.morton_sort()
        keys, perm = cd.export_keys()
        assert cd.stats().sort_passes == 2 and np.array_equal(keys, ck) and np.array_equal(perm, cp)
    # 12 000 triangles inside ONE cell of the grid of key bits 44..59 (x: 6 bits, y and z: 5 bits of the frame) -- a run longer than
    # any window for both hybrid forms -- but spread over the 4096 cells of the high key half inside it: hybrid -> hybrid -> half-key
    rng0 = np.random.default_rng(8)
    f_off, f_span = np.array([0.004501, -0.476622, -0.381965]), np.array([3.08, 0.76, 2.36])       # morton.h:45,51,57
    cell = f_span / np.array([64.0, 32.0, 32.0])
    c1 = f_off + (np.array([20, 16, 10]) + 0.5) * cell + (rng0.random((12000, 1, 3)) - 0.5) * 0.9 * cell
    v1 = (c1 + (rng0.random((12000, 3, 3)) - 0.5) * 1e-5).reshape(-1, 3)
    t1 = np.arange(36000, dtype=np.uint32).reshape(12000, 3)
    k1 = oracle.centroid_morton(v1, t1)
    _, cnt32 = np.unique(k1 >> np.uint64(32), return_counts=True)
    assert len(np.unique(k1 >> np.uint64(44))) == 1 and cnt32.max() <= 16              # one run too long to window, short equal-high-half runs
    with mi355cd.CollisionDetector(v1, t1) as cd:
        cd.morton_sort()
        keys, perm = cd.export_keys()
        rk1, rp1 = oracle.sort_by_key(k1)
        assert cd.stats().sort_passes == 4 and np.array_equal(keys, rk1) and np.array_equal(perm, rp1)
    # keys that differ ONLY in their low 32 bits: one long run of equal high halves (200 triangles inside one coarse cell)
    rng = np.random.default_rng(5)
    c = np.array([1.0, 0.0, 0.5]) + (rng.random((200, 1, 3)) - 0.5) * 1e-5
    v2 = (c + (rng.random((200, 3, 3)) - 0.5) * 1e-6).reshape(-1, 3)
    t2 = np.arange(600, dtype=np.uint32).reshape(200, 3)
    k2 = oracle.centroid_morton(v2, t2)
    assert len(np.unique(k2 >> np.uint64(32))) < 10 and len(np.unique(k2)) > 100
    with mi355cd.CollisionDetector(v2, t2) as cd:
        pairs, n, rc = cd.self_collide()
        keys, perm = cd.export_keys()
        assert cd.stats().sort_passes == 8                                  # fell back
        rk, rp = oracle.sort_by_key(k2)
        assert np.array_equal(keys, rk) and np.array_equal(perm, rp)
        r = oracle.pipeline(v2, t2)
        assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
        cd.self_collide()
        assert cd.stats().sort_passes == 8                                  # and stays in full mode for this context
    # the same on a FRESH context with thousands of triangles in one cell: the fused call has already enqueued
    # hierarchy / refit / traversal behind the failed half-key sort -- they must stay in bounds and terminate
    # (the traversal kernels skip on the sort's flags), and the redo must give the oracle's result
    c3 = np.array([1.0, 0.0, 0.5]) + (rng.random((4000, 1, 3)) - 0.5) * 1e-5
    v3 = (c3 + (rng.random((4000, 3, 3)) - 0.5) * 1e-6).reshape(-1, 3)
    t3 = np.arange(12000, dtype=np.uint32).reshape(4000, 3)
    r3 = oracle.pipeline(v3, t3)
    for variant in VARIANTS:
        with mi355cd.CollisionDetector(v3, t3) as cd:
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
            pairs, n, rc = cd.self_collide(cap=1 << 22)
            assert rc == 0 and cd.stats().sort_passes == 8
            assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r3["pairs"]))
            assert cd.stats().pairs_tested == r3["stats"].pairs_tested


def test_stage_timing_switch_changes_only_the_timers():
    """CD_OPT_STAGE_TIMING 0 drops the per-stage HIP events (and the fused call zeroes its counters with one memset):
    results, counters and repeated calls must be unaffected; ms_pipeline / ms_descend stay measured."""
    verts, vidx = synth.cloth_pair(60)
    r = oracle.pipeline(verts, vidx)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        for timing in (0, 1, 0):
            cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, timing)
            for _ in range(2):
                pairs, n, rc = cd.self_collide()
                st = cd.stats()
                assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
                assert st.pairs_tested == r["stats"].pairs_tested
                assert st.ms_pipeline > 0 and 0 < st.ms_descend <= st.ms_traverse <= st.ms_pipeline
                assert (st.ms_sort > 0) == bool(timing)
                if timing:
                    assert st.ms_morton + st.ms_sort + st.ms_hierarchy + st.ms_refit + st.ms_traverse <= st.ms_pipeline * 1.001
        # CD_OPT_KERNEL_STAMPS: with stage timing off, a time that was not stamped in THIS call reads 0 (never an earlier call's)
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
        for mask, want in ((2, (0, 1, 0, 0)), (0, (0, 0, 0, 0)), (7, (1, 1, 1, 0)), (15, (1, 1, 1, 1)), (14, (0, 1, 1, 1))):
            cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, mask)
            pairs, n, rc = cd.self_collide()
            st = cd.stats()
            assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])) and st.pairs_tested == r["stats"].pairs_tested
            assert (st.ms_build_block > 0, st.ms_descend > 0, st.ms_exact > 0, st.ms_pipeline > 0) == tuple(bool(w) for w in want), (mask, st.ms_build_block, st.ms_descend, st.ms_exact, st.ms_pipeline)
        # the stage-wise API after a fused call zeroes its own counters again
        cd.morton_sort(); cd.build_hierarchy(); cd.refit_boxes()
        assert cd.check_internal().tolist() == [1, 0, 0, 0, 0]
        pairs, n, rc = cd.find_collisions()
        assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))


@pytest.mark.parametrize("kind", ["double", "mixed", "float"])
def test_exact_leaf_test_with_and_without_fp32_representable_boxes(kind):
    """Boxes whose coordinates are fp32 values are decided exactly by the fp32 descent (flagged candidates skip the
    FP64 box fetch in k_exact); anything else goes through the FP64 test.  Full-precision doubles, float-rounded
    input and a half-and-half mix must all give the oracle's pairs-tested count and pair set -- including meshes,
    whose neighbour boxes touch exactly (strict overlap says no)."""
    rng = np.random.default_rng(99)
    verts, vidx = synth.cloth_pair(40)
    sv, st = synth.soup(3000, 0.05, 17)
    if kind == "double":
        verts = verts + (rng.random(verts.shape) - 0.5) * 1e-9              # no longer fp32 values; mesh neighbours still share vertices
        sv = sv + (rng.random(sv.shape) - 0.5) * 1e-9
    elif kind == "mixed":
        sv = sv + (rng.random(sv.shape) - 0.5) * 1e-9
    assert (verts.astype(np.float32).astype(np.float64) == verts).all() == (kind != "double")
    v = np.concatenate([verts, sv]); t = np.concatenate([vidx, st + verts.shape[0]]).astype(np.uint32)
    r = oracle.pipeline(v, t)
    for variant in VARIANTS:
        with mi355cd.CollisionDetector(v, t) as cd:
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
            pairs, n, rc = cd.self_collide(cap=1 << 22)
            assert rc == 0 and n == r["stats"].n_pairs
            assert cd.stats().pairs_tested == r["stats"].pairs_tested
            assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """bench.py's N > 1 path end to end on the device: two processes (torch.distributed.run) share this GPU, the
    collectives run on gloo with host staging (MI355_DIST_BACKEND=gloo -- RCCL refuses two ranks on one device), every
    other line is the shipped multi-GPU step: local trees, root all-gather, query exchange, cross traversal.  The pair
    total must equal the oracle's on the merged two-object mesh."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    quads = 60
    env = dict(os.environ, MI355_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--quads", str(quads)],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    parts = [synth.cloth_shard(r, quads) for r in range(2)]
    v = np.concatenate([p[0] for p in parts]); ids = np.concatenate([p[2] for p in parts])
    t = np.concatenate([p[1] + (0 if r == 0 else parts[0][0].shape[0]) for r, p in enumerate(parts)]).astype(np.uint32)
    ref = oracle.pipeline(v, t, ids)
    assert line["config"]["colliding_pairs"] == ref["stats"].n_pairs
    assert line["config"]["last_step_rank0"]["peers"] == [1] and line["config"]["last_step_rank0"]["sent_queries"] > 0
    _assert_multi_line_is_gradeable(line, 2)


def _assert_multi_line_is_gradeable(line, world):
    """the N > 1 line carries `roofline` (whole path of the whole job against N x 8 TB/s) and `cpu_baseline` (the oracle on one shard), like the N = 1 line."""
    rf, cb = line["roofline"], line["cpu_baseline"]
    nt = line["config"]["triangles_per_gpu"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 * world and rf["triangles_all_ranks"] == nt * world
    assert abs(rf["achieved"] - 460.0 * nt * world / (line["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"] and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0 < rf["frac"] < 1 and rf["rank0_local_pipeline"]["dominant_kernel"]["avg_launch_ms"] > 0 and all(v > 0 for v in line["kernel_ms"].values())
    assert rf["rank0_local_pipeline"]["total_collision_ms_device"] < line["ms_per_step"] * 1.5
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "pairs_tested/s" and "ONE shard" in cb["sample"]
    assert line["speedup_vs_cpu_1core"] == pytest.approx(line["value"] / cb["value"])


def test_bench_started_plainly_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` WITHOUT a launcher around it (what a driver that does not know about torch.distributed.run does): the script starts
    `python -m torch.distributed.run --nproc-per-node 2 ... bench.py <same arguments>` itself, as a child process and before anything touches the GPU, relays rank 0's
    line and returns the child's exit code (gloo rehearsal backend here: RCCL refuses two ranks on one device).  And --gpus 1 on the multi-GPU code path
    (MI355_BENCH_MULTI_PATH=1: cd_multi_step with a one-rank communicator and no peer) gives the plain line's pairs and says which path ran."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MI355_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--quads", "60"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "launching" in out.stderr
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and line["parity_checked"] is True
    _assert_multi_line_is_gradeable(line, 2)
    # a launch that fails hands back the child's code: a mesh without triangles
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--quads", "0"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert bad.returncode != 0
    # N = 1 on both paths
    env1 = dict(env); env1.pop("MI355_DIST_BACKEND")
    lines = {}
    for name, extra in (("plain", {}), ("multi", {"MI355_BENCH_MULTI_PATH": "1"}), ("fallback", {"MI355_BENCH_MULTI_PATH": "1", "MI355_BENCH_NO_CD_MULTI": "1"})):
        o = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "10", "--quads", "250", "--no-extras", "--no-ray"],
                           capture_output=True, text=True, timeout=600, env=dict(env1, **extra), cwd=root)
        assert o.returncode == 0, o.stdout[-2000:] + o.stderr[-2000:]
        lines[name] = json.loads([l for l in o.stdout.splitlines() if l.startswith("{")][-1])
    assert lines["plain"]["path"] == "cd_self_collide" and lines["multi"]["path"].startswith("cd_multi_step") and "no peer" in lines["multi"]["path"]
    assert lines["plain"]["config"]["colliding_pairs"] == lines["multi"]["config"]["colliding_pairs"] > 0
    assert lines["plain"]["config"]["pairs_tested_per_step"] == lines["multi"]["config"]["pairs_tested_per_step"]
    assert lines["plain"]["parity_checked"] is True and lines["multi"]["parity_checked"] is True
    # a rank on which the library cannot set its step up: every rank drops to the step orchestrated from Python over torch.distributed, and the line says so
    fb = lines["fallback"]
    assert "FALLBACK" in fb["path"] and "multi_fallback" in fb and fb["parity_checked"] is True and fb["backend"].startswith("rccl through torch.distributed")
    assert fb["config"]["colliding_pairs"] == lines["plain"]["config"]["colliding_pairs"] and fb["config"]["pairs_tested_per_step"] == lines["plain"]["config"]["pairs_tested_per_step"]
    assert "roofline" in fb and "cpu_baseline" in fb and fb["value"] > 0
    # the N = 1 point of a scaling curve on the N > 1 code path: the same keys as the N > 1 line, and a value near the plain line's (the multi-GPU step pays two host
    # synchronisations and two one-rank all-gathers a step where the plain step polls a word: what that costs at 250 k triangles is printed, and bounded)
    _assert_multi_line_is_gradeable(lines["multi"], 1)
    assert "roofline" in lines["plain"] and "cpu_baseline" in lines["plain"]
    ratio = lines["multi"]["value"] / lines["plain"]["value"]
    print("N = 1: multi-path value / plain value =", ratio, lines["multi"]["ms_per_step"], lines["plain"]["ms_per_step"])
    assert 0.6 < ratio < 1.05


def test_every_morton_key_equal():
    """All centroids are EXACTLY one point (p1 = c + a, p2 = c + b, p3 = c - a - b with few-bit a, b): one Morton key, so the
    order and the whole tree are the index tie-break's (delta = 64 + clz(i ^ j)), across six 512-leaf blocks -- the cross nodes'
    ranges, splits and the parent rule of k_cross_fused all run on tie-break deltas -- and every triangle contains c: all
    n (n - 1) / 2 pairs collide.  The sort falls back to its full form (one run of equal keys)."""
    n = 2800
    i = np.arange(n, dtype=np.float64)
    c = np.array([1.0, 0.0, 0.5])
    a = np.stack([(i % 97 + 1) * 2.0 ** -12, (i // 97 + 1) * 2.0 ** -13, np.zeros(n)], axis=1)
    b = np.stack([np.zeros(n), (i % 89 + 1) * 2.0 ** -12, (i // 89 + 1) * 2.0 ** -13], axis=1)
    verts = np.stack([c + a, c + b, c - a - b], axis=1).reshape(-1, 3)
    vidx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    assert len(np.unique(oracle.centroid_morton(verts, vidx))) == 1
    _assert_fused_records_equal_stagewise(verts, vidx)
    r = oracle.pipeline(verts, vidx)
    assert r["stats"].n_pairs == n * (n - 1) // 2
    for variant in VARIANTS:
        with mi355cd.CollisionDetector(verts, vidx) as cd:
            cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
            for _ in range(2):
                pairs, npairs, rc = cd.self_collide(cap=1 << 22)
                assert rc == 0 and npairs == r["stats"].n_pairs and cd.stats().pairs_tested == r["stats"].pairs_tested
                assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))
            keys, perm = cd.export_keys()
            assert np.array_equal(keys, r["keys"]) and np.array_equal(perm, r["perm"]) and cd.stats().sort_passes == 8


def test_medium_random_meshes_match_the_oracle():
    """tools/soak_medium.py (seeded): meshes of 5 k .. 700 k triangles -- uniform and clustered soups, cloth pairs, float and
    double coordinates -- two fused steps each against the oracle's pair set, pairs-tested count, keys and permutation.  The
    sizes at which trees have hundreds of blocks, windows of the hybrid sort hold runs of every length and a step reuses the
    scratch its predecessor cleaned."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_medium.py"), "20261004", "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_meshes_property():
    """Seeded random inputs of many shapes (sizes 1..3000, soups of very different density, shared-vertex meshes, exact
    duplicates, zero-area triangles, coordinates that are / are not fp32 values, custom IDs in random order): every
    traversal variant must reproduce the oracle's pair set and pairs-tested count, and the default pipeline its keys,
    permutation and tree."""
    rng = np.random.default_rng(20261003)
    for case in range(120):
        n = int(rng.choice([1, 2, 3, 5, 17, 64, 65, 130, 511, 513, 1200, 3000]))
        kind = case % 4
        if kind == 0:                                                       # soup, density from sparse to very dense
            verts, vidx = synth.soup(n, float(rng.choice([0.01, 0.1, 0.5, 1.5])), int(rng.integers(1 << 30)))
        elif kind == 1:                                                     # mesh with shared vertices (+ a few soup triangles)
            q = max(1, int(np.sqrt(n / 4)))
            verts, vidx = synth.cloth_pair(q)
        elif kind == 2:                                                     # duplicates and degenerate triangles
            verts, vidx = synth.soup(max(1, n // 2), 0.3, int(rng.integers(1 << 30)))
            vidx = np.concatenate([vidx, vidx[: max(1, n // 4)]]).astype(np.uint32)          # exact duplicates (shared vertices -> filtered)
            deg = vidx[:3].copy(); deg[:, 2] = deg[:, 1]                                        # zero-area triangles
            vidx = np.concatenate([vidx, deg]).astype(np.uint32)
        else:                                                               # full-precision doubles
            verts, vidx = synth.soup(n, 0.2, int(rng.integers(1 << 30)))
            verts = verts + (rng.random(verts.shape) - 0.5) * 1e-7
        ids = None
        if case % 3 == 0:
            ids = rng.permutation(vidx.shape[0]).astype(np.uint32) + 7
        r = oracle.pipeline(verts, vidx, ids)
        _assert_fused_records_equal_stagewise(verts, vidx, ids)            # the two builds write the same traversal records
        for variant in VARIANTS:
            with mi355cd.CollisionDetector(verts, vidx, ids) as cd:
                cd.set_option(mi355cd.CD_OPT_TRAVERSAL, variant)
                pairs, npairs, rc = cd.self_collide(cap=1 << 22)
                assert rc == 0 and npairs == r["stats"].n_pairs, (case, variant)
                assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])), (case, variant)
                assert cd.stats().pairs_tested == r["stats"].pairs_tested, (case, variant)
                if variant == 3:
                    keys, perm = cd.export_keys()
                    assert np.array_equal(keys, r["keys"]) and np.array_equal(perm, r["perm"]), case
                    parent, left, right, boxes, bounded = cd.export_tree()
                    assert np.array_equal(left, r["left"]) and np.array_equal(right, r["right"]) and np.array_equal(parent, r["parent"]), case
                    assert np.array_equal(boxes.view(np.uint64), r["boxes"].view(np.uint64)), case


# ---------------------------------------------------------------------------------------------------------------------
# The code path bench.py times, at the sizes BASELINE.json names, against the oracle: cd_self_collide
# with DEFAULT options -- fused build (k_build_block + k_cross_fused), half traversal (k_descend_half + k_exact), hybrid
# sort -- two consecutive steps on one context (the second one runs on the scratch the first one cleaned: no memset),
# then the same with the options bench.py sets for its timed region (no stage events, only the descent's time stamps).
_ORACLE_CACHE = {}


def _baseline_config(name):
    if name not in _ORACLE_CACHE:
        verts, vidx = synth.soup(100_000, 0.02, 1234) if name == "config2-100k-soup" else synth.cloth_pair(500)
        _ORACLE_CACHE[name] = (verts, vidx, oracle.pipeline(verts, vidx))
    return _ORACLE_CACHE[name]


@pytest.mark.parametrize("config", ["config2-100k-soup", "config3-1M-cloth"])
def test_benched_path_default_options_matches_oracle_at_baseline_size(config):
    verts, vidx, r = _baseline_config(config)
    want_set = oracle.pair_set(r["pairs"])
    assert r["stats"].n_pairs > 1000
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        for phase in ("default", "default-2nd-step", "bench-options", "bench-options-2nd-step"):
            if phase == "bench-options":
                cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
                cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 2)
            pairs, n, rc = cd.self_collide(cap=1 << 22)
            st = cd.stats()
            assert rc == 0 and n == r["stats"].n_pairs, (phase, n)
            assert st.pairs_tested == r["stats"].pairs_tested, phase
            assert np.array_equal(oracle.pair_set(pairs), want_set), phase
            assert (pairs[:, 0] < pairs[:, 1]).all()
            assert st.sort_passes == 2 and st.stack_overflows == 0, phase            # hybrid sort, no deep pass: the benched kernels
            keys, perm = cd.export_keys()
            assert np.array_equal(keys, r["keys"]) and np.array_equal(perm, r["perm"]), phase
        # the tree the LAST fused step walked, as the descent reads it: byte-equal to the stage-wise build's records
        fused_records = cd.debug_records() + (cd.root_box(),)
        # ... and on request the reference-shaped tree (k_hierarchy + FP64 refit on the same sorted keys) equals the oracle's
        parent, left, right, boxes, bounded = cd.export_tree()
    assert np.array_equal(left, r["left"]) and np.array_equal(right, r["right"]) and np.array_equal(parent, r["parent"])
    assert np.array_equal(boxes.view(np.uint64), r["boxes"].view(np.uint64)) and (bounded == 2).all()
    with mi355cd.CollisionDetector(verts, vidx) as cd0:
        cd0.debug_set(mi355cd.CD_DBG_STAGEWISE_BUILD, 1)                                  # stage-wise build (proven against the oracle link by link above)
        cd0.build_tree()
        _compare_records(vidx.shape[0], fused_records, cd0.debug_records() + (cd0.root_box(),))


# Reference-COMPILED Morton vectors (tests/golden/morton_ref.npz, made by the reference's morton.h in the build container;
# see tests/test_oracle_pins.py) against the device functions and against the keys of BASELINE config 3.
import morton_inputs as mi  # noqa: E402


def test_device_morton_equals_reference_compiled_vectors():
    ref = np.load(os.path.join(GOLD, "morton_ref.npz"))
    v = mi.expand_inputs()
    assert mi.sha(v) == str(ref["expand_in_sha"])
    assert np.array_equal(mi355cd.expand64_values(v), ref["expand_out"])                      # morton.h:7-29, 131 072 values
    pts = mi.frame_points()
    assert mi.sha(pts) == str(ref["points_in_sha"])
    keys = mi355cd.morton3d_points(pts)                                                        # morton.h:70-89, 1 048 576 points
    assert np.array_equal(keys[:mi.N_FULL], ref["points_keys_head"])
    assert np.array_equal(keys[::mi.SAMPLE_STRIDE], ref["points_keys_sample"])
    assert mi.sha(keys) == str(ref["points_keys_sha"])
    assert mi355cd.morton3d_points(ref["anchor_points"]).tolist() == [384255804010903211, 1008806316530991104]
    # a custom frame equal to the reference's constants is the same function
    assert np.array_equal(mi355cd.morton3d_points(pts[:4096], mi.REF_OFF, mi.REF_SPAN), keys[:4096])


def test_config3_sorted_keys_equal_reference_compiled_keys():
    """The keys cd_self_collide sorts for BASELINE config 3 are the reference's morton3D of the reference's centroids
    (load_obj.h:89-101), all 1 000 000 of them, and the permutation is the stable order of those keys."""
    ref = np.load(os.path.join(GOLD, "morton_ref.npz"))
    cen, verts, vidx = mi.cloth_centroids(500)
    assert mi.sha(verts) == str(ref["cloth_verts_sha"]) and mi.sha(cen) == str(ref["cloth_centroids_sha"])
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        pairs, n, rc = cd.self_collide(cap=1 << 22)
        keys, perm = cd.export_keys()
    assert rc == 0
    assert np.array_equal(keys[::mi.SAMPLE_STRIDE], ref["cloth_sorted_sample"])
    assert mi.sha(keys) == str(ref["cloth_sorted_sha"])
    assert [int(keys[0]), int(keys[-1])] == ref["cloth_first_last"].tolist()
    unsorted = mi355cd.morton3d_points(cen)
    assert mi.sha(unsorted) == str(ref["cloth_keys_sha"])
    assert np.array_equal(unsorted[perm], keys) and np.array_equal(np.sort(perm), np.arange(perm.size, dtype=np.uint32))
    assert int(ref["cloth_distinct"]) == 1_000_000                  # all keys distinct: the permutation is unique


def test_config4_neighbour_pair_at_full_shard_size_over_the_loopback_transport():
    """BASELINE config 4 at its SHARD size on the one GPU there is: one neighbour pair of the row of eight -- two contexts of
    1 000 000 triangles each (cloth_pair(500)), 10 % overlap along x -- stepped by cd_multi_step (world 2) over the loopback
    transport: the union of the two ranks' pairs == the oracle's pair set on the merged 2 M-triangle mesh, no duplicates, and
    the summed pairs_tested == the single tree's.  What config 4 then still lacks is real links between real GPUs."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "multi_loopback_driver.py"), "500", "0", "2", "0,0.9"],
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    res = json.loads(lines[-1])
    assert res["ok"] and p.returncode == 0, res
    assert res["world"] == 2 and res["want_pairs"] > 40_000
    for st in res["steps"]:
        assert all(st["checks"].values()), st
        assert st["attempts"] == 1 and st["host_syncs"] == [2, 2], st
        assert st["sent"][0] > 90_000 and st["sent"][1] > 90_000 and sum(st["cross"]) > 1000, st       # ~101 k overlapping triangles each way


def test_config4_whole_eight_shards_of_one_million_triangles_over_the_loopback_transport():
    """BASELINE config 4 WHOLE on the one GPU there is: eight contexts of 1 000 000 triangles each (cloth_pair(500) shifted along x, 10 %
    overlap between neighbours, global vertex and triangle IDs), stepped twice by cd_multi_step with world 8 over the loopback transport.
    The union of the eight ranks' pairs == the oracle's pair set on the merged 8 M-triangle mesh, no duplicates; the summed pairs_tested ==
    the single tree's; every rank exchanges queries with exactly its neighbours in the row (1 2 2 2 2 2 2 1 peers, ~101 k queries each way);
    one collective attempt, two host synchronisations per rank and step.  What config 4 then still lacks is real links between real GPUs."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    xs = ",".join("%.1f" % (0.9 * r) for r in range(8))
    p = subprocess.run([sys.executable, os.path.join(here, "multi_loopback_driver.py"), "500", "0", "2", xs],
                       capture_output=True, text=True, timeout=1100)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    res = json.loads(lines[-1])
    assert res["ok"] and p.returncode == 0, res
    assert res["world"] == 8 and res["want_peers"] == [1, 2, 2, 2, 2, 2, 2, 1] and res["want_pairs"] > 8 * 20_000
    for st in res["steps"]:
        assert all(st["checks"].values()), st
        assert st["attempts"] == 1 and st["host_syncs"] == [2] * 8, st
        assert st["sent"][0] > 90_000 and st["sent"][7] > 90_000 and all(v > 180_000 for v in st["sent"][1:7]), st     # ~101 k overlapping triangles per neighbour
        assert sum(st["cross"]) > 7 * 1000 and st["got_pairs"] == res["want_pairs"], st


def test_graph_replay_of_the_steady_state_step_gives_the_same_results():
    """CD_OPT_GRAPH: the fused step replayed as one hipGraph launch -- same pair set, same counters as the stream path and the oracle,
    across new vertex positions (same graph), another capacity (captured again), an overflowing capacity, and a mesh whose
    traversal needs the deep pass (handed to the stream path's traversal)."""
    verts, vidx = synth.cloth_pair(120)
    r = oracle.pipeline(verts, vidx)
    want = oracle.pair_set(r["pairs"])
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        cd.set_option(mi355cd.CD_OPT_GRAPH, 1)
        for it in range(6):                                  # step 0 runs on the stream (nothing to replay yet), the rest are replays
            pairs, n, rc = cd.self_collide(cap=1 << 20)
            assert rc == 0 and n == r["stats"].n_pairs and np.array_equal(oracle.pair_set(pairs), want), it
            assert cd.stats().pairs_tested == r["stats"].pairs_tested
            assert cd.stats().traverse_launches == (SHALLOW_LAUNCHES if it == 0 else 0), it         # 0 = the step was one graph launch
        keys, perm = cd.export_keys()
        assert np.array_equal(keys, r["keys"]) and np.array_equal(perm, r["perm"])
        v2 = verts.copy(); v2[:, 1] += 0.003 * np.sin(40.0 * v2[:, 0]); v2 = v2.astype(np.float32).astype(np.float64)
        r2 = oracle.pipeline(v2, vidx)
        cd.update_vertices(v2)
        for it in range(3):
            pairs, n, rc = cd.self_collide(cap=1 << 20)
            assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r2["pairs"])) and cd.stats().pairs_tested == r2["stats"].pairs_tested
            assert cd.stats().traverse_launches == 0                   # (a replay)
            if it == 0:
                # the records the FIRST replay after the vertices moved leaves behind are the stage-wise build's of the new positions, byte for byte:
                # nothing a launch derives from the geometry may be carried over from an earlier replay (the cross kernel's published upper levels
                # and their flag word -- a replay carries the same sequence number every time)
                with mi355cd.CollisionDetector(v2, vidx) as cd0:
                    cd0.debug_set(mi355cd.CD_DBG_STAGEWISE_BUILD, 1); cd0.build_tree()
                    assert cd0.debug_get(mi355cd.CD_DBG_GET_TREE_WAS_FUSED) == 0
                    _compare_records(vidx.shape[0], cd.debug_records() + (cd.root_box(),), cd0.debug_records() + (cd0.root_box(),))
        pairs, n, rc = cd.self_collide(cap=1 << 18)          # another capacity: another capture
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r2["pairs"]))
        pairs, n, rc = cd.self_collide(cap=16)               # too small: the true count comes back with CD_OVERFLOW
        assert rc == mi355cd.CD_OVERFLOW and n == r2["stats"].n_pairs and len(pairs) == 16
        pairs, n, rc = cd.self_collide(cap=1 << 18)
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r2["pairs"]))
        cd.set_option(mi355cd.CD_OPT_GRAPH, 0)
        pairs, n, rc = cd.self_collide(cap=1 << 18)
        assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r2["pairs"]))
    # a comb tree whose chain overflows a lane's stack: the replayed step reports deferred subtrees, the stream path's traversal finishes them
    off = np.zeros(3); span = np.full(3, 1048576.0)
    cv, ct = _comb([1 << (59 - k) for k in range(60)], big_first=True)
    rc_ = oracle.pipeline(cv, ct, off=off, span=span)
    with mi355cd.CollisionDetector(cv, ct) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_CUSTOM, off, span)
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0); cd.set_option(mi355cd.CD_OPT_GRAPH, 1)
        for it in range(4):
            pairs, n, rc = cd.self_collide()
            assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(rc_["pairs"])), it
            assert cd.stats().pairs_tested == rc_["stats"].pairs_tested


def test_graph_replay_of_a_large_tree():
    """CD_OPT_GRAPH on a tree of more than 2048 blocks (round 5): the captured step then also holds k_tile_chunks (chunk totals of the first sort pass),
    the small window form of k_local_sort and k_top_publish / k_top_publish_upper in front of k_cross_fused, whose flag word every replay sets to the same
    sequence number and k_build_block clears in between.  Replays give the oracle's pairs and counters, across new vertex positions too."""
    verts, vidx = synth.cloth_pair(600)                                      # 1 440 000 triangles: 2813 blocks in 4096 slots, 352 sort tiles
    r = oracle.pipeline(verts, vidx)
    want = oracle.pair_set(r["pairs"])
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        cd.set_option(mi355cd.CD_OPT_GRAPH, 1)
        for it in range(5):
            pairs, n, rc = cd.self_collide(cap=1 << 21)
            assert rc == 0 and np.array_equal(oracle.pair_set(pairs), want) and cd.stats().pairs_tested == r["stats"].pairs_tested, it
            assert cd.stats().traverse_launches == (SHALLOW_LAUNCHES if it == 0 else 0), it
        v2 = verts.copy(); v2[:, 1] += 0.002 * np.sin(30.0 * v2[:, 0]); v2 = v2.astype(np.float32).astype(np.float64)
        r2 = oracle.pipeline(v2, vidx)
        cd.update_vertices(v2)
        for it in range(3):
            pairs, n, rc = cd.self_collide(cap=1 << 21)
            assert rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r2["pairs"])) and cd.stats().pairs_tested == r2["stats"].pairs_tested, it
            assert cd.stats().traverse_launches == 0


def test_pinned_host_pair_buffer_receives_the_pairs_directly():
    """cd_alloc_host_pairs: with a pair buffer in pinned host memory from the library, the report kernel writes the pairs straight
    into it (no staging copy); same pairs as through an ordinary buffer -- below and above the 32 768 pairs that travel with the
    report, with the stream path and with the graph replay, and with a capacity below the buffer's."""
    for verts, vidx in (synth.cloth_pair(120), synth.soup(60_000, 0.08, 21)):          # ~1 k pairs; ~90 k pairs (> 32 768)
        r = oracle.pipeline(verts, vidx)
        want = oracle.pair_set(r["pairs"])
        with mi355cd.CollisionDetector(verts, vidx) as cd, mi355cd.HostPairs(1 << 18) as hp:
            plain = np.empty((1 << 18, 2), dtype=np.uint32)
            for graph in (0, 1):
                cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0); cd.set_option(mi355cd.CD_OPT_GRAPH, graph)
                for it in range(3):
                    hp.array[:] = 0xffffffff
                    n, rc = cd.self_collide_into(hp.array)
                    assert rc == 0 and n == len(want) and np.array_equal(oracle.pair_set(hp.array[:n]), want), (graph, it)
                    assert (hp.array[n:n + 8] == 0xffffffff).all()                      # nothing written past the list
                    n2, rc2 = cd.self_collide_into(plain)
                    assert rc2 == 0 and np.array_equal(oracle.pair_set(plain[:n2]), want)
            sub = hp.array[:1 << 12]                                                     # same pointer, smaller capacity
            n, rc = cd.self_collide_into(sub)
            assert n == len(want) and rc == (mi355cd.CD_OVERFLOW if len(want) > (1 << 12) else 0)
            got = oracle.pair_set(sub[:min(n, 1 << 12)])
            assert np.isin(got, want).all() and len(np.unique(got)) == len(got)


def test_polled_completion_gives_what_the_stream_synchronise_gives():
    """CD_OPT_POLL (default on, effective when no time stamp is pending): the host reads the end of a step off the sequence word the
    report kernel stores last, instead of synchronising the stream.  Same pairs, counters and stats either way -- through a pinned
    and an ordinary buffer, over many back-to-back steps (every 64th synchronises anyway), and on the comb whose step needs a deep
    pass, i.e. TWO reports in one step."""
    verts, vidx = synth.cloth_pair(120)
    r = oracle.pipeline(verts, vidx)
    want = oracle.pair_set(r["pairs"])
    with mi355cd.CollisionDetector(verts, vidx) as cd, mi355cd.HostPairs(1 << 16) as hp:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        plain = np.empty((1 << 16, 2), dtype=np.uint32)
        for poll in (1, 0, 1):
            cd.set_option(mi355cd.CD_OPT_POLL, poll)
            for it in range(70 if poll else 3):
                buf = hp.array if it % 2 == 0 else plain
                buf[:] = 0xffffffff
                n, rc = cd.self_collide_into(buf)
                assert rc == 0 and n == len(want) and cd.fast_stats.pairs_tested == r["stats"].pairs_tested, (poll, it)
                assert np.array_equal(oracle.pair_set(buf[:n]), want), (poll, it)
        st = cd.stats()
        assert st.pairs_tested == r["stats"].pairs_tested and st.n_pairs == len(want) and st.ms_descend_clock > 0
        # a capacity below the pair count, polled: the true count comes back with CD_OVERFLOW, what fits is a subset of the pairs, nothing is written past it
        for buf in (hp.array[:64], plain[:64]):
            big = hp.array if buf.base is hp.array or buf is hp.array else plain
            big[:] = 0xffffffff
            n, rc = cd.self_collide_into(buf)
            assert rc == mi355cd.CD_OVERFLOW and n == len(want)
            got = oracle.pair_set(buf[:64])
            assert np.isin(got, want).all() and len(np.unique(got)) == 64 and (big[64:72] == 0xffffffff).all()
    off = np.zeros(3); span = np.full(3, 1048576.0)
    verts, vidx = _comb([1 << (59 - k) for k in range(60)], big_first=True)
    r = oracle.pipeline(verts, vidx, off=off, span=span)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_CUSTOM, off, span)
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for it in range(3):
            pairs, n, rc = cd.self_collide()
            st = cd.stats()
            assert st.stack_overflows > 0 and st.traverse_launches == SHALLOW_LAUNCHES + 2
            assert n == r["stats"].n_pairs > 0 and st.pairs_tested == r["stats"].pairs_tested
            assert np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"]))


def test_polled_completion_never_sees_the_word_before_the_pairs():
    """The hazard of a polled completion: the sequence word overtaking pairs still on their way to host memory.  CD_DBG_POLL_SCAN makes the
    library fill the pair area with 0xff before every step and scan it the moment the word is seen (CD_DBG_GET_POLL_STALE: steps with a pair missing;
    tools/poll_stress.py runs this for 20 000 steps a mesh, with the host link loaded, and has a negative control build that posts the
    word first -- which this scan catches on 96 % of the steps).  A fresh context each, on the mesh (4 948 pairs) round 2's attempt at this
    failed on."""
    for rep in range(3):
        for verts, vidx in (synth.cloth_pair(122), synth.soup(60_000, 0.08, 21)):        # 4 948 pairs; ~29 k pairs (nearly all the report kernel posts)
            with mi355cd.CollisionDetector(verts, vidx) as cd, mi355cd.HostPairs(1 << 17) as hp:
                plain = np.empty((1 << 17, 2), dtype=np.uint32)
                cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0); cd.debug_set(mi355cd.CD_DBG_POLL_SCAN, 1)
                n0, rc = cd.self_collide_into(plain)                                      # (the first two reports into an area end in a stream synchronise: never-written host pages)
                want = oracle.pair_set(plain[:n0].copy())
                for it in range(150):
                    buf = plain if it % 2 == 0 else hp.array
                    n, rc = cd.self_collide_into(buf)
                    assert rc == 0 and n == n0 and np.array_equal(oracle.pair_set(buf[:n]), want), (rep, it)
                assert cd.debug_get(mi355cd.CD_DBG_GET_POLL_STALE) == 0 and cd.debug_get(mi355cd.CD_DBG_GET_POLL_FALLBACKS) == 0 and cd.debug_get(mi355cd.CD_DBG_GET_POLLED_STEPS) > 100


@pytest.mark.parametrize("config", ["cloth1M", "soup100k"])
def test_the_step_bench_py_times_polled_into_a_pinned_buffer_at_baseline_size(config):
    """bench.py's timed step EXACTLY -- CD_OPT_STAGE_TIMING 0, CD_OPT_KERNEL_STAMPS 0, cd_self_collide_into a HostPairs buffer, default
    CD_OPT_POLL -- 80 consecutive steps on BASELINE config 3 (1 M cloth) and config 2 (100 k soup), a fresh context and a fresh pinned
    buffer each, with the library's scan on (the pair area is poisoned before every step and scanned the moment the sequence word is
    seen): every step's pair set and pairs_tested equal the REFERENCE-COMPILED end result, the steps after the warm-up were polled
    (not synchronised), no fall-back to the stream, and no step saw the word before a pair."""
    ref = np.load(os.path.join(GOLD, "contact_ref.npz"))
    verts, vidx = end_mesh(config)
    want = ref[config + "_pairs"]
    with mi355cd.CollisionDetector(verts, vidx) as cd, mi355cd.HostPairs(1 << 22) as hp:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        cd.debug_set(mi355cd.CD_DBG_POLL_SCAN, 1)
        for it in range(80):
            n, rc = cd.self_collide_into(hp.array)
            assert rc == 0 and n == int(ref[config + "_count"]), it
            assert cd.fast_stats.pairs_tested == int(ref[config + "_tested"]), it
            if it < 4 or it % 8 == 0 or it >= 76:                        # (sorting 20 k pairs on the host 80 times is the test's time: the scan covers the rest)
                assert np.array_equal(ci.pair_keys(hp.array[:n]), want), it
        polled, stale, fb = cd.debug_get(mi355cd.CD_DBG_GET_POLLED_STEPS), cd.debug_get(mi355cd.CD_DBG_GET_POLL_STALE), cd.debug_get(mi355cd.CD_DBG_GET_POLL_FALLBACKS)
        assert polled >= 76 and stale == 0 and fb == 0, (polled, stale, fb)
        check_end_result(ref, config, hp.array[:n].copy(), cd.stats().pairs_tested)


def test_a_recycled_pinned_buffer_address_starts_cold_and_reports_of_other_grid_sizes_interleave():
    """ cd_free_host_pairs + cd_alloc_host_pairs may hand the same ADDRESS out again: the library tells buffers apart by serial
    number, so the new buffer's first reports end in a stream synchronise again (CD_DBG_GET_POLLED_STEPS does not move for POLL_WARM steps).
    And k_report's arrival counter starts from zero for every launch: a one-workgroup report (no pairs wanted) followed by a 32-workgroup
    one on the same state, polled, with the scan on."""
    verts, vidx = synth.cloth_pair(122)
    r = oracle.pipeline(verts, vidx)
    want = oracle.pair_set(r["pairs"])
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        cd.debug_set(mi355cd.CD_DBG_POLL_SCAN, 1)
        seen = set()
        for rep in range(6):
            with mi355cd.HostPairs(1 << 14) as hp:
                seen.add(hp.array.ctypes.data)
                p0 = cd.debug_get(mi355cd.CD_DBG_GET_POLLED_STEPS)
                for it in range(2):                                              # POLL_WARM steps into a buffer the device has not written: synchronised
                    n, rc = cd.self_collide_into(hp.array)
                    assert rc == 0 and np.array_equal(oracle.pair_set(hp.array[:n]), want)
                assert cd.debug_get(mi355cd.CD_DBG_GET_POLLED_STEPS) == p0, rep
                for it in range(6):
                    n, rc = cd.self_collide_into(hp.array)
                    assert rc == 0 and np.array_equal(oracle.pair_set(hp.array[:n]), want)
                    nn = C.c_uint64(0)                                           # counters only: k_report with ONE workgroup, polled too
                    rc = cd.lib.cd_self_collide(cd._ctx, None, 0, C.byref(nn))
                    assert rc == mi355cd.CD_OVERFLOW and nn.value == len(want)
                assert cd.debug_get(mi355cd.CD_DBG_GET_POLLED_STEPS) >= p0 + 6
        assert cd.debug_get(mi355cd.CD_DBG_GET_POLL_STALE) == 0 and cd.debug_get(mi355cd.CD_DBG_GET_POLL_FALLBACKS) == 0
        assert len(seen) < 6                                                     # (the allocator did hand an address out again: the case this test is about)


def test_multi_step_box_of_the_triangles_ignores_unreferenced_vertices():
    """The box a rank publishes is the box of its TRIANGLES.  When every vertex belongs to a triangle the step takes it from the
    vertices in one streaming launch (k_vertex_box); vertices no triangle uses -- here far outside, so that they would blow the box
    up -- switch it to the triangle-wise reduction.  Both must give the box node 0 of the tree holds (FP64 bit patterns) and the
    oracle's pairs."""
    verts, vidx = synth.cloth_pair(50)
    r = oracle.pipeline(verts, vidx)
    stray = np.array([[1e6, -1e6, 1e6], [-1e6, 1e6, -1e6]])
    for v in (verts, np.concatenate([verts, stray], axis=0)):
        with mi355cd.CollisionDetector(v, vidx) as cd:
            with mi355cd.MultiStep(cd, mi355cd.multi_unique_id(), 0, 1, flags=mi355cd.CD_MULTI_SELF_PEER) as ms:
                for it in range(2):
                    pairs, n, rc, info = ms.step(cap=1 << 20)
                    nl = r["stats"].n_pairs
                    assert rc == 0 and info.local_pairs == nl and info.cross_pairs == nl
                    assert np.array_equal(oracle.pair_set(pairs[:nl]), oracle.pair_set(r["pairs"]))
                    assert np.array_equal(cd.root_box().view(np.uint64), r["boxes"][0].view(np.uint64))     # what the step published == node 0's box


# Reference-COMPILED exact-test vectors and END RESULTS (tests/golden/contact_ref.npz, made by the reference's tri_contact.cuh /
# box.cuh / triangle.cuh / vec3f.cuh compiled unmodified in the build container; see tests/test_oracle_pins.py) against the
# device functions and against cd_self_collide with default options on BASELINE configs 2 and 3.
import contact_inputs as ci  # noqa: E402
from test_oracle_pins import check_end_result, end_mesh, _unbits  # noqa: E402


def test_device_exact_test_equals_reference_compiled_vectors():
    ref = np.load(os.path.join(GOLD, "contact_ref.npz"))
    tri, fam = ci.tri_pairs()
    n = tri.shape[0]
    assert n >= 1_000_000 and ci.sha(tri) == str(ref["tri_in_sha"])
    want = _unbits(ref["tri_contact_bits"], n)
    got = mi355cd.tri_contact_points(tri)                                               # tri_contact.cuh:19-78, 1 179 648 pairs
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, (bad[:8], fam[bad[:8]])
    # the same pairs through cd_test_pairs (the vertex fetch + ID rule + neighbour gate around it): triangles 2k, 2k+1 of one context
    verts = tri.reshape(-1, 3)
    vidx = np.arange(6 * n, dtype=np.uint32).reshape(2 * n, 3)
    pairs = np.arange(2 * n, dtype=np.uint32).reshape(n, 2)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        assert np.array_equal(cd.test_pairs(pairs).astype(np.int32), want)
        assert not cd.test_pairs(pairs[:4096, ::-1].copy()).any()                       # ID rule, tri_contact.cuh:81
    # indexed pairs: checkTriangleContactHelper + Triangle::neighborCount, IDs equal / reversed / ordered
    pv, va, ida, vb, idb = ci.indexed_pairs()
    m = va.shape[0]
    helper = _unbits(ref["helper_bits"], m); nc = ref["neighbor"].astype(np.int32)
    with mi355cd.CollisionDetector(pv, np.concatenate([va, vb]), np.concatenate([ida, idb])) as cd:
        got = cd.test_pairs(np.stack([np.arange(m), np.arange(m) + m], axis=1).astype(np.uint32)).astype(np.int32)
    assert np.array_equal(got, ((nc < 1) & (helper > 0)).astype(np.int32))              # collision.cuh:36-37
    assert 1000 < int(got.sum()) < int(helper.sum())                                    # the neighbour gate removes some of the helper's hits


def test_exact_test_with_hardware_min_max_falls_back_where_a_projection_is_not_a_number():
    """k_exact's SAT takes max / min of an axis' projections with v_max_f64 / v_min_f64 (cd_math.h, tri_contact_fast) -- the compare-selects of mathop.cuh:17-44
    whenever no projection is a NaN, up to the sign of a zero -- and returns to tri_contact itself for a pair that has one.  65 536 pairs whose coordinates are NaN,
    +-inf, +-0, huge (products overflow: inf - inf), tiny (products underflow) or ordinary, in every mixture (contact_inputs.nonfinite_pairs): the verdicts of the
    REFERENCE's tri_contact.cuh compiled unmodified (tests/golden/contact_nonfinite_ref.npz), through the points kernel and through cd_test_pairs."""
    ref = np.load(os.path.join(GOLD, "contact_nonfinite_ref.npz"))
    tri = ci.nonfinite_pairs()
    n = tri.shape[0]
    assert ci.sha(tri) == str(ref["tri_in_sha"]) and int(np.isnan(tri).any(axis=(1, 2)).sum()) == int(ref["pairs_with_a_nan"]) > n // 8
    want = _unbits(ref["tri_contact_bits"], n)
    assert int(want.sum()) == int(ref["contacts"]) > 1000
    got = mi355cd.tri_contact_points(tri)                                               # tri_contact_fast: what k_exact runs
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, (bad[:8], tri[bad[:2]])
    assert np.array_equal(oracle.tri_contact_points(tri), want)                        # (and the CPU restatement, on the same box)


def test_device_boxes_equal_reference_compiled_vectors():
    ref = np.load(os.path.join(GOLD, "contact_ref.npz"))
    a, b = ci.box_pairs()
    assert ci.sha(np.concatenate([a, b])) == str(ref["box_in_sha"])
    ov, _ = mi355cd.box_pairs(a, b, want_merged=False)                                  # box.cuh:40-43, 1 048 576 pairs
    assert np.array_equal(ov.astype(np.int32), _unbits(ref["box_overlap_bits"], a.shape[0]))
    # Box::set as the leaf boxes of a built tree, Box::merge on the reference's operands
    pv, va, ida, vb, idb = ci.indexed_pairs()
    m = va.shape[0]
    both = np.concatenate([va, vb])
    with mi355cd.CollisionDetector(pv, both) as cd:
        cd.set_morton_frame(mi355cd.CD_FRAME_AUTO, None, None)
        cd.build_tree()
        keys, perm = cd.export_keys()
        parent, left, right, boxes, bounded = cd.export_tree()
    leaf = np.zeros((2 * m, 6), dtype=np.float64)
    leaf[perm] = boxes[2 * m - 1:]
    assert ci.sha(leaf) == str(ref["box_set_sha"]) and np.array_equal(leaf[:4096].view(np.uint64), ref["box_set_head"].view(np.uint64))
    _, mg = mi355cd.box_pairs(leaf[:m], leaf[m:])
    assert ci.sha(mg) == str(ref["box_merge_sha"]) and np.array_equal(mg[:4096].view(np.uint64), ref["box_merge_head"].view(np.uint64))


@pytest.mark.parametrize("name", ["soup100k", "cloth1M", "soup1M", "cloth1M_double", "soup100k_mt64", "soup1M_mt64"])
def test_self_collide_default_options_equals_reference_compiled_end_result(name):
    """cd_self_collide with the library's default options, two consecutive steps: the pair set (SHA-256 of the sorted keys, every
    64th key, the full list where the fixture holds it) and pairs_tested equal what the REFERENCE'S compiled predicates give for
    BASELINE config 2 (plain O(N^2) on the reference side), config 3, the 1 M soup and config 3 with full-double vertices."""
    ref = np.load(os.path.join(GOLD, "contact_ref.npz"))
    verts, vidx = end_mesh(name)
    assert ci.sha(verts) == str(ref[name + "_verts_sha"]) and ci.sha(vidx) == str(ref[name + "_vidx_sha"])
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        for step in range(2):
            pairs, n, rc = cd.self_collide(cap=1 << 22)
            assert rc == 0
            check_end_result(ref, name, pairs, cd.stats().pairs_tested)
        if name == "soup100k":                                                         # and the device's own all-pairs pass (check.cuh:117-141)
            bp, bn, _ = cd.brute_force(box_filter=True)
            check_end_result(ref, name, bp, cd.stats().pairs_tested)
