"""The synthetic-input generators (gpu-computing-course_amd/pyhost/mi355_synth.py): host code, no GPU."""
import hashlib

import numpy as np

import mi355_synth as synth


def test_mt19937_64_known_answers():
    """std::mt19937_64 in numpy: [rand.predef]'s known answer (the 10000th output of a default-seeded engine), the first outputs of seed 5489 as
    the reference implementation of Matsumoto & Nishimura prints them for init_genrand64(5489), and block boundaries (312-word state)."""
    g = synth.MT19937_64()
    x = g.raw(10000)
    assert int(x[9999]) == 9981545732273789042
    assert [int(v) for v in x[:3]] == [14514284786278117030, 4620546740167642908, 13109570281517897720]
    a = synth.MT19937_64(1234).raw(1000)
    b = synth.MT19937_64(1234)
    assert np.array_equal(a[:312], b.raw(312)) and np.array_equal(a[312:624], b.raw(312))        # (whole blocks: one call or two)


def test_soup_mt64_is_the_recipe_of_the_survey():
    """mt19937_64(seed = 1234), centroids uniform in the generation box, three vertices = centroid + U(-e/2, e/2)^3, rounded to float,
    V = 3 N.  Bit-equal to a g++ / libstdc++ program drawing with std::uniform_real_distribution in this order (checked when the generator was written;
    the hash pins it).

    The survey's own recorded answers for this recipe are reproduced only in part, and this file is where that is said: this draw order gives the
    survey's 1 326 colliding pairs at 100 k but 117 666 pairs tested where the survey recorded 117 850, and 16 992 pairs / 1 224 320 tested at 1 M (e = 0.01) where it
    recorded 16 795 / 1 222 266.  The survey kept the DESCRIPTION of its generator, not the code.  What other readings of that description give at 100 k (the oracle,
    round 6): centroid and its nine offsets per triangle 1 263 / 117 380; all centroids first with the offsets axis-major 1 340 / 117 638; draws from the top 53 bits
    the same 1 326 / 117 666; this order with e = 0.0201 1 346 / 117 952; numpy's generator (synth.soup, the bench's config-2 mesh) 1 315 / 117 802.  None gives both of
    the survey's counts, so it cannot be said WHICH draw differs; what can be said is that the survey's pair count is this order's and that its tested count is not
    reached by a slightly larger `e` without moving the pairs too.  The committed vectors (tests/golden/, the reference-compiled end results of
    tests/test_oracle_pins.py) are for THIS order; nothing is compared against the survey's four numbers."""
    v, t = synth.soup_mt64(100_000, 0.02, 1234)
    assert v.shape == (300_000, 3) and t.shape == (100_000, 3) and np.array_equal(t.ravel(), np.arange(300_000, dtype=np.uint32))
    assert np.array_equal(v, v.astype(np.float32).astype(np.float64))
    c = v.reshape(100_000, 3, 3).mean(1)
    assert (c > synth.BOX_LO - 0.011).all() and (c < synth.BOX_HI + 0.011).all()
    assert np.abs(v.reshape(100_000, 3, 3) - c[:, None, :]).max() < 0.02
    assert hashlib.sha256(v.tobytes()).hexdigest()[:16] == SOUP_MT64_SHA16


SOUP_MT64_SHA16 = "ed8fd5983c635885"


def test_config4_merged_is_the_concatenation_of_the_shards():
    """BASELINE config 4 as ONE mesh (bench.py's config4_merged_8M, at a small size here): the shards of cloth_shard(r) concatenated with global vertex indices and
    triangle IDs, neighbours overlapping by 10 % of their width along x, and the Morton frame of the merged centroids (off = min, span = (max - min)(1 + 2^-20))."""
    world, quads = 4, 20
    verts, vidx, ids, off, span = synth.config4_merged(world, quads)
    nt = 2 * 2 * quads * quads
    assert vidx.shape == (world * nt, 3) and np.array_equal(ids, np.arange(world * nt, dtype=np.uint32))
    nv = verts.shape[0] // world
    for r in range(world):
        v, t, i, vb = synth.cloth_shard(r, quads)
        assert vb == r * nv and np.array_equal(verts[r * nv:(r + 1) * nv], v)
        assert np.array_equal(vidx[r * nt:(r + 1) * nt], t + np.uint32(vb)) and np.array_equal(ids[r * nt:(r + 1) * nt], i)
    x0 = [verts[r * nv:(r + 1) * nv, 0].min() for r in range(world)]; x1 = [verts[r * nv:(r + 1) * nv, 0].max() for r in range(world)]
    for r in range(world - 1):
        assert x0[r + 1] < x1[r] and abs((x1[r] - x0[r + 1]) / (x1[r] - x0[r]) - 0.10) < 0.01           # 10 % overlap between neighbours
    cen = (verts[vidx[:, 0]] + verts[vidx[:, 1]] + verts[vidx[:, 2]]) / 3
    assert np.array_equal(off, cen.min(0)) and np.allclose(span, (cen.max(0) - cen.min(0)) * (1 + 2.0 ** -20), rtol=0, atol=0)
    e = (cen - off) / span * 1048576.0
    assert e.min() >= 0 and e.max() < 1048576.0                             # every centroid inside the frame: keys below 2^60
