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
    the hash pins it)."""
    v, t = synth.soup_mt64(100_000, 0.02, 1234)
    assert v.shape == (300_000, 3) and t.shape == (100_000, 3) and np.array_equal(t.ravel(), np.arange(300_000, dtype=np.uint32))
    assert np.array_equal(v, v.astype(np.float32).astype(np.float64))
    c = v.reshape(100_000, 3, 3).mean(1)
    assert (c > synth.BOX_LO - 0.011).all() and (c < synth.BOX_HI + 0.011).all()
    assert np.abs(v.reshape(100_000, 3, 3) - c[:, None, :]).max() < 0.02
    assert hashlib.sha256(v.tobytes()).hexdigest()[:16] == SOUP_MT64_SHA16


SOUP_MT64_SHA16 = "ed8fd5983c635885"
