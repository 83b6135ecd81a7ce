"""One-off soak: medium-size random meshes (tens to hundreds of thousands of triangles, uniform and clustered, float and double
coordinates) -- the fused step against the oracle (pair set, pairs tested, keys, permutation) on ONE context driven over all of
them is not possible (sizes differ), so a context each; prints a line per case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost"), os.path.join(ROOT, "tests")]
import numpy as np, mi355cd, mi355_synth as synth, oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 777)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for case in range(ncase):
    n = int(rng.choice([5000, 20000, 60000, 150000, 300000, 700000]))
    kind = case % 4
    if kind == 0:
        verts, vidx = synth.soup(n, float(rng.choice([0.005, 0.02, 0.05])), int(rng.integers(1 << 30)))
    elif kind == 1:
        verts, vidx = synth.cloth_pair(max(2, int(np.sqrt(n / 4))))
    elif kind == 2:                                   # clustered (size capped: the oracle's time grows with the pairs)
        n = min(n, 150000)
        # clustered: most triangles in a few small blobs (long runs of equal top key bits), the rest spread
        verts, vidx = synth.soup(n, 0.01, int(rng.integers(1 << 30)))
        tri = verts.reshape(-1, 3, 3)
        nb = int(rng.integers(1, 6)); centres = rng.random((nb, 3)) * np.array([2.5, 0.5, 2.0]) + np.array([0.2, -0.4, -0.3])
        m = rng.random(n) < 0.5
        which = rng.integers(0, nb, n)
        c = tri.mean(axis=1)
        scale = float(rng.choice([0.15, 0.3, 0.6]))
        newc = centres[which] + (rng.random((n, 3)) - 0.5) * scale
        tri[m] += (newc - c)[m][:, None, :]
        verts = tri.reshape(-1, 3)
    else:
        verts, vidx = synth.soup(n, 0.03, int(rng.integers(1 << 30)))
        verts = verts + (rng.random(verts.shape) - 0.5) * 1e-7
    if case % 3 == 1:
        verts = verts.astype(np.float32).astype(np.float64)
    t0 = time.time()
    r = oracle.pipeline(verts, vidx)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        ok = True
        if os.environ.get("SOAK_NO_STAMPS"):          # the bench's options: no time stamps, so the polled completion (CD_OPT_POLL) is what ends a step
            cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for rep in range(6 if os.environ.get("SOAK_NO_STAMPS") else 2):   # second step: the self-cleaning scratch
            pairs, npairs, rc = cd.self_collide(cap=1 << 23)
            ok &= rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(r["pairs"])) and cd.stats().pairs_tested == r["stats"].pairs_tested
        keys, perm = cd.export_keys()
        ok &= np.array_equal(keys, r["keys"]) and np.array_equal(perm, r["perm"])
        passes = cd.stats().sort_passes
    print(f"case {case} kind {kind} n {vidx.shape[0]} pairs {r['stats'].n_pairs} tested {r['stats'].pairs_tested} sort passes {passes}: {'ok' if ok else 'MISMATCH'}  ({time.time()-t0:.1f} s)", flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
