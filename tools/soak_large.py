#!/usr/bin/env python3
"""Seeded meshes around the size thresholds of round 5 -- 2048 blocks (k_top_publish), 512 sort tiles (k_tile_chunks), 1.31 M keys (the small window form of
k_local_sort) -- cloths and soups, float and full-double vertices, two steps each with cd_update_vertices in between: pair set and pairs_tested against the oracle.
usage: soak_large.py [CASES]   GPU only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost"), os.path.join(ROOT, "tests")]
import numpy as np, mi355cd, mi355_synth as synth, oracle
rng = np.random.default_rng(2025)
sizes = [1_048_576 + 3, 1_048_576 + 512 * 7 + 1, 1_310_720, 1_310_721, 1_400_000, 2_097_152, 2_097_153, 2_101_248 + 5, 2_500_000, 3_300_000, 4_194_305]
cases = int(sys.argv[1]) if len(sys.argv) > 1 else len(sizes)
bad = 0
for k, n in enumerate(sizes[:cases]):
    kind = k % 3
    if kind == 0:
        verts, vidx = synth.soup(n, float(rng.choice([0.003, 0.005])), int(rng.integers(1 << 30)))
    elif kind == 1:
        q = int(round((n / 4) ** 0.5)); verts, vidx = synth.cloth_pair(q)
    else:
        q = int(round((n / 4) ** 0.5)); verts, vidx = synth.cloth_pair(q, round_f32=False)
    t0 = time.time()
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        for step in range(2):
            v = verts if step == 0 else verts + rng.normal(0.0, 0.0005, verts.shape)
            if kind != 2: v = v.astype(np.float32).astype(np.float64)
            if step: cd.update_vertices(v)
            want, st, _ = oracle.self_collide(v, vidx, cap=1 << 23)
            pairs, npairs, rc = cd.self_collide(cap=1 << 23)
            ok = rc == 0 and np.array_equal(oracle.pair_set(pairs), oracle.pair_set(want)) and cd.stats().pairs_tested == st.pairs_tested
            bad += 0 if ok else 1
            print(f"case {k} kind {kind} n {vidx.shape[0]} step {step}: pairs {npairs} tested {cd.stats().pairs_tested} sort passes {cd.stats().sort_passes}: {'ok' if ok else 'MISMATCH'}  ({time.time() - t0:.1f} s)", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
