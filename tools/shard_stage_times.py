#!/usr/bin/env python3
"""Stage times of ONE config-4 shard (CD_FRAME_AUTO) next to the config-3 mesh (reference frame): which sort form runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd
cases = [("config3 reference frame", synth.cloth_pair(500), None, mi355cd.CD_FRAME_REFERENCE)]
for r in (0, 3):
    v, t, ids, vb = synth.cloth_shard(r, 500)
    cases.append((f"shard {r} auto frame", (v, t), ids, mi355cd.CD_FRAME_AUTO))
for name, (v, t), ids, frame in cases:
    with mi355cd.CollisionDetector(v, t, ids) as cd:
        cd.set_morton_frame(frame)
        for _ in range(3): cd.self_collide()
        s = cd.stats()
        print(f"{name}: morton {s.ms_morton*1e3:.0f} sort {s.ms_sort*1e3:.0f} (passes {s.sort_passes}) hierarchy {s.ms_hierarchy*1e3:.0f} refit {s.ms_refit*1e3:.0f} traverse {s.ms_traverse*1e3:.0f} us  pairs {s.n_pairs} tested {s.pairs_tested}")
