#!/usr/bin/env python3
"""As ab_option.py, plus the per-kernel device times of each value (a few extra steps with every kernel stamp on: build_block, descent, exact, pipeline).
usage: ab_option_kernels.py KEY V0 V1 [ROUNDS STEPS QUADS]   (KEY = dbg:N: a cd_debug_option key)   GPU only."""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import numpy as np
import mi355_synth as synth, mi355cd
dbg = sys.argv[1].startswith("dbg:")          # KEY = dbg:N addresses a cd_debug_option key
key, v0, v1 = int(sys.argv[1].split(":")[-1]), int(sys.argv[2]), int(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 100
quads = int(sys.argv[6]) if len(sys.argv) > 6 else 500
verts, vidx = synth.cloth_pair(quads)
buf = np.empty((1 << 22, 2), dtype=np.uint32)
with mi355cd.CollisionDetector(verts, vidx) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for _ in range(30): cd.self_collide_into(buf)
    wall = {v0: [], v1: []}; res = {}; kern = {v0: [], v1: []}
    for r in range(rounds):
        for v in ((v0, v1) if r % 2 == 0 else (v1, v0)):
            (cd.debug_set if dbg else cd.set_option)(key, v)
            for _ in range(5): cd.self_collide_into(buf)
            t0 = time.perf_counter()
            for _ in range(steps): n, rc = cd.self_collide_into(buf)
            wall[v].append((time.perf_counter() - t0) * 1e6 / steps)
            res[v] = (n, cd.fast_stats.pairs_tested, cd.fast_stats.ms_descend_clock)
            cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 15)
            for _ in range(10):
                cd.self_collide_into(buf); st = cd.fast_stats
                kern[v].append((st.ms_build_block * 1e3, st.ms_descend * 1e3, st.ms_exact * 1e3, st.ms_pipeline * 1e3))
            cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for v in (v0, v1):
        k = [statistics.median(x[i] for x in kern[v]) for i in range(4)]
        print(f"key {key} = {v}: wall per step median {statistics.median(wall[v]):7.2f} us  min {min(wall[v]):7.2f} us   pairs {res[v][0]} tested {res[v][1]} descent (device clock) {res[v][2]*1e3:.1f} us"
              f" | stamped: build_block {k[0]:.1f}  descent {k[1]:.1f}  exact {k[2]:.1f}  pipeline {k[3]:.1f} us")
