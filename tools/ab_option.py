#!/usr/bin/env python3
"""A/B of one cd_set_option key in ONE process on the bench workload: blocks of STEPS fused steps alternate between the two
values (ROUNDS rounds), wall time per step (perf_counter around the block, device idle before) and the device pipeline time of the
last step of each block; median and min per value.  usage: ab_option.py KEY V0 V1 [ROUNDS STEPS QUADS]   GPU only."""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import numpy as np
import mi355_synth as synth, mi355cd
key, v0, v1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 12
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 100
quads = int(sys.argv[6]) if len(sys.argv) > 6 else 500
verts, vidx = synth.cloth_pair(quads)
buf = np.empty((1 << 22, 2), dtype=np.uint32)
with mi355cd.CollisionDetector(verts, vidx) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for _ in range(30): cd.self_collide_into(buf)
    wall = {v0: [], v1: []}; res = {}
    for r in range(rounds):
        for v in ((v0, v1) if r % 2 == 0 else (v1, v0)):
            cd.set_option(key, v)
            for _ in range(5): cd.self_collide_into(buf)
            t0 = time.perf_counter()
            for _ in range(steps): n, rc = cd.self_collide_into(buf)
            wall[v].append((time.perf_counter() - t0) * 1e6 / steps)
            res[v] = (n, cd.fast_stats.pairs_tested, cd.fast_stats.ms_descend_clock)
    for v in (v0, v1):
        print(f"key {key} = {v}: wall per step median {statistics.median(wall[v]):7.2f} us  min {min(wall[v]):7.2f} us   pairs {res[v][0]} tested {res[v][1]} descent (device clock) {res[v][2]*1e3:.1f} us")
