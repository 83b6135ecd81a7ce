#!/usr/bin/env python3
"""The inline exact stage (CD_OPT_INLINE_EXACT) against descent + k_exact, and over the number of consumer workgroups (CD_DBG_POOL_CONSUMERS), on the 1 M cloth
and the 1 M soup: wall time per step (blocks of STEPS steps, ROUNDS rounds, round-robin) and the kernel's own clock.  usage: pool_sweep.py [ROUNDS STEPS]   GPU only."""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import numpy as np
import mi355_synth as synth, mi355cd
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
configs = [("k_exact", 0, 0)] + [(f"inline/{c or 'auto'}", 1, c) for c in (0, 256, 512, 1024, 2048, 4096)]
buf = np.empty((1 << 22, 2), dtype=np.uint32)
for name, (verts, vidx) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for _ in range(30): cd.self_collide_into(buf)
        wall = {c[0]: [] for c in configs}; res = {}
        for r in range(rounds):
            for label, inl, cons in (configs if r % 2 == 0 else configs[::-1]):
                cd.set_option(mi355cd.CD_OPT_INLINE_EXACT, inl); cd.debug_set(mi355cd.CD_DBG_POOL_CONSUMERS, cons)
                for _ in range(5): cd.self_collide_into(buf)
                t0 = time.perf_counter()
                for _ in range(steps): n, rc = cd.self_collide_into(buf)
                wall[label].append((time.perf_counter() - t0) * 1e6 / steps)
                res[label] = (n, cd.fast_stats.pairs_tested, cd.fast_stats.ms_descend_clock, cd.debug_get(mi355cd.CD_DBG_GET_POOL_FALLBACKS))
        for label, _, _ in configs:
            print(f"{name} {label:12s}: wall per step median {statistics.median(wall[label]):7.2f} us  min {min(wall[label]):7.2f} us   pairs {res[label][0]} tested {res[label][1]} "
                  f"kernel (device clock) {res[label][2]*1e3:.1f} us  pool fallbacks {res[label][3]}", flush=True)
