#!/usr/bin/env python3
"""Occupancy sweep of the default descent: extra dynamic LDS per workgroup (debug key 100) lowers the waves a CU holds (3648 B static per
one-wave workgroup; 160 KB per CU).  One process, interleaved rounds, the kernel's own device clock.  GPU only."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import numpy as np, mi355_synth as synth, mi355cd

pads = [int(x) for x in sys.argv[1:]] or [0, 2050, 2952, 4252, 6052, 10000]
for name, (verts, vidx) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
    buf = np.empty((1 << 22, 2), dtype=np.uint32)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for _ in range(10): cd.self_collide_into(buf)
        t = {p: [] for p in pads}
        for r in range(10):
            for p in pads:
                cd.set_option(100, p)
                for _ in range(5): cd.self_collide_into(buf); t[p].append(cd.fast_stats.ms_descend_clock)
        for p in pads:
            print(f"{name} lds pad {p:6d} B ({163840 // (3648 + p):3d} waves per CU by LDS, at most 32): descend median {statistics.median(t[p]) * 1e3:7.1f} us  min {min(t[p]) * 1e3:7.1f} us")
