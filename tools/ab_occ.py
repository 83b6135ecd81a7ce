#!/usr/bin/env python3
"""Occupancy sweep of the default descent: extra dynamic LDS per workgroup (debug key 100) lowers workgroups per CU.
One process, interleaved rounds.  GPU only."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd

pads = [int(x) for x in sys.argv[1:]] or [0, 8192, 22528, 50000]
verts, vidx = synth.cloth_pair(500)
with mi355cd.CollisionDetector(verts, vidx) as cd:
    cd.self_collide()
    t = {p: [] for p in pads}
    for r in range(10):
        for p in pads:
            cd.set_option(100, p)
            cd.find_collisions(cap=1 << 22)
            t[p].append(cd.stats().ms_descend)
    for p in pads:
        print(f"lds pad {p:6d} B: descend median {statistics.median(t[p]) * 1e3:7.1f} us  min {min(t[p]) * 1e3:7.1f} us")
