#!/usr/bin/env python3
"""Occupancy experiment: traversal time vs extra LDS padding per workgroup (limits workgroups/CU)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd
verts, vidx = synth.cloth_pair(500)
with mi355cd.CollisionDetector(verts, vidx) as cd:
    cd.self_collide()
    for variant in (1,):
        for qpw in (64,):
            for pad in (0, 8192, 16384, 24576, 36864, 65536):
                cd.set_option(0, variant); cd.set_option(1, qpw); cd.set_option(100, pad)
                t = []
                for _ in range(6):
                    cd.find_collisions(cap=1 << 22); t.append(cd.stats().ms_traverse)
                blocks = 160 * 1024 // (18432 + pad)
                print(f"variant={variant} qpw={qpw} pad={pad:6d} (<= {blocks} wg/CU)  median={statistics.median(t)*1e3:7.1f} us")
