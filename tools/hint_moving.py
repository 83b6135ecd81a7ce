#!/usr/bin/env python3
"""The order hint (CD_OPT_ORDER_HINT) on a mesh that MOVES between steps: the 1 M cloth pair, sheet B translated along x by SHIFT quads per frame
(cd_update_vertices + cd_self_collide per frame), the descent's own clock with the hint on and off, frame by frame alternating runs.
usage: hint_moving.py [FRAMES] [SHIFT_IN_QUADS ...]   GPU only."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import numpy as np
import mi355_synth as synth, mi355cd
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
shifts = [float(x) for x in sys.argv[2:]] or [0.0, 0.25, 1.0, 4.0, 16.0]
verts, vidx = synth.cloth_pair(500)
quad = 2.9 / 500
h = verts.shape[0] // 2
buf = np.empty((1 << 22, 2), dtype=np.uint32)
with mi355cd.CollisionDetector(verts, vidx) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for shift in shifts:
        res = {}
        for hint in (1, 0, 1, 0):
            cd.set_option(mi355cd.CD_OPT_ORDER_HINT, hint)
            clk = []; pairs = []
            for f in range(frames):
                v = verts.copy()
                pos = 20.0 - abs(20.0 - (f % 40))                                     # out and back: 26 quads out the sheet leaves the reference's Morton frame (keys beyond 2^60: the sort's next form)
                v[h:, 0] = np.float32(v[h:, 0] + np.float32(pos * shift * quad))      # (float-valued like the loader's: no cell table)
                cd.update_vertices(v)
                n, rc = cd.self_collide_into(buf)
                if f >= 5: clk.append(cd.fast_stats.ms_descend_clock * 1e3)
                pairs.append(n)
            res.setdefault(hint, []).append((statistics.median(clk), pairs[-1]))
        on = statistics.mean(x[0] for x in res[1]); off = statistics.mean(x[0] for x in res[0])
        print(f"sheet B moves {shift:5.2f} quads per frame: descent (device clock, median over frames) hint on {on:5.1f} us  off {off:5.1f} us   pairs in the last frame {res[1][0][1]} / {res[0][0][1]}", flush=True)
