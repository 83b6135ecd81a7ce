import os, sys, time
sys.path.insert(0, "gpu-computing-course_amd/pyhost")
import numpy as np, mi355cd, mi355_synth as synth
v, t = synth.cloth_pair(500)
for mode in (mi355cd.CD_FRAME_REFERENCE, mi355cd.CD_FRAME_AUTO):
    with mi355cd.CollisionDetector(v, t) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_morton_frame(mode)
        for _ in range(10): cd.self_collide(1 << 22, copy=False)
        t0 = time.perf_counter()
        for _ in range(200): cd.self_collide(1 << 22, copy=False)
        print("frame mode", mode, f"{(time.perf_counter()-t0)/200*1e6:.1f} us per step", flush=True)
