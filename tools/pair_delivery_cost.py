"""What handing the pair list to the caller costs per step: the fused step with a pair buffer against the same step counting only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import numpy as np, mi355cd, mi355_synth as synth
v, t = synth.cloth_pair(500)
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
    cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for rep in range(3):
        for cap in (1 << 22, 0):
            for _ in range(10): cd.self_collide(cap, copy=False)
            t0 = time.perf_counter()
            for _ in range(200): p, n, rc = cd.self_collide(cap, copy=False)
            w = (time.perf_counter() - t0) / 200
            print(f"cap {cap}: {w*1e6:.1f} us per step, {n} pairs", flush=True)
