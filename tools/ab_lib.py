"""A/B of two builds of libmi355cd.so on one box: descent time and wall per step (MI355CD_LIB picks the build; run once per build)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import numpy as np, mi355cd, mi355_synth as synth
for name, (v, t) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
    with mi355cd.CollisionDetector(v, t) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
        for _ in range(10): cd.self_collide(1 << 22, copy=False)
        d = []
        t0 = time.perf_counter()
        for _ in range(100): cd.self_collide(1 << 22, copy=False); d.append(cd.stats().ms_descend)
        w = (time.perf_counter() - t0) / 100
        print(f"{os.environ.get('MI355CD_LIB', 'default')[-20:]} {name}: descend {sorted(d)[50]*1e3:.1f} us, wall {w*1e6:.1f} us per step")
