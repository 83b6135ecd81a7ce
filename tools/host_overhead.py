"""Where the wall time of one step goes beyond the device pipeline: raw C-ABI call vs the Python forms of the step (same box, same run)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import numpy as np, mi355cd, mi355_synth as synth
v, t = synth.cloth_pair(500)
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
    buf = np.empty((1 << 22, 2), dtype=np.uint32); n = C.c_uint64(0)
    ptr = buf.ctypes.data_as(C.c_void_p)
    for _ in range(20): cd.lib.cd_self_collide(cd._ctx, ptr, 1 << 22, C.byref(n))
    K = 300
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(K): cd.lib.cd_self_collide(cd._ctx, ptr, 1 << 22, C.byref(n))
        raw = (time.perf_counter() - t0) / K
        t0 = time.perf_counter(); pipe = 0.0
        for _ in range(K): cd.self_collide(1 << 22, copy=False); pipe += cd.stats().ms_pipeline
        view = (time.perf_counter() - t0) / K
        t0 = time.perf_counter()
        for _ in range(K): cd.self_collide(1 << 22); cd.stats()
        copy = (time.perf_counter() - t0) / K
        print(f"raw C-ABI call {raw*1e6:.1f} us/step | python, pairs as a view + stats {view*1e6:.1f} | python, pairs copied + stats {copy*1e6:.1f} | device pipeline {pipe/K*1e3:.1f} us, pairs {n.value}")
