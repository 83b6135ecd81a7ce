"""What the time stamps on the dispatch packets cost: wall time of the raw C call with the block build / descent / exact kernels stamped or not (debug key 105)."""
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
        for mask in (7, 2, 0, 6, 3):
            cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, mask)
            for _ in range(10): cd.lib.cd_self_collide(cd._ctx, ptr, 1 << 22, C.byref(n))
            t0 = time.perf_counter()
            for _ in range(K): cd.lib.cd_self_collide(cd._ctx, ptr, 1 << 22, C.byref(n))
            print(f"stamps mask {mask} (1 block build, 2 descent, 4 exact): {(time.perf_counter() - t0) / K * 1e6:.1f} us per step")
