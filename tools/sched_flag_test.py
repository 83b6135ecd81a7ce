"""Does the runtime's wait policy move the step's wall time?  hipSetDeviceFlags(spin / yield / blocking) before the first context."""
import os, sys, time, ctypes as C
flag = int(sys.argv[1]) if len(sys.argv) > 1 else -1
hip = C.CDLL("libamdhip64.so")
if flag >= 0:
    print("hipSetDeviceFlags", flag, "->", hip.hipSetDeviceFlags(C.c_uint(flag)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import numpy as np, mi355cd, mi355_synth as synth
v, t = synth.cloth_pair(500)
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 2)
    for _ in range(20): cd.self_collide(1 << 22, copy=False)
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200): cd.self_collide(1 << 22, copy=False)
        print(f"flag {flag}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step", flush=True)
