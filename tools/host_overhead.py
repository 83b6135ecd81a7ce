"""Where the wall time of one step goes beyond the device pipeline: raw C-ABI call vs the Python step."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import numpy as np, mi355cd, mi355_synth as synth, mi355_multi as multi
v, t = synth.cloth_pair(500)
import torch
eng = multi.HipEngine(v, t, None, torch.device("cuda:0"), mi355cd.CD_FRAME_REFERENCE)
cd = eng.cd
cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
buf = np.empty((1 << 22, 2), dtype=np.uint32); n = C.c_uint64(0)
ptr = buf.ctypes.data_as(C.c_void_p)
for _ in range(20): cd.lib.cd_self_collide(cd._ctx, ptr, 1 << 22, C.byref(n))
K = 300
t0 = time.perf_counter()
for _ in range(K): cd.lib.cd_self_collide(cd._ctx, ptr, 1 << 22, C.byref(n))
raw = (time.perf_counter() - t0) / K
pipe = 0.0
t0 = time.perf_counter()
for _ in range(K):
    multi.collide_step(eng, None, 0, 1, 1 << 22, None); pipe += cd.stats().ms_pipeline
step = (time.perf_counter() - t0) / K
print(f"raw C-ABI call {raw*1e6:.1f} us/step, python step {step*1e6:.1f} us/step, device pipeline {pipe/K*1e3:.1f} us, pairs {n.value}")
