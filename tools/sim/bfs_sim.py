"""CPU model: wave-steps of a level-synchronous phase 2 shared by G consecutive queries (tools/sim/window_sim.c bfs_sim)."""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, oracle
so = os.path.join(HERE, "window_sim.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, os.path.join(HERE, "window_sim.c")], check=True)
L = C.CDLL(so)
for name, (v, t) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234)), ("soup100k", synth.soup(100_000, 0.02, 1234))):
    r = oracle.pipeline(v, t)
    n = t.shape[0]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for G in (64, 128, 256, 512, 1024):
        out = np.zeros(8, dtype=np.uint64)
        L.bfs_sim(n, p(r["left"]), p(r["right"]), p(r["range_last"]), p(np.ascontiguousarray(r["boxes"])), G, p(out))
        o = out.astype(float)
        print(f"{name} G={G:4d}: wave-steps per 64 queries {o[0]/(n/64):.2f}  levels per workgroup {o[1]/o[4]:.2f}  (levels with < 32 items: {o[5]/o[4]:.2f})  max frontier {int(out[2])}  visits/q {o[3]/n:.3f}  lane use {o[3]/(64*o[0]):.2f}")
