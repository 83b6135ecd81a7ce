"""CPU model: half traversal with small hit subtrees (<= F leaves) handed over as leaf ranges instead of being descended."""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, oracle
so = os.path.join(HERE, "window_sim.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, os.path.join(HERE, "window_sim.c")], check=True)
L = C.CDLL(so)
for name, (v, t) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
    r = oracle.pipeline(v, t); n = t.shape[0]; nw = (n + 63) // 64
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for F in (0, 2, 4, 8, 16, 32, 64):
        out = np.zeros(8, dtype=np.uint64)
        L.flat_sim(n, p(r["left"]), p(r["right"]), p(r["range_first"]), p(r["range_last"]), p(np.ascontiguousarray(r["boxes"])), F, p(out))
        o = out.astype(float)
        print(f"{name} F={F:2d}: tree visits/q {o[0]/n:.3f}  descent levels/wave {o[3]/nw:.2f} | range items/q {o[1]/n:.3f}  leaf tests in ranges/q {o[2]/n:.2f}  flat steps/wave (8 items) {o[6]/nw:.2f} | hits {int(out[4]+out[5])}")
