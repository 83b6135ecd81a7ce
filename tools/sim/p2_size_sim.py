"""CPU model: phase-2 visits of the half traversal by node size, and how many distinct (64-query group, node) pairs they are."""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, oracle
so = os.path.join(HERE, "window_sim.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, os.path.join(HERE, "window_sim.c")], check=True)
L = C.CDLL(so)
for name, (v, t) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
    r = oracle.pipeline(v, t); n = t.shape[0]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    out = np.zeros(48, dtype=np.uint64)
    L.p2_size_sim(n, p(r["left"]), p(r["right"]), p(r["range_first"]), p(r["range_last"]), p(np.ascontiguousarray(r["boxes"])), p(out))
    tot = out[0::2].sum()
    print(name, "phase-2 visits", int(tot), f"({tot/n:.3f} per query)")
    for k in range(24):
        if out[2 * k]:
            print(f"   nodes of 2^{k:2d}.. leaves: visits {int(out[2*k]):8d} ({100*out[2*k]/tot:5.1f} %)  distinct (group, node) {int(out[2*k+1]):8d}  sharing x{out[2*k]/max(1,out[2*k+1]):.1f}")
