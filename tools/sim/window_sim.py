"""CPU model of the windowed half traversal (tools/sim/window_sim.c): counts per K for BASELINE config 3 / a 1 M soup."""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, oracle
so = os.path.join(HERE, "window_sim.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, os.path.join(HERE, "window_sim.c")], check=True)
L = C.CDLL(so)
for name, (v, t) in (("cloth1M", synth.cloth_pair(int(sys.argv[1]) if len(sys.argv) > 1 else 500)), ("soup1M", synth.soup(1_000_000 if len(sys.argv) < 2 else 4 * int(sys.argv[1]) ** 2, 0.01, 1234))):
    r = oracle.pipeline(v, t)
    n = t.shape[0]; nw = (n + 63) // 64
    print(name, "n", n, "pairs_tested", r["stats"].pairs_tested, "node_visits", r["stats"].node_visits)
    for K in (0, 2, 4, 8, 12, 16, 24, 32):
        out = np.zeros(16, dtype=np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        L.window_sim(n, p(r["left"]), p(r["right"]), p(r["range_first"]), p(r["range_last"]), p(np.ascontiguousarray(r["boxes"])), K, p(out))
        o = out.astype(float)
        print(f" K={K:2d}: window tests/q {o[0]/n:5.2f} hits/q {o[1]/n:.3f} | hops in wave/q {o[2]/n:.2f} tests {o[3]/n:.2f} | chain steps/wave {o[4]/nw:.2f} lane tests/q {o[5]/n:.2f} | "
              f"p2 visits/q {o[6]/n:.3f} descents/q {o[9]/n:.3f} tree leaf hits/q {o[7]/n:.3f} | max hops/wave {o[8]/nw:.2f} max private/wave {o[10]/nw:.2f} | total hits {int(out[1]+out[7])} (want {(r['stats'].pairs_tested - n)//2 if True else 0}+self)")
