"""CPU model (round 4): the half traversal with phase 2 split off into a second kernel that works a list of
(query, subtree) items off, idle lanes taking the next item of their wave's chunk.  Needs no GPU.  python tools/sim/pool_sim.py"""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, oracle
so = os.path.join(HERE, "pool_sim.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, os.path.join(HERE, "pool_sim.c")], check=True)
L = C.CDLL(so); L.pool_items.restype = C.c_uint64
p = lambda a: a.ctypes.data_as(C.c_void_p)
for name, (v, t) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234)), ("soup100k", synth.soup(100_000, 0.02, 1234))):
    r = oracle.pipeline(v, t); n = t.shape[0]
    cap = 8 * n
    iv = np.zeros(cap, dtype=np.uint32); out = np.zeros(80, dtype=np.uint64)
    cnt = L.pool_items(n, p(r["left"]), p(r["right"]), p(r["range_last"]), p(np.ascontiguousarray(r["boxes"])), p(iv), C.c_uint64(cap), p(out))
    iv = iv[:cnt]
    nw = (n + 63) // 64
    print(f"{name}: {cnt} items ({cnt/n:.3f} per query), phase-2 visits {int(out[1])} ({out[1]/n:.3f} per query, {out[1]/cnt:.2f} per item, max {int(out[2])}); "
          f"today: {out[3]/nw:.2f} phase-2 steps per wave without sharing (measured with sharing: 9.7 cloth) = {int(out[3])} wave-steps, lane use {out[1]/(64*out[3]):.2f}")
    h = out[8:72]
    print("   items by visits: " + "  ".join(f"{k}:{int(h[k])}" for k in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32) ) + f"  >=63:{int(h[63])};  p50 {int(np.percentile(iv,50))} p90 {int(np.percentile(iv,90))} p99 {int(np.percentile(iv,99))}")
    for waves in (2048, 4096, 8192, 16384):
        Cc = max(64, int(np.ceil(cnt / waves / 64.0)) * 64)
        for share in (0, 1):
            o = np.zeros(4, dtype=np.uint64)
            L.pool_chunks(p(iv), C.c_uint64(cnt), Cc, share, p(o))
            print(f"   item kernel, chunk {Cc:5d} items ({int(o[0]):5d} waves), {'ideal sharing' if share else 'no sharing   '}: wave-steps {int(o[1]):7d} "
                  f"(today {int(out[3])}: x{out[3]/max(1,o[1]):.2f} fewer), per wave mean {o[1]/o[0]:.1f} max {int(o[2])}, lane use {o[3]/(64*max(1,o[1])):.2f}")
