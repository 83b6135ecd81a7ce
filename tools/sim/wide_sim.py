"""CPU model: wave-steps of the half traversal with binary records (today) and with records that hold two binary levels (tools/sim/wide_sim.c).
usage: python3 tools/sim/wide_sim.py [cloth1M] [cfg4_2M] [soup1M]"""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, oracle
so = os.path.join(HERE, "wide_sim.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, os.path.join(HERE, "wide_sim.c")], check=True)
L = C.CDLL(so)
p = lambda a: a.ctypes.data_as(C.c_void_p)
# what a wave-step costs on the device (1 M cloth, tools/half_diag.py): a wave lives 23.6 us -- phase 0 8 %, 1a 14 % (6.2 steps), 1b 15 % (10.8 steps),
# phase 2 53 % (9.7 steps), hand-over 10 %
LIFE = 23.6
US = {"1a": 0.14 * LIFE / 6.2, "1b": 0.15 * LIFE / 10.8, "2": 0.53 * LIFE / 9.7}
FIXED = (0.08 + 0.10) * LIFE
for name in (sys.argv[1:] or ["cloth1M", "cfg4_2M"]):
    if name == "cloth1M":
        v, t = synth.cloth_pair(500); r = oracle.pipeline(v, t)
    elif name == "soup1M":
        v, t = synth.soup(1_000_000, 0.01, 1234); r = oracle.pipeline(v, t)
    else:
        v, t, ids, _, _ = synth.config4_merged(8, 250); off, sp, lay = oracle.auto_frame(v, t); r = oracle.pipeline(v, t, ids, off=off, span=sp, layout=lay)
    n = t.shape[0]
    out = np.zeros(16, dtype=np.uint64)
    L.wide_sim(n, p(r["left"]), p(r["right"]), p(r["range_last"]), p(np.ascontiguousarray(r["boxes"])), p(out))
    w = float(out[0])
    s1a, s1b, s2, s2w, s1aw, s1bw = (float(out[k]) / w for k in (1, 2, 3, 4, 5, 6))
    print(f"{name}: {n} triangles, {int(out[0])} waves")
    print(f"   steps a wave, binary: 1a {s1a:.2f}  1b {s1b:.2f}  phase 2 {s2:.2f}   (device, 1 M cloth: 6.2 / 10.8 / 9.7)   lanes busy in phase 2 {float(out[8]) / float(out[3]):.1f} of 64, longest wave {int(out[12])} steps")
    print(f"   steps a wave, wide  : 1a {s1aw:.2f}  1b {s1bw:.2f}  phase 2 {s2w:.2f}   lanes busy in phase 2 {float(out[9]) / max(float(out[4]), 1):.1f}, longest wave {int(out[13])} steps;  box tests x{float(out[11]) / float(out[10]):.2f}")
    life_b = FIXED + s1a * US["1a"] + s1b * US["1b"] + s2 * US["2"]
    for label, c2, c1 in (("a wide step costs what a binary one does (pure latency)", 1.0, 1.0), ("a wide step costs 1.25 x (128-byte fetch, four box tests)", 1.25, 1.25), ("1.5 x", 1.5, 1.5)):
        only2 = FIXED + s1a * US["1a"] + s1b * US["1b"] + s2w * US["2"] * c2
        both = FIXED + s1aw * US["1a"] * c1 + s1bw * US["1b"] * c1 + s2w * US["2"] * c2
        print(f"   wave life {life_b:.1f} us -> {only2:.1f} (phase 2 wide) / {both:.1f} (chain hops two at a time too)   [{label}]: kernel x{only2 / life_b:.2f} / x{both / life_b:.2f}")
