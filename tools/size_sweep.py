import os, sys, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import mi355cd, mi355_synth as synth
cases = [("soup 100k e=0.02 (config 2)", lambda: synth.soup(100_000, 0.02, 1234)), ("cloth 80k", lambda: synth.cloth_pair(141)),
         ("cloth 1M (config 3)", lambda: synth.cloth_pair(500)), ("soup 1M e=0.01", lambda: synth.soup(1_000_000, 0.01, 1234)),
         ("cloth 4M", lambda: synth.cloth_pair(1000)), ("soup 8M e=0.005", lambda: synth.soup(8_000_000, 0.005, 1234))]
for name, gen in cases:
    v, t = gen()
    with mi355cd.CollisionDetector(v, t) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
        for _ in range(3): pairs, n, rc = cd.self_collide(cap=1 << 23)
        K = 20
        t0 = time.perf_counter()
        for _ in range(K): cd.self_collide(cap=1 << 23)
        wall = (time.perf_counter() - t0) / K
        st = cd.stats()
        print(f"{name:28s} n={t.shape[0]:8d} wall {wall*1e3:7.3f} ms  device {st.ms_pipeline:7.3f} ms  descend {st.ms_descend*1e3:7.1f} us  pairs {n:7d} tested {st.pairs_tested:9d} sort_passes {st.sort_passes} overflows {st.stack_overflows}", flush=True)
