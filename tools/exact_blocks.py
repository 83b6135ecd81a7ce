import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import mi355cd, mi355_synth as synth
v, t = synth.cloth_pair(500)
with mi355cd.CollisionDetector(v, t) as cd:
    for eb in (512, 768, 1024, 1280, 1536, 2048, 3072, 4096):
        cd.set_option(101, eb)
        ms = []
        for _ in range(12):
            cd.self_collide(); ms.append(cd.stats().ms_exact)
        print(eb, f"{min(ms[2:])*1e3:.1f} us (median {sorted(ms[2:])[5]*1e3:.1f})")
