import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import mi355cd, mi355_synth as synth
for name, (v, t) in (("cloth 1M", synth.cloth_pair(500)), ("soup 1M", synth.soup(1_000_000, 0.01, 1234))):
    for opt, label in ((0, "hybrid"), (2, "half-key"), (1, "full")):
        with mi355cd.CollisionDetector(v, t) as cd:
            cd.set_option(mi355cd.CD_OPT_SORT_FULL, opt)
            ms = []
            for _ in range(8):
                cd.morton_sort(); ms.append(cd.stats().ms_sort + cd.stats().ms_morton)
            print(f"{name:9s} {label:9s} morton+sort {min(ms[2:])*1e3:7.1f} us  passes {cd.stats().sort_passes}")
