import sys; sys.path.insert(0, "gpu-computing-course_amd/pyhost")
import mi355cd, mi355_synth as synth
v, t = synth.cloth_pair(500)
for flags, name in [(0, "normal"), (16, "no ticket"), (2, "no lookback"), (4, "no ranking"), (8, "no scatter"), (2 | 4, "no lookback+rank"), (2 | 4 | 8 | 16, "load only")]:
    with mi355cd.CollisionDetector(v, t) as cd:
        cd.set_option(103, flags)
        ms = []
        for _ in range(6):
            try: cd.morton_sort()
            except Exception as e: pass
            ms.append(cd.stats().ms_sort)
        print(f"{name:20s} sort stage {min(ms[1:])*1e3:7.1f} us")
