import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import statistics, mi355rt, mi355_synth as synth
spheres, shifts = synth.sphere_scene(4096, 4096, seed=7)
with mi355rt.RayTracer(spheres, 4096) as rt:
    rt.set_mode(mi355rt.RT_MODE_BINNED)
    ms = []
    for _ in range(40): rt.render(shifts, download=False); ms.append(rt.stats().ms_render)
    ms = ms[5:]
    print(f"binned: min {min(ms)*1e3:.1f} us, median {statistics.median(ms)*1e3:.1f} us, max {max(ms)*1e3:.1f} us")
