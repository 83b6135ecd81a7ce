import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import mi355rt, mi355_synth as synth
spheres, shifts = synth.sphere_scene(4096, 4096, seed=7)
with mi355rt.RayTracer(spheres, 4096) as rt:
    rt.set_mode(mi355rt.RT_MODE_BINNED)
    for _ in range(12): rt.render(shifts, download=False)
    print(rt.stats().ms_render)
