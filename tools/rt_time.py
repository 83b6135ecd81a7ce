#!/usr/bin/env python3
"""Frame time of the two ray-tracer modes at BASELINE config 5 (4096^2, 4096 spheres).  GPU only."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355rt
spheres, shifts = synth.sphere_scene(4096, 4096, seed=7)
with mi355rt.RayTracer(spheres, 4096) as rt:
    for name, mode in (("binned", mi355rt.RT_MODE_BINNED),):
        rt.set_mode(mode)
        rt.render(shifts, download=False)
        ms = []
        for _ in range(20):
            rt.render(shifts, download=False); ms.append(rt.stats().ms_render)
        print(name, "median %.1f us  min %.1f us  tests/frame %d" % (statistics.median(ms) * 1e3, min(ms) * 1e3, rt.stats().sphere_tests))
        rb = []
        for _ in range(5):
            rt.render_repeat(shifts, 16, download=False); rb.append(rt.stats().ms_render)
        print(name, "16 frames back to back: median %.1f us per frame" % (statistics.median(rb) * 1e3))
        rt.anim_init()
        la = []
        try:
            for _ in range(5):
                rt.anim_loop(32, 2, 35, 1, 18, download=False); la.append(rt.stats().ms_render)
            print(name, "animation loop, 32 frames, one launch per frame (rt_anim_loop): median %.1f us per frame" % (statistics.median(la) * 1e3))
        except mi355rt.RtError:
            print(name, "no rt_anim_loop in this build")
