import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd
verts, vidx = synth.cloth_pair(500)
with mi355cd.CollisionDetector(verts, vidx) as cd:
    cd.self_collide()
    for g in (0, 1024, 2048, 4096, 8192):
        cd.set_option(100, g)
        t = []
        for _ in range(10):
            cd.self_collide(); t.append(cd.stats().ms_refit)
        print("cross kernels grid", g or "default(1954)", "refit stage median %.1f us" % (statistics.median(t) * 1e3))
