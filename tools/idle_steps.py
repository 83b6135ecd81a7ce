"""The static 1 M cloth stepped with an idle host between steps (SLEEP_MS, default 15): what a step costs when the GPU has been idle -- to tell the effect of
idling (clocks, caches) from that of other geometry in the moving-mesh figures.  To be run under rocprofv3 --kernel-trace --stats.  usage: idle_steps.py [SLEEP_MS] [UPLOAD]"""
import os, sys, time
sys.path[:0] = [os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpu-computing-course_amd", "pyhost")]
import numpy as np, mi355_synth as synth, mi355cd
ms = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
upload = len(sys.argv) > 2 and sys.argv[2] == "1"
v, t = synth.cloth_pair(500)
buf = np.empty((1 << 22, 2), dtype=np.uint32)
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for _ in range(120):
        if ms > 0: time.sleep(ms * 1e-3)
        if upload: cd.update_vertices(v)
        n, rc = cd.self_collide_into(buf)
    print("sleep", ms, "upload", upload, "pairs", n)
